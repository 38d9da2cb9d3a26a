"""
The two kernel generations of libehmpc against each other on the device (same inputs, same
C-ABI): generation 2 (shared constant LP block in LDS, psi-coordinates, several wavefronts
per workgroup) must reproduce generation 1 (one wavefront per workgroup, barycentric
coordinates) for every oracle kind and grow the identical tree -- also with the sign-only
termination of the suboptimality test.  Both are checked against the CPU oracle elsewhere
(test_gpu_oracles.py, test_gpu_partition.py); this file pins them to each other at sizes
the CPU oracle cannot reach in seconds.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

RTOL = 1e-8


def rel(a, b):
    return float(np.max(np.abs(a - b) / (1. + np.abs(b))))


def both(gp, fn):
    gp.set_solver(1)
    a = fn()
    gp.set_solver(2)
    b = fn()
    return a, b


def test_wave_primitives_of_every_instance():
    from explicit_hybrid_mpc_amd import engine
    out = engine.selftest()
    assert out.shape[0] == 31     # 28 wave-local + 2 streaming wide + the LDS-resident wide family
    assert np.abs(out - np.array([1072., 99., 25., 1. / 3., -1.])).max() < 1e-13


@pytest.mark.parametrize('kind', ['di', 'lin', 'pwa'])
def test_batched_oracles_agree(kind):
    from explicit_hybrid_mpc_amd import engine, examples
    mpc = helpers.make_instance(kind, 0)
    can = mpc.compile()
    gp = engine.GpuProblem(can, 0.05, 0.05)
    rng = np.random.default_rng(3)
    half = examples.theta_box(mpc)
    n = 700
    theta = rng.uniform(-1.3, 1.3, (n, can.p)) * half
    delta = can.deltas[rng.integers(can.n_delta, size=n)]     # unsorted on purpose
    (f1, t1), (f2, t2) = both(gp, lambda: gp.feasible_ptd(theta, delta))
    assert (f1 == f2).all() and rel(t2, t1) < 1e-7
    ok = f1
    (J1, u1, s1, _), (J2, u2, s2, _) = both(gp, lambda: gp.solve_ptd(theta[ok], delta[ok]))
    assert (s1 == 0).all() and (s2 == 0).all()
    assert rel(J2, J1) < RTOL
    (Ja, _, da), (Jb, _, db) = both(gp, lambda: gp.solve_pt(theta))
    assert (da == db).all() and rel(Jb[da >= 0], Ja[da >= 0]) < RTOL
    R = helpers.random_simplices(mpc, rng, 300, -2.5, -0.5)
    didx, vJ, _ = gp.v_r(R)
    keep = didx >= 0
    R, vJ, dl = R[keep], vJ[keep], can.deltas[didx[keep]]
    (ta, aa, sa), (tb, ab, sb) = both(gp, lambda: gp.slack(R, vJ, dl))
    assert (sa == 0).all() and (sb == 0).all()
    assert rel(tb, ta) < 1e-7 and ((ta >= 0) == (tb >= 0)).all()
    assert np.abs(ab.sum(axis=1) - 1).max() < 1e-9 and ab.min() > -1e-7
    (ma, _), (mb, _) = both(gp, lambda: gp.min_simplex(R, dl))
    assert rel(mb, ma) < RTOL
    (ca, tba), (cb, tbb) = both(gp, lambda: gp.bar_e(R, vJ))
    assert (ca == cb).all()
    gp.close()


# abs_frac = 0.02 is bench.py's full-size workload (BASELINE configs[1]): 1 610 186 nodes,
# 805 104 regions -- far beyond what the CPU oracle can finish, so parity at that size rests on
# size-independent properties: three independent numerical paths (one-wavefront kernels,
# shared-block kernels to full accuracy, shared-block kernels with the sign-only stop) must grow
# the bit-identical tree, and the closed leaves must tile Theta (volume closure).
@pytest.mark.parametrize('abs_frac', [0.25, 0.08, 0.02])
def test_partition_identical_across_generations_and_decide_modes(abs_frac):
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    mpc = helpers.make_instance('lin', 0)
    gp = engine.GpuProblem(mpc.compile(), 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    eps_a = float(np.max(gp.solve_pt(abs_frac * V)[0]))
    gp.set_eps(eps_a, 1e-2)
    roots, _ = ehm_tools.delaunay_roots(V)
    trees = []
    for gen, full, eng in ((1, 1, 0), (2, 1, 0), (2, 0, 0), (2, 0, 1)):
        # the last one: everything in ONE launch of the persistent frontier kernel, whose
        # midpoint LPs run in the wider (suboptimality-test) instance -- a fourth arithmetic
        gp.set_solver(gen)
        gp.set_option('decide_full', full)
        trees.append(gp.partition(roots, action='ecc', max_nodes=1 << 22, engine=eng))
    gp.close()
    ref = trees[0]
    if abs_frac == 0.02:
        assert ref.n_nodes == 1610186 and ref.info['n_closed'] == 805104
    total = np.prod(2 * examples.theta_box(mpc))
    assert abs(ref.info['volume_closed'] - total) <= 1e-9 * total
    for t in trees[1:]:
        assert t.n_nodes == ref.n_nodes
        assert np.array_equal(t.vertices, ref.vertices)          # bit-identical geometry
        assert np.array_equal(t.left, ref.left)
        assert np.array_equal(t.flags & 1, ref.flags & 1)        # same closed leaves
        assert rel(t.vertex_costs.ravel(), ref.vertex_costs.ravel()) < RTOL
        assert abs(t.info['volume_closed'] - ref.info['volume_closed']) < 1e-9
    assert trees[3].info['decide_launches'] == 1
    full, sign = trees[1], trees[2]
    assert rel(full.tstar, ref.tstar) < 1e-6
    # the sign-only stop records a lower bound of |t*| and needs fewer iterations
    assert sign.info['min_margin'] <= full.info['min_margin'] * (1 + 1e-9)
    assert sign.info['decide_iters'] < 0.8 * full.info['decide_iters']
    assert ((sign.tstar >= 0) == (full.tstar >= 0)).all()


def test_midpoint_first_flow_grows_the_same_tree():
    """Option "mid_first" (include/ehmpc.h): identical tree, fewer suboptimality-test LPs."""
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    mpc = helpers.make_instance('lin', 0)
    gp = engine.GpuProblem(mpc.compile(), 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    gp.set_eps(float(np.max(gp.solve_pt(0.05 * V)[0])), 1e-2)
    roots, _ = ehm_tools.delaunay_roots(V)
    gp.set_option('mid_first', 0)
    ref = gp.partition(roots, action='ecc', max_nodes=1 << 22)
    gp.set_option('mid_first', 1)
    new = gp.partition(roots, action='ecc', max_nodes=1 << 22)
    gp.close()
    assert new.info['witness_open'] > 0
    assert new.info['decide_solves'] < ref.info['decide_solves']
    assert new.n_nodes == ref.n_nodes
    assert np.array_equal(new.vertices, ref.vertices) and np.array_equal(new.left, ref.left)
    assert np.array_equal(new.flags & 1, ref.flags & 1)
    assert rel(new.vertex_costs.ravel(), ref.vertex_costs.ravel()) < RTOL
    assert ((new.tstar >= 0) == (ref.tstar >= 0)).all()


def test_inherited_witnesses_grow_the_same_tree():
    """
    Option "inherit_witness" (include/ehmpc.h, DevTree::wit): a node the suboptimality-test LP
    finds open hands the point that proved it to the child that contains it.  Identical tree,
    fewer LPs; budgeted rounds of the persistent kernel and sharded launches keep the witnesses
    with the nodes (they live in the node pool), nodes received from another rank carry none.
    """
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    mpc = helpers.make_instance('lin', 0)
    gp = engine.GpuProblem(mpc.compile(), 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    gp.set_eps(float(np.max(gp.solve_pt(0.05 * V)[0])), 1e-2)
    roots, _ = ehm_tools.delaunay_roots(V)
    gp.set_option('inherit_witness', 0)
    ref = gp.partition(roots, action='ecc', max_nodes=1 << 22)
    gp.set_option('decide_full', 1)
    full = gp.partition(roots, action='ecc', max_nodes=1 << 22)
    gp.set_option('decide_full', 0)
    gp.set_option('inherit_witness', 1)
    new = gp.partition(roots, action='ecc', max_nodes=1 << 22)
    # the same run in budgeted rounds (ehm_partition_advance)
    run = gp.begin(roots, action='ecc', max_nodes=1 << 22)
    n_rounds = 0
    while True:
        n_rounds += 1
        if run.advance(20000) == 0:
            break
    rounds = run.finish()
    assert n_rounds >= 3
    gp.close()
    assert ref.info['witness_inherited'] == 0 and full.info['witness_inherited'] == 0
    assert new.info['witness_inherited'] > 0.1 * new.info['n_leaves']
    assert new.info['lp_solves'] < 0.93 * ref.info['lp_solves']
    assert rounds.info['witness_inherited'] > 0
    for t in (new, full):
        assert t.n_nodes == ref.n_nodes
        assert np.array_equal(t.vertices, ref.vertices) and np.array_equal(t.left, ref.left)
        assert np.array_equal(t.flags & 1, ref.flags & 1)
        assert rel(t.vertex_costs.ravel(), ref.vertex_costs.ravel()) < RTOL
        assert ((t.tstar >= 0) == (ref.tstar >= 0)).all()
    # the run in rounds numbers its nodes in another order: compare the cells themselves
    assert rounds.n_nodes == ref.n_nodes
    cells = {ref.vertices[k].tobytes(): (ref.left[k] < 0, ref.flags[k] & 1)
             for k in range(ref.n_nodes)}
    assert len(cells) == ref.n_nodes
    for k in range(rounds.n_nodes):
        assert cells[rounds.vertices[k].tobytes()] == (rounds.left[k] < 0, rounds.flags[k] & 1)
    # a decision taken by a witness has a margin above the routing threshold
    assert new.info['min_margin'] == ref.info['min_margin']


def test_shared_midpoint_optima_grow_the_same_tree():
    """
    Option "share_midpoints" (include/ehmpc.h, csrc/ehm_midtable.h): the simplices around an edge
    ask for the same midpoint problem; the first wavefront to ask solves and publishes it, the
    others take the entry.  Bit-identical tree and vertex data (an LP's optimum does not depend on
    who solves it), most midpoint problems gone; the same in budgeted rounds of the kernel.
    """
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    mpc = helpers.make_instance('lin', 0)
    gp = engine.GpuProblem(mpc.compile(), 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    gp.set_eps(float(np.max(gp.solve_pt(0.05 * V)[0])), 1e-2)
    roots, _ = ehm_tools.delaunay_roots(V)
    gp.set_option('share_midpoints', 0)
    ref = gp.partition(roots, action='ecc', max_nodes=1 << 22)
    gp.set_option('share_midpoints', 1)
    new = gp.partition(roots, action='ecc', max_nodes=1 << 22)
    again = gp.partition(roots, action='ecc', max_nodes=1 << 22)      # a second run: table cleared
    run = gp.begin(roots, action='ecc', max_nodes=1 << 22)
    while run.advance(20000):
        pass
    rounds = run.finish()
    gp.close()
    splits = (ref.n_nodes - len(roots)) // 2
    print('\nshare_midpoints: %d splits, %d midpoint optima taken from the table (%.1f per solve), '
          'LP solves %d -> %d, device seconds %.4f -> %.4f' % (
              splits, new.info['midpoints_shared'],
              splits / max(splits - new.info['midpoints_shared'], 1),
              ref.info['lp_solves'], new.info['lp_solves'],
              ref.info['device_seconds'], new.info['device_seconds']))
    assert ref.info['midpoints_shared'] == 0 and ref.info['witness_table'] == 0
    assert new.info['midpoints_shared'] > 0.5 * splits
    # a midpoint optimum taken from the table is a problem not solved; a node proved open by
    # another edge's midpoint in the table saves its suboptimality test (and hands its children a
    # different witness than the test's would have been, so there is no exact count)
    for t in (new, again):
        assert t.info['lp_solves'] + t.info['midpoints_shared'] <= ref.info['lp_solves']
        assert t.info['lp_solves'] + t.info['midpoints_shared'] + t.info['witness_table'] >= \
            ref.info['lp_solves'] - t.info['witness_table']
    # (the slot a midpoint lands in depends on who came first; the count of hits only through
    # neighbourhoods of 16 full slots, which a table 1/8 full practically never has)
    assert abs(again.info['midpoints_shared'] - new.info['midpoints_shared']) <= 1e-3 * splits
    assert rounds.info['midpoints_shared'] > 0.5 * splits
    for t in (new, again):
        assert t.n_nodes == ref.n_nodes
        assert np.array_equal(t.vertices, ref.vertices) and np.array_equal(t.left, ref.left)
        assert np.array_equal(t.flags, ref.flags)
        assert np.array_equal(t.vertex_costs, ref.vertex_costs)           # bit for bit
        assert np.array_equal(t.vertex_inputs, ref.vertex_inputs)
        # recorded margins: a table witness (and the witnesses its node hands down) records a lower
        # bound of the optimum the suboptimality test would have found -- same sign everywhere,
        # identical on the closed leaves
        assert np.array_equal(t.tstar >= 0, ref.tstar >= 0)
        closed = (ref.flags & 1) > 0
        assert np.array_equal(t.tstar[closed], ref.tstar[closed])
    assert rounds.n_nodes == ref.n_nodes
    cells = {ref.vertices[k].tobytes(): (ref.left[k] < 0, ref.flags[k] & 1, ref.vertex_costs[k].tobytes())
             for k in range(ref.n_nodes)}
    for k in range(rounds.n_nodes):
        assert cells[rounds.vertices[k].tobytes()] == (rounds.left[k] < 0, rounds.flags[k] & 1,
                                                       rounds.vertex_costs[k].tobytes())
