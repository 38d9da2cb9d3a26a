"""
Batched explicit-MPC evaluation (ehm_explicit_*) against the CPU restatement of the reference's
ExplicitMPC (oracle/explicit_cpu.py, lib/mpc_library.py:685-792) on a partition grown by the
engine: same containing leaf, same interpolated input.  Both tree layouts are exercised -- the
reference's nested right-spine tree and the engine's flat forest.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def test_explicit_evaluation_matches_reference_walk():
    from explicit_hybrid_mpc_amd import examples, explicit, partition
    from oracle.explicit_cpu import ExplicitCPU
    orc = examples.create_oracle(helpers.make_instance('lin', 0),
                                 examples.box_vertices(examples.theta_box(helpers.make_instance('lin', 0))),
                                 abs_frac=0.3, abs_err=None, rel_err=0.5)
    V = examples.box_vertices(examples.theta_box(orc.mpc))
    root, flat = partition.partition_set(orc, V)
    assert flat.n_nodes > 500
    cpu = ExplicitCPU(root)
    rng = np.random.default_rng(5)
    half = examples.theta_box(orc.mpc)
    X = rng.uniform(-1, 1, (4000, half.size)) * half
    # states hugging the corners / faces of the set (1e-7 inside: exactly ON a face the
    # reference's own eps-test walk is ill-defined and can end in a leaf that does not contain
    # the state)
    X[:16] = V[:16] * (1 - 1e-7)
    X[16:32] = 0.5 * (V[:16] + V[rng.integers(16, size=16)]) * (1 - 1e-7)
    nested = explicit.ExplicitMPC(root, orc)
    forest = explicit.ExplicitMPC(flat, orc)
    u_n, leaf_n, vis_n, _ = nested.evaluate(X, return_info=True)
    u_f, leaf_f, vis_f, _ = forest.evaluate(X, return_info=True)
    u_ref = np.array([cpu(x) for x in X[:1500]])
    cells = [cpu.get_containing_cell(x) for x in X[:1500]]
    same = 0
    for k in range(1500):
        got = nested.nodes[leaf_n[k]].data
        if got is cells[k]:
            same += 1
            assert np.allclose(u_n[k], u_ref[k], rtol=1e-10, atol=1e-12)
        else:
            # a state within rounding of a shared face may be given to the neighbour, which
            # must then contain it too
            E = np.column_stack([v - got.vertices[0] for v in got.vertices[1:]])
            a = np.linalg.solve(E, X[k] - got.vertices[0])
            assert min(a.min(), 1 - a.sum()) > -1e-9
    assert same >= 1490
    # the two layouts walk the same tree
    assert np.allclose(u_n, u_f, rtol=1e-12, atol=1e-13)
    for k in range(0, 4000, 97):
        assert np.array_equal(nested.nodes[leaf_n[k]].data.vertices, flat.vertices[leaf_f[k]])
    # single-state call signature of the reference
    u1, t1 = nested(X[40])
    assert np.allclose(u1, u_n[40]) and t1 >= 0.
    assert nested.get_containing_cell(X[40]) is nested.nodes[leaf_n[40]].data
    nested.close()
    forest.close()


def test_long_spine_goes_through_the_root_locator():
    """
    652 Delaunay roots (the p = 6 box of config 4): the root is found by k_explicit_locate (a
    visibility walk over the face adjacency of the roots; states within 1e-9 of a face fall back to
    the serial walk) and must be the one the reference's serial walk over the right spine ends in
    -- the flat restatement of that walk on the CPU is the checker.
    """
    from explicit_hybrid_mpc_amd import engine, examples, explicit
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from oracle.explicit_cpu import ExplicitFlatCPU
    mpc = helpers.make_instance('chain', 0)
    gp = engine.GpuProblem(mpc.compile(), 1., 1.)
    half = examples.theta_box(mpc)
    V = examples.box_vertices(half)
    J, _, _ = gp.solve_pt(0.5 * V)
    gp.set_eps(float(np.max(J)), 1.0)
    roots, _ = ehm_tools.delaunay_roots(V)
    assert len(roots) >= 128
    flat = gp.partition(np.array(roots), action='ecc')
    gp.close()
    ex = explicit.ExplicitMPC(flat)
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, (3000, half.size)) * half
    # states ON faces of the roots (midpoints of root edges, pulled 1e-12 towards a vertex): the
    # locator must hand them to the serial walk
    R0 = np.array(roots)
    X[:40] = 0.5 * (R0[:40, 0] + R0[:40, 1]) * (1 - 1e-12)
    u, leaf, visited, _ = ex.evaluate(X, return_info=True)
    cpu = ExplicitFlatCPU(flat.vertices, flat.vertex_inputs, flat.left, flat.right,
                          flat.info['n_roots'])
    same = 0
    for k in range(300):
        u_ref, k_ref = cpu(X[k])
        if k_ref == leaf[k]:
            same += 1
            assert np.allclose(u[k], u_ref, rtol=1e-9, atol=1e-11)
        else:       # within rounding of a shared face: the neighbour must contain the state too
            R = flat.vertices[leaf[k]]
            a = np.linalg.solve(np.column_stack([v - R[0] for v in R[1:]]), X[k] - R[0])
            assert min(a.min(), 1 - a.sum()) > -1e-9
    assert same >= 290
    assert (visited >= 1).all() and flat.is_leaf(leaf[0])
    ex.close()


def _interpolated_cost(flat, leaf, X):
    """sum_i alpha_i V_i at X in the leaves `leaf` of a FlatTree (barycentric weights)."""
    R = flat.vertices[leaf]                                  # (n, p+1, p)
    E = np.transpose(R[:, 1:] - R[:, :1], (0, 2, 1))         # columns v_q - v_0
    beta = np.linalg.solve(E, (X - R[:, 0])[:, :, None])[:, :, 0]
    alpha = np.concatenate([1. - beta.sum(axis=1, keepdims=True), beta], axis=1)
    return np.sum(alpha * flat.vertex_costs[leaf], axis=1), alpha


@pytest.mark.parametrize('kind', ['lin', 'lin_quadratic', 'cwh_z'])
def test_partition_delivers_the_epsilon_suboptimality_guarantee(kind):
    """
    The property the whole path exists for, checked end to end on the device at sizes no CPU
    oracle can follow: a leaf is closed iff NO point of it has BOTH  Vbar - V* >= eps_a  and
    Vbar - V* >= eps_r V*  (lib/oracle.py:89-97), so everywhere in the partitioned set
        0 <= Vbar(x) - V*(x) < max(eps_a, eps_r V*(x)),
    Vbar = the leaf's interpolated vertex costs (what ExplicitMPC interpolates the inputs
    with, lib/mpc_library.py:769-792), V* = the implicit law's optimal cost ``P_theta(x)``
    (lib/mpc_library.py:643-660).  Partition engine, point location and the mixed-integer
    oracle all have to be right for this to hold at every sampled state.
    """
    from explicit_hybrid_mpc_amd import examples, explicit, partition
    if kind == 'cwh_z':
        full_set, _, orc = examples.example('cwh_z', abs_frac=0.1, rel_err=0.1)
        n_min = 5000
    else:
        mpc = examples.linear_mpc(0, cost='quadratic' if kind == 'lin_quadratic' else 'inf')
        full_set = examples.box_vertices(examples.theta_box(mpc))
        af, er = (0.1, 0.1) if kind == 'lin_quadratic' else (0.04, 0.01)
        orc = examples.create_oracle(mpc, full_set, abs_frac=af, abs_err=None, rel_err=er)
        n_min = 50000
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    roots, _ = ehm_tools.delaunay_roots(full_set)
    flat = partition.run_engine(orc, roots, action='ecc', max_nodes=1 << 22)
    assert flat.n_nodes > n_min
    law = explicit.ExplicitMPC(flat, orc)
    rng = np.random.default_rng(9)
    half = np.abs(full_set).max(axis=0)
    X = rng.uniform(-1, 1, (100000, half.size)) * half * (1 - 1e-9)
    u, leaf, visits, _ = law.evaluate(X, return_info=True)
    law.close()
    assert (flat.left[leaf] < 0).all() and (flat.flags[leaf] & 1).all()     # closed leaves
    Vbar, alpha = _interpolated_cost(flat, leaf, X)
    assert alpha.min() > -1e-7                                              # x is in its leaf
    Vstar, _, didx = orc.gpu.solve_pt(X)
    assert (didx >= 0).all()
    gap = Vbar - Vstar
    tol = 1e-7 * (1 + np.abs(Vstar))
    assert (gap >= -tol).all()
    bound = np.maximum(orc.eps_a, orc.eps_r * Vstar)
    assert (gap <= bound + tol).all()
    # the guarantee is tight somewhere: the partition is not needlessly fine
    assert (gap > 0.3 * bound).any()
    # the implicit law through its reference interface agrees with the batch
    imp = explicit.ImplicitMPC(orc)
    u1, t1 = imp(X[7])
    assert np.allclose(u1, imp.evaluate(X[7:8])[0], rtol=1e-9, atol=1e-12) and t1 >= 0
    orc.close()
