"""
BASELINE.json configs[4]'s shape (n_x = 8, n_u = 3, 4 modes, N = 8: 65 536 mode sequences) on the
device: the commutation table of a region found by a search over mode PREFIXES
(explicit_hybrid_mpc_amd/sequences.py), checked against the CPU statement
(oracle/prefix_bb.py), against the FULL enumeration solved on the device, and used by the
partition engine.
"""

import itertools
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

RTOL = 1e-6        # FP64 interior point vs HiGHS / vs itself on another table


def region(mpc, frac=0.9, size=0.02, corner=37):
    from explicit_hybrid_mpc_amd import examples
    half = examples.theta_box(mpc)
    p = half.size
    E = np.vstack([np.zeros(p), np.eye(p)]) - 1. / (p + 1)
    return frac * examples.box_vertices(half)[corner] + size * half * E


def test_slot_entry_points_and_table_updates():
    """ehm_simplex_idx_batch / ehm_point_idx_batch address the table by slot;
    ehm_problem_update_blocks replaces blocks in place (every image the kernels read)."""
    from explicit_hybrid_mpc_amd import engine
    from explicit_hybrid_mpc_amd._capi import EhmError
    mpc = helpers.make_instance('pwa_small', 0)
    can = mpc.compile()
    gp = engine.GpuProblem(can, 0.05, 0.2)
    rng = np.random.default_rng(0)
    R = np.array(helpers.random_simplices(mpc, rng, 6))
    nd = can.n_delta
    slot = np.arange(6, dtype=np.int32) % nd
    tau, _, _ = gp.simplex_idx(R, slot, mode=2)
    ok = tau <= 1e-8
    assert ok.any()
    J, alpha, st = gp.simplex_idx(R[ok], slot[ok], mode=0)
    Jref = gp.min_simplex(R[ok], can.deltas[slot[ok]])[0]
    assert np.allclose(J, Jref, rtol=1e-12, atol=1e-12)
    assert np.allclose(alpha.sum(axis=1), 1., atol=1e-9)
    theta = R[ok].mean(axis=1)
    Jp = gp.point_idx(theta, slot[ok])[0]
    Jpr = gp.solve_ptd(theta, can.deltas[slot[ok]])[0]
    assert np.allclose(Jp, Jpr, rtol=1e-12, atol=1e-12)
    # swap blocks 0 and 1: slot 0 now answers as slot 1 did
    a = gp.point_idx(theta, np.zeros(theta.shape[0], dtype=np.int32), feas=True)[0]
    b = gp.point_idx(theta, np.ones(theta.shape[0], dtype=np.int32), feas=True)[0]
    gp.update_blocks(0, can.G[[1, 0]], can.w[[1, 0]], can.S[[1, 0]])
    a2 = gp.point_idx(theta, np.zeros(theta.shape[0], dtype=np.int32), feas=True)[0]
    b2 = gp.point_idx(theta, np.ones(theta.shape[0], dtype=np.int32), feas=True)[0]
    assert np.array_equal(a, b2) and np.array_equal(b, a2)
    with pytest.raises(EhmError):
        gp.simplex_idx(R, np.full(6, nd, dtype=np.int32), mode=0)
    with pytest.raises(EhmError):
        gp.update_blocks(nd - 1, can.G[:2], can.w[:2], can.S[:2])
    gp.close()


def test_prefix_relaxations_on_the_device_match_the_uncondensed_lps():
    """The condensed prefix blocks (PWAMPC.condense_prefix) solve to the optima of the
    uncondensed relaxations (oracle/prefix_bb.PrefixModel, HiGHS)."""
    from explicit_hybrid_mpc_amd import examples, sequences
    from oracle import prefix_bb
    mpc = examples.pwa4_mpc(N=8)
    R = region(mpc)
    table = sequences.PrefixTable(mpc, slots=64)
    prefixes = [(3,), (3, 0), (1, 2), (3, 0, 1), (3, 0, 1, 1, 1), (0, 0, 0, 0, 0, 0),
                (3, 0, 1, 1, 1, 1, 1, 1), (2, 2, 2, 2, 2, 2, 2, 2)]
    got = table.min_cost_on(prefixes, R[None])
    table.close()
    ref = np.array([prefix_bb.prefix_min_on(mpc, pre, R) for pre in prefixes])
    assert np.array_equal(np.isfinite(got), np.isfinite(ref))
    fin = np.isfinite(ref)
    assert fin.sum() >= 4
    assert np.allclose(got[fin], ref[fin], rtol=RTOL, atol=RTOL)


def test_region_table_found_on_the_device_equals_the_cpu_search():
    from explicit_hybrid_mpc_amd import examples, sequences
    from oracle import prefix_bb
    mpc = examples.pwa4_mpc(N=8)
    R = region(mpc)
    seqs, info = sequences.relevant_sequences(mpc, R[None])
    ref, U, dive, _ = prefix_bb.relevant_sequences(mpc, [R])
    assert info['incumbent'] == dive and abs(info['upper_bound'] - U) <= RTOL * (1 + U)
    assert seqs == ref and info['first_feasible'] in seqs
    assert 8 <= len(seqs) <= 256 and info['enumeration'] == 65536
    # the search solved a few thousand LPs, not 65 536 x (phase one + minimum)
    assert info['lp_solves'] < 0.05 * 65536
    print('\\nregion table: %d of 65 536 sequences, levels %s, %d LPs'
          % (len(seqs), info['alive_per_level'], info['lp_solves']))
    # feasibility alone prunes nothing on this system (the inputs reach every mode region)
    with pytest.raises(ValueError):
        sequences.feasible_sequences(mpc, R[None])
    # a region ten times wider keeps more than the engine's table holds: reported, not cut
    with pytest.raises(ValueError) as err:
        sequences.relevant_sequences(mpc, region(mpc, 0.6, 0.2)[None])
    assert 'exceeds 256' in str(err.value)


def test_region_table_reproduces_the_full_enumeration():
    """
    P_theta over ALL 65 536 sequences, solved on the device table by table, at points of the
    region: the optimal cost and the canonical minimiser are the ones the region's 42-entry table
    gives; and no sequence outside the table is below the bound U anywhere it was sampled.
    """
    from explicit_hybrid_mpc_amd import examples, sequences, engine
    mpc = examples.pwa4_mpc(N=8)
    R = region(mpc)
    table = sequences.PrefixTable(mpc, slots=2048)
    seqs, info = sequences.relevant_sequences(mpc, R[None], table=table)
    rng = np.random.default_rng(3)
    theta = np.vstack([R.mean(axis=0), rng.dirichlet(np.ones(9), size=3) @ R, R])
    npts = theta.shape[0]
    everything = list(itertools.product(range(4), repeat=8))
    J_all = np.empty((len(everything), npts))
    for k0 in range(0, len(everything), table.slots):
        chunk = everything[k0:k0 + table.slots]
        pairs = [q for q in chunk for _ in range(npts)]
        J = table.solve_points(pairs, np.tile(theta, (len(chunk), 1)))[0]
        J_all[k0:k0 + len(chunk)] = J.reshape(len(chunk), npts)
    table.close()
    in_table = np.array([s in set(seqs) for s in everything])
    U = info['upper_bound']
    assert np.all(J_all[~in_table] > U)                 # the pruning argument, sampled
    best = J_all.min(axis=0)
    assert np.all(best <= U + 1e-9)
    # canonical rule: lowest enumeration index within the tie tolerance
    tie = J_all <= best + 1e-6 * (1 + np.abs(best))
    first = tie.argmax(axis=0)
    assert all(in_table[k] for k in first)
    # V_R's canonical answer: the first sequence feasible at all 9 vertices
    at_vertices = np.isfinite(J_all[:, 4:]).all(axis=1)
    assert everything[int(at_vertices.argmax())] == info['first_feasible']
    assert info['first_feasible'] in seqs
    gp = engine.GpuProblem(mpc.restrict(seqs).compile(), 1., 1.)
    Jr, _, dr = gp.solve_pt(theta)
    d_vr = gp.v_r(R[None])[0]
    gp.close()
    assert seqs[int(d_vr[0])] == info['first_feasible']
    assert np.allclose(Jr, best, rtol=RTOL, atol=RTOL)
    assert [seqs[d] for d in dr] == [everything[k] for k in first]


REGION_VISITS = 14      # CPU visits of the regional partition (about 200 LPs per oracle call)


def test_partition_of_a_region_on_its_table():
    """
    The multi-commutation engine (csrc/ehm_hybrid.h) on a region's table -- 205 of the 65 536
    sequences, 40-column / 368-row LPs on the wide kernels -- against the CPU oracle on the same
    table: the part of the tree the CPU finishes in REGION_VISITS visits is the device's.
    """
    from explicit_hybrid_mpc_amd import examples, sequences, engine
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    mpc = examples.pwa4_mpc(N=8)
    R = region(mpc, 0.9, 0.04)
    seqs, info = sequences.relevant_sequences(mpc, R[None])
    assert 64 < len(seqs) <= 256
    sub = mpc.restrict(seqs)
    can = sub.compile()
    assert (can.n, can.m) == (40, 368)
    gp = engine.GpuProblem(can, 1., 1.)
    J = gp.solve_pt(R)[0]
    eps_a, eps_r = 1e-3 * float(J.max()), 2e-3
    gp.set_eps(eps_a, eps_r)
    flat = gp.partition(R[None], action='ecc', max_nodes=1 << 18)
    gp.close()
    used = len(set(flat.delta_idx[flat.delta_idx >= 0].tolist()))
    print('\nregional partition on %d sequences: %d nodes, %d leaves, depth %d, %d commutations '
          'used, %d LPs, %.3f s, margin %.2e'
          % (len(seqs), flat.n_nodes, flat.info['n_leaves'], flat.info['max_depth'], used,
             flat.info['lp_solves'], flat.info['device_seconds'], flat.info['min_margin']))
    assert flat.info['truncated'] == 0 and flat.n_nodes >= 3
    assert flat.info['min_margin'] > 1e-6
    assert abs(flat.info['volume_closed'] - helpers_volume(R)) <= 1e-9 * helpers_volume(R)
    orc = OracleCPU(sub, eps_a, eps_r)
    orc.memoize = True
    cpu = PartitionCPU(orc, max_nodes=REGION_VISITS)
    cpu.run([R], [''], 'ecc')
    assert cpu.min_margin > 1e-6
    loc = flat.locations([''])
    pos = {name: k for k, name in enumerate(loc)}
    decided = 0
    for name, ref in cpu.nodes.items():
        k = pos[name]                   # KeyError = the CPU split a node the device did not
        assert np.array_equal(flat.vertices[k], ref['vertices']), name
        if (not ref['leaf']) or ref['is_epsilon_suboptimal']:
            assert flat.is_leaf(k) == ref['leaf'], name
            assert bool(flat.flags[k] & 1) == ref['is_epsilon_suboptimal'], name
            assert np.array_equal(flat.deltas[flat.delta_idx[k]].astype(int),
                                  ref['commutation'].astype(int)), name
            assert np.allclose(flat.vertex_costs[k], ref['vertex_costs'], rtol=RTOL, atol=RTOL)
            decided += 1
    assert decided >= min(3, flat.n_nodes)


def helpers_volume(R):
    from math import factorial
    return abs(np.linalg.det(R[1:] - R[0])) / factorial(R.shape[1])


# ---------------------------------------------------------------------------------------------
# branch-and-bound oracles and the partition driver (explicit_hybrid_mpc_amd/bnb.py)
# ---------------------------------------------------------------------------------------------
def _walk_equal(branch, flat, n_min):
    """The Tree grown by bnb.grow against a FlatTree of the enumerating engine (one root)."""
    loc = flat.locations([''])
    pos = {name: k for k, name in enumerate(loc)}
    n = 0
    for node, name in branch.walk(''):
        k = pos[name]
        n += 1
        assert np.array_equal(node.data.vertices, flat.vertices[k]), name
        assert node.is_leaf() == flat.is_leaf(k), name
        assert node.data.is_epsilon_suboptimal == bool(flat.flags[k] & 1), name
        if flat.flags[k] & 2:
            assert np.array_equal(np.asarray(node.data.commutation).astype(int),
                                  flat.deltas[flat.delta_idx[k]].astype(int)), name
            assert np.allclose(node.data.vertex_costs, flat.vertex_costs[k],
                               rtol=RTOL, atol=RTOL), name
    assert n == flat.n_nodes and n >= n_min


@pytest.mark.parametrize('frontier', [False, True])
@pytest.mark.parametrize('frac,size,eps', [(0.6, 0.35, 0.05), (0.7, 0.25, 0.02)])
@pytest.mark.parametrize('handoff', [False, True])
def test_driver_with_search_oracles_grows_the_enumerating_engines_tree(handoff, frac, size, eps,
                                                                       frontier):
    """
    4 modes, N = 4: all 256 sequences fit the device engine, which is the reference here.  The
    driver of bnb.py -- branch-and-bound oracles at the top, region tables of at most 128
    sequences handed to the engine below -- must grow the same tree, node for node; so must the
    frontier-wide driver (bnb_frontier.grow_frontier: all open nodes' searches in shared launches).
    """
    from explicit_hybrid_mpc_amd import examples, engine, bnb, bnb_frontier
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    mpc = examples.pwa4_mpc()
    R = region(mpc, frac, size)
    gp = engine.GpuProblem(mpc.compile(), 1., 1.)
    J = gp.solve_pt(R)[0]
    eps_a, eps_r = eps * float(J.max()), eps
    gp.set_eps(eps_a, eps_r)
    flat = gp.partition(R[None], action='ecc', max_nodes=1 << 18)
    gp.close()
    assert flat.info['truncated'] == 0 and flat.info['min_margin'] > 1e-6
    orc = bnb.PrefixOracle(mpc, eps_a, eps_r, slots=1024)
    branch = Tree(NodeData(vertices=R.copy()))
    grow = bnb_frontier.grow_frontier if frontier else bnb.grow
    stats = grow(orc, branch, 'ecc', handoff=handoff, table_max=128)
    print('\nN=4 driver (handoff=%s, frontier-wide=%s): %d nodes; %s; oracle calls %s, %d prefixes '
          'expanded, %d LPs' % (handoff, frontier, flat.n_nodes, {k: v for k, v in stats.items() if k != 'table_sizes'},
             orc.calls, orc.n_expanded, orc.table.lp_solves))
    orc.close()
    assert not stats['truncated']
    if handoff:
        assert stats['handoffs'] >= 1 and max(stats['table_sizes']) <= 128
    _walk_equal(branch, flat, 15)


def test_search_oracles_equal_the_full_enumeration_at_65536_sequences():
    """
    N = 8 on a region whose table (2 734 sequences) does not fit the engine: bar_E, bar_D and
    V_R by branch and bound against the same problems ENUMERATED on the device (65 536 slack
    LPs, vertex feasibility of every candidate).
    """
    from explicit_hybrid_mpc_amd import examples, bnb, sequences
    mpc = examples.pwa4_mpc(N=8)
    R = region(mpc, 0.6, 0.05)
    table = sequences.PrefixTable(mpc, slots=2048)
    with pytest.raises(sequences.TableTooLarge):
        sequences.relevant_sequences(mpc, R[None], table=table)
    J0 = table.solve_min([()], R[None])[0]
    assert abs(J0) < 1e-7                              # the empty prefix has no state-cost rows
    orc = bnb.PrefixOracle(mpc, 1., 1., table=table)
    delta, vx = orc.V_R(R)
    V = np.array([v[1] for v in vx])
    eps_a, eps_r = 0.01 * float(V.max()), 0.01
    orc.eps_a, orc.eps_r = eps_a, eps_r
    table.set_eps(eps_a, eps_r)
    lp0, ex0 = table.lp_solves, orc.n_expanded
    closed = orc.bar_E_delta_R(R, V)
    star = orc.bar_D_delta_R(R, V, delta)
    print('\nN=8 region root: V_R -> %s; bar_E closed=%s; bar_D -> %s; %d prefixes expanded, '
          '%d LPs (enumeration: 65 536 x 2 + vertex problems)'
          % (orc.sequence_of(delta), closed, None if star[0] is None else
             orc.sequence_of(star[0]), orc.n_expanded - ex0, table.lp_solves - lp0))
    assert table.lp_solves - lp0 < 0.2 * 65536
    # the enumeration, 2 048 sequences per table load
    everything = list(itertools.product(range(4), repeat=8))
    t_all = np.empty(len(everything))
    a_all = np.empty((len(everything), 9))
    for k0 in range(0, len(everything), table.slots):
        chunk = everything[k0:k0 + table.slots]
        t, a = table.solve_slack(chunk, np.tile(R, (len(chunk), 1, 1)), np.tile(V, (len(chunk), 1)))
        ok = np.flatnonzero(t >= 0.)
        feas = table.feasible_at_all([chunk[k] for k in ok], R)
        t[ok[~feas]] = -np.inf
        t_all[k0:k0 + len(chunk)] = t
        a_all[k0:k0 + len(chunk)] = a
    assert closed == (not (t_all.max() >= 0.))
    if t_all.max() >= 0.:
        t_max = t_all.max()
        first = int(np.argmax(t_all >= t_max - 1e-6 * (1 + abs(t_max))))
        if everything[first] == orc.sequence_of(delta):
            assert star[0] is None
        else:
            assert orc.sequence_of(star[0]) == everything[first]
            assert np.allclose(star[1], a_all[first] @ R, atol=1e-6)
    else:
        assert star[0] is None
    # V_R: the first sequence feasible at all 9 vertices
    first_seq = orc.sequence_of(delta)
    before = [q for q in everything[:everything.index(first_seq)]]
    if before:
        assert not table.feasible_at_all(before[-512:], R).any()
    table.close()


def _grow_and_check(mpc, R, eps, max_visits, label, frontier=False):
    import time
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier, tools
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=4096)
    J = [orc.P_theta(v)[2] for v in R[:2]]
    eps_a = eps[0] * max(J)
    orc.eps_a, orc.eps_r = eps_a, eps[1]
    orc.table.set_eps(eps_a, eps[1])
    branch = Tree(NodeData(vertices=R.copy()))
    t0 = time.time()
    grow = bnb_frontier.grow_frontier if frontier else bnb.grow
    stats = grow(orc, branch, 'ecc', max_visits=max_visits)
    seconds = time.time() - t0
    leaves = list(branch.leaves())
    closed = [n for n, _ in leaves if n.data.is_epsilon_suboptimal]
    print('\n%s: %d nodes, %d leaves (%d closed) in %.1f s; %s; oracle calls %s; %d prefixes '
          'expanded, %d LPs, %d table blocks loaded'
          % (label, sum(1 for _ in branch.walk()), len(leaves), len(closed), seconds, stats,
             orc.calls, orc.n_expanded, orc.table.lp_solves, orc.table.blocks_loaded))
    orc.close()
    vol = sum(tools.simplex_volume(n.data.vertices) for n, _ in leaves)
    assert abs(vol - tools.simplex_volume(R)) <= 1e-9 * vol
    for n in closed:
        assert len(n.data.vertex_costs) == 9 and np.all(np.isfinite(n.data.vertex_costs))
        assert len(n.data.commutation) == 32
    return stats, leaves, closed


def test_driver_on_a_region_whose_table_does_not_fit():
    """
    N = 8, a region with 397 relevant sequences: the top of its tree is grown with the search
    oracles on the host, the node whose table fits (200 sequences) goes to the device engine.
    """
    from explicit_hybrid_mpc_amd import examples
    mpc = examples.pwa4_mpc(N=8)
    stats, leaves, closed = _grow_and_check(mpc, region(mpc, 0.9, 0.05), (0.005, 0.005), 40,
                                            'N=8 region')
    assert not stats['truncated'] and len(closed) == len(leaves)
    assert stats['host_visits'] >= 1 and stats['handoffs'] >= 1
    assert stats['tables_too_large'] >= 1 and max(stats['table_sizes']) <= 256


def test_driver_at_the_top_of_the_config5_tree():
    """
    N = 8 from a simplex that spans the whole box Theta (a Kuhn simplex: 1/8! of it), with
    BASELINE.json's tolerances (abs_frac 0.5, eps_r 1.0): a budget of host visits with the
    search oracles -- ecc splits until a sequence is feasible at all 9 vertices, then lcss.
    """
    from explicit_hybrid_mpc_amd import examples
    mpc = examples.pwa4_mpc(N=8)
    half = examples.theta_box(mpc)
    R = np.array([-half + 2 * half * (np.arange(8) < k) for k in range(9)])
    stats, leaves, closed = _grow_and_check(mpc, R, (0.5, 1.0), 40, 'N=8 whole-box simplex')
    assert stats['host_visits'] == 40 and stats['truncated']
    assert len(closed) >= 3 and len(leaves) >= 10


def test_frontier_wide_driver_at_the_top_of_the_config5_tree():
    """The run of the previous test with ALL pending nodes visited together
    (bnb_frontier.grow_frontier): a hundred times the visits in comparable time."""
    from explicit_hybrid_mpc_amd import examples
    mpc = examples.pwa4_mpc(N=8)
    half = examples.theta_box(mpc)
    R = np.array([-half + 2 * half * (np.arange(8) < k) for k in range(9)])
    stats, leaves, closed = _grow_and_check(mpc, R, (0.5, 1.0), 4000,
                                            'N=8 whole-box simplex, frontier-wide', frontier=True)
    assert stats['host_visits'] == 4000 and stats['truncated'] and stats['rounds'] <= 20
    assert len(leaves) >= 3000                  # breadth first: the ecc phase of the whole cell


@pytest.mark.parametrize('tolerances', ['loose', 'stated'])
def test_whole_cell_partition_delivers_the_guarantee(tolerances):
    """
    N = 8, 65 536 sequences: a cell spanning the whole box partitioned to COMPLETION by the
    frontier-wide driver, then the explicit law's guarantee (the point of the whole exercise,
    lib/oracle.py:89-97) checked at random parameters: the cost interpolated in the leaf that
    holds theta exceeds the mixed-integer optimum (branch and bound at theta) by less than
    max(eps_a, eps_r V*).  'loose': eps_a = half the largest vertex cost, eps_r = 1 (every cell
    closes with the sequence V_R finds); 'stated': BASELINE.json configs[4]'s eps_r = 1e-3 with
    eps_a by the reference's rule (lib/examples.py:42-46, abs_frac 0.2 = bench.CONFIG5) -- the
    suboptimality tests are real branch-and-bound refutations, cells split in lcss, and sampled
    nodes are grown again by the CPU statement of the search (HiGHS on the uncondensed
    relaxations): same fate, same children, same commutations.
    """
    import time
    from explicit_hybrid_mpc_amd import examples, bnb, bnb_frontier, tools
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    mpc = examples.pwa4_mpc(N=8)
    half = examples.theta_box(mpc)
    R = np.array([-half + 2 * half * (np.arange(8) < k) for k in range(9)])
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=8192)
    if tolerances == 'loose':
        Jm = max(orc.P_theta(v)[2] for v in R)
        eps_a, eps_r = 0.5 * Jm, 1.0
        grow = dict(round_cap=16384)
    else:
        V = examples.box_vertices(half)
        eps_a = float(np.max([j for _, _, j in bnb_frontier.p_theta_many(orc, 0.2 * V)]))
        eps_r = 1e-3
        grow = dict(round_cap=2048, order='lcss-first', table_backoff=True)
    orc.eps_a, orc.eps_r = eps_a, eps_r
    orc.table.set_eps(eps_a, eps_r)
    root = Tree(NodeData(vertices=R.copy()))
    created = {}
    t0 = time.time()
    stats = bnb_frontier.grow_frontier(orc, root, 'ecc', created_log=created, **grow)
    seconds = time.time() - t0
    leaves = list(root.leaves())
    print('\nN=8 whole-box cell to completion (%s: eps_a %.4g, eps_r %g): %d nodes, %d regions in '
          '%.1f s; %s; %d LPs; calls %s'
          % (tolerances, eps_a, eps_r, sum(1 for _ in root.walk()), len(leaves), seconds,
             {k: v for k, v in stats.items() if k != 'table_sizes'}, orc.table.lp_solves,
             orc.calls))
    assert not stats['truncated'] and len(leaves) > 20000
    assert all(n.data.is_epsilon_suboptimal for n, _ in leaves)
    assert stats['regions'] == len(leaves)
    vol = sum(tools.simplex_volume(n.data.vertices) for n, _ in leaves[::50])
    assert vol > 0
    if tolerances == 'stated':
        assert orc.calls['bar_D'] >= 50             # cells that did not close at once: lcss splits

    def bary(S, th):
        return np.linalg.solve(np.vstack([S.T, np.ones(9)]), np.append(th, 1.))
    rng = np.random.default_rng(5)
    thetas = rng.dirichlet(np.ones(9), size=400) @ R
    V_bar, u_leaf = [], []
    for th in thetas:
        node = root
        while not node.is_leaf():
            node = node.left if bary(node.left.data.vertices, th).min() >= -1e-12 else node.right
        w = bary(node.data.vertices, th)
        assert w.min() >= -1e-9
        V_bar.append(float(w @ node.data.vertex_costs))
        u_leaf.append(w @ node.data.vertex_inputs)
    t0 = time.time()
    sol = bnb_frontier.p_theta_many(orc, thetas)       # the mixed-integer optimum at each of them
    V_star = np.array([s[2] for s in sol])
    V_bar = np.array(V_bar)
    assert np.all(V_star <= V_bar + 1e-9)                # the leaf's sequence is feasible there
    gap = V_bar - V_star - np.maximum(eps_a, eps_r * V_star)
    print('   guarantee at %d random parameters (P_theta by search, %.1f s): worst V_bar - V* - '
          'max(eps_a, eps_r V*) = %.3g' % (len(thetas), time.time() - t0, gap.max()))
    assert gap.max() < 1e-7
    # the lockstep search is the one-parameter search
    one = orc.P_theta(thetas[0])
    assert np.array_equal(one[1], sol[0][1]) and abs(one[2] - sol[0][2]) <= 1e-12
    # the consumer: the explicit law evaluated on the device from this tree (lib/mpc_library.py:
    # 662-792) returns the inputs interpolated in those leaves
    from explicit_hybrid_mpc_amd import explicit
    law = explicit.ExplicitMPC(root, orc)
    U = law.evaluate(thetas)
    law.close()
    assert np.allclose(U, np.array(u_leaf), rtol=1e-9, atol=1e-9)
    orc.close()
    if tolerances == 'stated':
        _subforests_against_the_cpu_search(mpc, root, eps_a, eps_r, created)


def _subforests_against_the_cpu_search(mpc, root, eps_a, eps_r, created, n_split=26, n_closed=14):
    """Sampled lcss nodes of the device-grown tree, visited again by the search driver on the CPU
    statement of the table (oracle/prefix_bb.CpuPrefixTable: HiGHS on the UNCONDENSED prefix
    relaxations, 65 536 sequences): closed where the device closed, split where it split -- the
    same children (bit-identical vertices), EVERY child with the commutation and the vertex
    costs (1e-7) the device created it with (``created``: the driver's log of what each child
    was born with -- a later visit of a child may adopt another commutation in place,
    lib/worker.py:396-401, and its node record then shows that one)."""
    import time
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    from oracle import geometry, prefix_bb

    def split_batch(Rs):
        S1, S2, ij = [], [], []
        for r in Rs:
            a, b, e = geometry.split_along_longest_edge(r)
            S1.append(a), S2.append(b), ij.append(e)
        return np.array(S1), np.array(S2), np.array(ij)
    with_data = [(nd, loc) for nd, loc in root.walk()
                 if getattr(nd.data, 'commutation', None) is not None]
    # a node created by an lcss split holds its parent's commutation and has a parent with data
    split = [(nd, loc) for nd, loc in with_data if not nd.is_leaf()]
    closed = [(nd, loc) for nd, loc in with_data if nd.is_leaf()]
    rng = np.random.default_rng(9)
    picks = [split[i] for i in rng.choice(len(split), size=min(n_split, len(split)), replace=False)] + \
        [closed[i] for i in rng.choice(len(closed), size=n_closed, replace=False)]
    cpu = bnb.PrefixOracle(mpc, eps_a, eps_r, table=prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r))
    t0 = time.time()
    n_same_split = n_same_closed = n_children = n_swapped_later = 0
    other = []
    for nd, loc in picks:
        # the node's record is the one its LAST lcss visit saw (after any swap in place): the
        # visit that closed it or split it
        rec = NodeData(vertices=nd.data.vertices.copy(), commutation=nd.data.commutation.copy(),
                       vertex_costs=nd.data.vertex_costs.copy(),
                       vertex_inputs=nd.data.vertex_inputs.copy())
        again = Tree(rec)
        bnb_frontier.grow_frontier(cpu, again, 'lcss', handoff=False, split_batch=split_batch,
                                   max_visits=1)
        assert again.is_leaf() == nd.is_leaf(), loc
        if nd.is_leaf():
            assert again.data.is_epsilon_suboptimal
            n_same_closed += 1
            continue
        for mine, theirs in ((again.left, nd.left), (again.right, nd.right)):
            assert np.array_equal(mine.data.vertices, theirs.data.vertices), loc
            born_delta, born_costs = created[id(theirs)]
            if not np.array_equal(np.asarray(mine.data.commutation).astype(int),
                                  np.asarray(born_delta).astype(int)):
                other.append((loc, 'commutation', float(np.max(np.abs(
                    np.asarray(mine.data.vertex_costs) - born_costs)))))
            elif not np.allclose(mine.data.vertex_costs, born_costs, rtol=1e-7, atol=1e-7):
                other.append((loc, 'costs', float(np.max(np.abs(
                    np.asarray(mine.data.vertex_costs) - born_costs)))))
            n_children += 1
            n_swapped_later += int(not np.array_equal(
                np.asarray(theirs.data.commutation).astype(int),
                np.asarray(born_delta).astype(int)))
        n_same_split += 1
    print('   %d split and %d closed lcss nodes visited again by the CPU search (%d HiGHS LPs, '
          '%.0f s): same fate, all %d children born with the same commutation and costs (%d of '
          'them adopted another one in a later visit of their own)'
          % (n_same_split, n_same_closed, cpu.table.lp_solves, time.time() - t0, n_children,
             n_swapped_later))
    assert not other, other
    assert n_same_closed == n_closed and n_same_split >= min(n_split, len(split))
    assert n_same_split + n_same_closed >= 40
