"""
The branch-and-bound oracles and the partition driver of explicit_hybrid_mpc_amd/bnb.py (host
logic), run on a CPU stand-in for the device table (oracle/prefix_bb.CpuPrefixTable: the same
pair solvers, HiGHS) against the ENUMERATING CPU oracle: same canonical answers, same tree.
"""

import pytest
import itertools

import numpy as np

from explicit_hybrid_mpc_amd import bnb, examples
from explicit_hybrid_mpc_amd.tree import Tree, NodeData
from oracle import prefix_bb, geometry
from oracle.oracle_cpu import OracleCPU
from oracle.partition_cpu import PartitionCPU
from tests import helpers


def _same_bar_d(r1, r2):
    assert (r1[0] is None) == (r2[0] is None)
    if r1[0] is not None:
        assert np.array_equal(r1[0].astype(int), r2[0].astype(int))
        assert np.allclose(r1[1], r2[1], atol=1e-7) and r1[3] == r2[3]
        assert np.allclose([v[1] for v in r1[2]], [v[1] for v in r2[2]], atol=1e-8)
    return r1[0] is not None


def test_oracles_equal_enumeration_on_a_small_instance():
    mpc = helpers.make_instance('pwa_small', 0)          # 2 modes, N = 3
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    full = OracleCPU(mpc, eps_a, 0.2)
    bo = bnb.PrefixOracle(mpc, eps_a, 0.2, table=prefix_bb.CpuPrefixTable(mpc))
    rng = np.random.default_rng(0)
    half = examples.theta_box(mpc)
    for _ in range(20):
        th = rng.uniform(-1, 1, 2) * half
        a, b = full.P_theta(th), bo.P_theta(th)
        assert (a[1] is None) == (b[1] is None)
        assert full.P_theta(th, check_feasibility=True) == bo.P_theta(th, check_feasibility=True)
        if a[1] is not None:
            assert abs(a[2] - b[2]) < 1e-8 and np.array_equal(a[1], b[1])
            assert np.allclose(a[0], b[0], atol=1e-7)
            c = bo.P_theta_delta(th, a[1])
            assert abs(c[1] - a[2]) < 1e-8 and bo.P_theta_delta(th, a[1], check_feasibility=True)
    n_open = n_swap = 0
    for R in helpers.random_simplices(mpc, rng, 60, scale_lo=-1.):
        d1, vx1 = full.V_R(R)
        d2, vx2 = bo.V_R(R)
        assert (d1 is None) == (d2 is None)
        if d1 is None:
            continue
        assert np.array_equal(d1, d2)
        V = np.array([v[1] for v in vx1])
        assert np.allclose(V, [v[1] for v in vx2], atol=1e-8)
        e1, e2 = full.bar_E_delta_R(R, V), bo.bar_E_delta_R(R, V)
        assert e1 == e2
        n_open += not e1
        n_swap += _same_bar_d(full.bar_D_delta_R(R, V, d1), bo.bar_D_delta_R(R, V, d1))
    assert n_open >= 3 and n_swap >= 1


def test_driver_grows_the_tree_of_the_enumerating_partition():
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    cpu = PartitionCPU(OracleCPU(mpc, eps_a, 0.2))
    cpu.run(roots, locs, 'ecc')
    bo = bnb.PrefixOracle(mpc, eps_a, 0.2, table=prefix_bb.CpuPrefixTable(mpc))
    n = 0
    for R, loc in zip(roots, locs):
        branch = Tree(NodeData(vertices=np.array(R)))
        stats = bnb.grow(bo, branch, 'ecc', handoff=False, split=geometry.split_along_longest_edge)
        assert not stats['truncated'] and stats['host_visits'] > 0
        for node, name in branch.walk(loc):
            ref = cpu.nodes[name]
            n += 1
            assert np.array_equal(node.data.vertices, ref['vertices'])
            assert node.is_leaf() == ref['leaf']
            assert node.data.is_epsilon_suboptimal == ref['is_epsilon_suboptimal']
            if ref['commutation'] is None:           # split by ecc: no record (lib/tree.py:34-39)
                assert not hasattr(node.data, 'commutation')
                continue
            assert np.array_equal(node.data.commutation.astype(int),
                                  ref['commutation'].astype(int))
            assert np.allclose(node.data.vertex_costs, ref['vertex_costs'], atol=1e-8)
    assert n == len(cpu.nodes) and n > 20
    # a budget of visits leaves an incomplete tree and says so
    branch = Tree(NodeData(vertices=np.array(roots[0])))
    assert bnb.grow(bo, branch, 'ecc', handoff=False, max_visits=2,
                    split=geometry.split_along_longest_edge)['truncated']


def test_oracles_equal_enumeration_at_256_sequences():
    """4 modes, N = 4 (examples.pwa4_mpc): the searches look at a fraction of the 256 sequences
    and return the enumeration's canonical answers."""
    mpc = examples.pwa4_mpc()
    half = examples.theta_box(mpc)
    E = np.vstack([np.zeros(8), np.eye(8)]) - 1. / 9.
    R = 0.7 * examples.box_vertices(half)[37] + 0.25 * half * E
    table = prefix_bb.CpuPrefixTable(mpc)
    J0 = table.solve_points([(3, 0, 0, 0)], R[:1])[0][0]
    eps_a, eps_r = 0.02 * J0, 0.02
    full = OracleCPU(mpc, eps_a, eps_r)
    full.memoize = True
    bo = bnb.PrefixOracle(mpc, eps_a, eps_r, table=table)
    th = R.mean(axis=0)
    a, b = full.P_theta(th), bo.P_theta(th)
    assert abs(a[2] - b[2]) < 1e-8 and np.array_equal(a[1], b[1])
    d1, vx1 = full.V_R(R)
    d2, vx2 = bo.V_R(R)
    assert np.array_equal(d1, d2)
    V = np.array([v[1] for v in vx1])
    assert full.bar_E_delta_R(R, V) == bo.bar_E_delta_R(R, V) == False
    before = table.lp_solves
    assert _same_bar_d(full.bar_D_delta_R(R, V, d1), bo.bar_D_delta_R(R, V, d1))
    assert table.lp_solves - before < 0.5 * 256 * 10     # enumeration: 256 x (9 + 1) problems


def _host_split_batch(R):
    out = [geometry.split_along_longest_edge(r) for r in R]
    return (np.array([o[0] for o in out]), np.array([o[1] for o in out]),
            np.array([o[2] for o in out], dtype=np.int32))


def test_frontier_wide_searches_equal_the_one_node_searches():
    """bnb_frontier: every node's search step in one shared call -- same answers, same tree."""
    from explicit_hybrid_mpc_amd import bnb_frontier
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    table = prefix_bb.CpuPrefixTable(mpc)
    bo = bnb.PrefixOracle(mpc, eps_a, 0.2, table=table)
    rng = np.random.default_rng(2)
    Rs = [np.array(R) for R in helpers.random_simplices(mpc, rng, 25, scale_lo=-1.)]
    # P_theta at many parameters in lockstep
    half = examples.theta_box(mpc)
    thetas = rng.uniform(-1, 1, (12, 2)) * half
    for th, (u, d, J) in zip(thetas, bnb_frontier.p_theta_many(bo, thetas)):
        a = bo.P_theta(th)
        assert (a[1] is None) == (d is None)
        if d is not None:
            assert np.array_equal(a[1], d) and a[2] == J and np.array_equal(a[0], u)
    # phase two of the lockstep form walks on the values phase one solved (no second solve of a
    # child): fewer problems than the one-parameter statement, the same expansions
    counts = []
    for many in (True, False):
        t2 = prefix_bb.CpuPrefixTable(mpc)
        b2 = bnb.PrefixOracle(mpc, eps_a, 0.2, table=t2)
        if many:
            bnb_frontier.p_theta_many(b2, thetas)
        else:
            [b2.P_theta(th) for th in thetas]
        counts.append((t2.lp_solves, b2.n_expanded))
    assert counts[0][1] == counts[1][1] and counts[0][0] < 0.8 * counts[1][0], counts
    one = [table.first_feasible(R) for R in Rs]
    assert bnb_frontier.first_feasible_many(table, Rs) == one
    assert any(s is not None for s in one)
    have = [k for k, s in enumerate(one) if s is not None]
    Rh = [Rs[k] for k in have]
    Vh = [np.array([v[1] for v in bo._vertex_solves(Rs[k], one[k])]) for k in have]
    dh = [bo.delta_of(one[k]) for k in have]
    closed, margins = bnb_frontier.bar_e_many(bo, Rh, Vh)
    for R, V, c, m in zip(Rh, Vh, closed, margins):
        assert bo.bar_E_delta_R(R, V) == c and bo.last_margin == m
    many = bnb_frontier.bar_d_many(bo, Rh, Vh, dh)
    n_star = 0
    for R, V, d, got in zip(Rh, Vh, dh, many):
        n_star += _same_bar_d(bo.bar_D_delta_R(R, V, d), got)
    assert n_star >= 1
    # the driver: the tree of bnb.grow (and so of the enumerating partition)
    roots, locs = helpers.roots_of(mpc)
    for R in roots:
        a = Tree(NodeData(vertices=np.array(R)))
        b = Tree(NodeData(vertices=np.array(R)))
        bnb.grow(bo, a, 'ecc', handoff=False, split=geometry.split_along_longest_edge)
        stats = bnb_frontier.grow_frontier(bo, b, 'ecc', handoff=False,
                                           split_batch=_host_split_batch, round_cap=7)
        assert not stats['truncated'] and stats['rounds'] > 2
        na = {loc: nd for nd, loc in a.walk()}
        nb = {loc: nd for nd, loc in b.walk()}
        assert set(na) == set(nb)
        for loc, x in na.items():
            y = nb[loc]
            assert np.array_equal(x.data.vertices, y.data.vertices)
            assert x.is_leaf() == y.is_leaf()
            assert x.data.is_epsilon_suboptimal == y.data.is_epsilon_suboptimal
            assert hasattr(x.data, 'commutation') == hasattr(y.data, 'commutation')
            if hasattr(x.data, 'commutation'):
                assert np.array_equal(x.data.commutation, y.data.commutation)
                assert np.allclose(x.data.vertex_costs, y.data.vertex_costs, atol=1e-12)
                assert np.allclose(x.data.vertex_inputs, y.data.vertex_inputs, atol=1e-9)
    # a budget of visits is respected
    c = Tree(NodeData(vertices=np.array(roots[0])))
    stats = bnb_frontier.grow_frontier(bo, c, 'ecc', handoff=False, max_visits=3,
                                       split_batch=_host_split_batch)
    assert stats['truncated'] and stats['host_visits'] == 3


def test_region_tables_of_many_nodes():
    """bnb_frontier.region_tables_many (bound = the node's own commutation's vertex costs) against
    the one-region search with the same bound, and its soundness: the enumerating oracle's bar_D
    answer on the node lies in the table."""
    from explicit_hybrid_mpc_amd import bnb_frontier
    mpc = examples.pwa4_mpc()                            # 4 modes, N = 4: 256 sequences
    half = examples.theta_box(mpc)
    E = np.vstack([np.zeros(8), np.eye(8)]) - 1. / 9.
    V = examples.box_vertices(half)
    table = prefix_bb.CpuPrefixTable(mpc)
    bo = bnb.PrefixOracle(mpc, 0.01, 0.02, table=table)
    Rs = [0.8 * V[37] + 0.08 * half * E, 0.7 * V[100] + 0.05 * half * E]
    comms, costs = [], []
    for R in Rs:
        delta, vx = bo.V_R(R)
        comms.append(delta)
        costs.append(np.array([v[1] for v in vx]))
    tables = bnb_frontier.region_tables_many(bo, Rs, comms, [float(c.max()) for c in costs], 256)
    full = OracleCPU(mpc, 0.01, 0.02)
    for R, delta, c, seqs in zip(Rs, comms, costs, tables):
        assert seqs is not None and bo.sequence_of(delta) in seqs and len(seqs) < 256
        # the same levels one region at a time
        U = float(c.max())
        alive = [()]
        for _ in range(mpc.N):
            cand = [q + (i,) for q in alive for i in range(4)]
            alive = [q for q, v in zip(cand, table.min_cost_on(cand, R[None]))
                     if v <= U + 1e-6 * (1 + abs(U))]
        assert sorted(set(alive) | {bo.sequence_of(delta)}) == seqs
        star = full.bar_D_delta_R(R, c, delta)[0]
        if star is not None:
            assert bo.sequence_of(star) in seqs
    # a table limit that is too small is reported as None, not cut
    assert bnb_frontier.region_tables_many(bo, Rs[:1], comms[:1], [float(costs[0].max())], 2) \
        == [None]


class _StubDeviceProblem:
    """Stands in for engine.GpuProblem under sequences.PrefixTable: keeps the blocks the table
    writes into its slots and solves the point problems on them with HiGHS -- the CONDENSED
    form  min c'z  s.t.  G z <= w + S theta  (phase one: min tau, G z - tau <= w + S theta)."""

    def __init__(self, can, eps_a, eps_r, device=0):
        self.can = can
        self.eps_a, self.eps_r = eps_a, eps_r
        self.G, self.w, self.S = can.G.copy(), can.w.copy(), can.S.copy()
        self.updates = []

    def set_eps(self, eps_a, eps_r):
        self.eps_a, self.eps_r = eps_a, eps_r

    def close(self):
        pass

    def simplex_idx(self, R, slot, mode, Vbar=None):
        """mode 0: min cost over the simplex; 1: suboptimality-test slack; 2: phase one."""
        from scipy.optimize import linprog
        R = np.asarray(R, dtype=np.float64)
        n, na = self.can.n, R.shape[1]
        obj = np.zeros(R.shape[0])
        alpha = np.zeros((R.shape[0], na))
        for k, s in enumerate(slot):
            G, w, S = self.G[s], self.w[s], self.S[s]
            m = G.shape[0]
            extra = 0 if mode == 0 else 1
            A = np.zeros((m + (2 if mode == 1 else 0), n + na + extra))
            A[:m, :n], A[:m, n:n + na] = G, -S @ R[k].T
            b = np.concatenate([w, np.zeros(2 if mode == 1 else 0)])
            c = np.zeros(n + na + extra)
            bounds = [(None, None)] * n + [(0., None)] * na
            if mode == 0:
                c[:n] = self.can.c
            elif mode == 2:
                A[:m, -1] = -1.
                c[-1] = 1.
                bounds.append((-1., None))
            else:                                   # t <= Vbar'a - c'z - eps_a, <= Vbar'a - (1+eps_r) c'z
                for r, scale, shift in ((m, 1., self.eps_a), (m + 1, 1. + self.eps_r, 0.)):
                    A[r, :n], A[r, n:n + na], A[r, -1] = scale * self.can.c, -Vbar[k], 1.
                    b[r] = -shift
                c[-1] = -1.
                bounds.append((None, None))
            Aeq = np.zeros((1, A.shape[1]))
            Aeq[0, n:n + na] = 1.
            res = linprog(c, A_ub=A, b_ub=b, A_eq=Aeq, b_eq=[1.], bounds=bounds, method='highs')
            assert res.status == 0, (mode, res.message)
            obj[k] = -res.fun if mode == 1 else res.fun
            alpha[k] = res.x[n:n + na]
        return obj, alpha, np.zeros(R.shape[0], dtype=np.int32)

    def update_blocks(self, first, G, w, S):
        n = G.shape[0]
        assert 0 <= first and first + n <= self.G.shape[0]
        self.G[first:first + n], self.w[first:first + n], self.S[first:first + n] = G, w, S
        self.updates.append((first, n))

    def point_idx(self, theta, slot, feas=False):
        from scipy.optimize import linprog
        theta = np.atleast_2d(theta)
        J = np.zeros(theta.shape[0])
        u0 = np.zeros((theta.shape[0], self.can.n_u))
        for k, (th, s) in enumerate(zip(theta, slot)):
            G, rhs = self.G[s], self.w[s] + self.S[s] @ th
            if feas:
                A = np.hstack([G, -np.ones((G.shape[0], 1))])
                c = np.zeros(A.shape[1])
                c[-1] = 1.
                res = linprog(c, A_ub=A, b_ub=rhs, bounds=[(None, None)] * G.shape[1] + [(-1., None)],
                              method='highs')
                J[k] = res.fun
            else:
                res = linprog(self.can.c, A_ub=G, b_ub=rhs, bounds=(None, None), method='highs')
                J[k] = res.fun
                u0[k] = res.x[:self.can.n_u]
        return J, u0, np.zeros(theta.shape[0], dtype=np.int32)


def test_device_table_bookkeeping_on_a_stub(monkeypatch):
    """
    sequences.PrefixTable itself (slots, the prefix -> slot map, chunking over more prefixes
    than slots, index-array pairs, the feasibility memo with midpoint inference) on a stub of
    the device problem that solves the CONDENSED blocks with HiGHS: the condensed prefix
    relaxations (PWAMPC.condense_prefix) give the optima of the uncondensed ones
    (oracle/prefix_bb.PrefixModel), whatever the slot traffic.
    """
    from explicit_hybrid_mpc_amd import sequences
    monkeypatch.setattr(sequences.engine, 'GpuProblem', _StubDeviceProblem)
    mpc = helpers.make_instance('pwa_small', 0)                  # 2 modes, N = 3
    table = sequences.PrefixTable(mpc, slots=4)
    ref = prefix_bb.CpuPrefixTable(mpc)
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(3)
    prefixes = [(), (0,), (1,), (0, 1), (1, 0), (1, 1, 0), (0, 0, 0), (1, 0, 1), (0, 1, 1)]
    pairs = [prefixes[k] for k in rng.integers(0, len(prefixes), 40)]
    thetas = rng.uniform(-1, 1, (40, 2)) * half
    J, u0 = table.solve_points(pairs, thetas)
    Jr, ur = ref.solve_points(pairs, thetas)
    assert np.array_equal(np.isfinite(J), np.isfinite(Jr)) and np.isfinite(J).sum() >= 10
    fin = np.isfinite(J)
    assert np.allclose(J[fin], Jr[fin], rtol=1e-8, atol=1e-9)
    # nine prefixes through four slots: several table loads, never more than four blocks at once
    assert len(table.gp.updates) >= 3 and all(n <= 4 for _, n in table.gp.updates)
    assert table.blocks_loaded >= len(set(pairs))
    # the index-array form is the tuple form
    uniq = sorted(set(pairs))
    idx = np.array([uniq.index(q) for q in pairs])
    J2, _ = table.solve_points_idx(uniq, idx, thetas)
    assert np.array_equal(np.isfinite(J2), fin) and np.allclose(J2[fin], J[fin], atol=1e-12)
    # feasibility memo: the second question costs nothing; a midpoint of two feasible points is
    # taken as feasible without an LP
    R = np.array(helpers.random_simplices(mpc, rng, 1, scale_lo=-2.)[0])
    seq = table.first_feasible(R)
    assert seq == ref.first_feasible(R)
    before = table.lp_solves
    assert table.first_feasible(R) == seq and table.lp_solves == before
    if seq is not None:
        mid = 0.5 * (R[0] + R[1])
        table.register_midpoints([mid], [R[0]], [R[1]])
        assert table.feasible_at_all([seq], mid[None])[0] and table.lp_solves == before
    # a full memo is dropped and rebuilt: same answers
    table.FEAS_MEMO_LIMIT = -1
    assert table.first_feasible(R) == seq and table.lp_solves > before
    assert table.search_counts()[0] > 0 and table._feas_n > 0
    table.close()


def test_searches_on_the_device_table_class_with_a_stub_device(monkeypatch):
    """
    The code the GPU tests run -- sequences.PrefixTable under bnb.PrefixOracle, the region-table
    search and the frontier-wide driver -- with the device problem replaced by the HiGHS stub on
    the CONDENSED blocks: same answers as on the uncondensed stand-in, same tree as the
    enumerating CPU partition.
    """
    from explicit_hybrid_mpc_amd import sequences, bnb_frontier
    monkeypatch.setattr(sequences.engine, 'GpuProblem', _StubDeviceProblem)
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    dev = bnb.PrefixOracle(mpc, eps_a, 0.2, slots=6)             # creates a sequences.PrefixTable
    assert isinstance(dev.table, sequences.PrefixTable)
    ref = bnb.PrefixOracle(mpc, eps_a, 0.2, table=prefix_bb.CpuPrefixTable(mpc))
    rng = np.random.default_rng(11)
    n_star = n_tables = 0
    for R in helpers.random_simplices(mpc, rng, 10, scale_lo=-1.):
        R = np.array(R)
        d1, vx1 = ref.V_R(R)
        d2, vx2 = dev.V_R(R)
        assert (d1 is None) == (d2 is None)
        if d1 is None:
            continue
        assert np.array_equal(d1, d2)
        V = np.array([v[1] for v in vx1])
        assert np.allclose(V, [v[1] for v in vx2], rtol=1e-8, atol=1e-9)
        assert ref.bar_E_delta_R(R, V) == dev.bar_E_delta_R(R, V)
        a, b = ref.bar_D_delta_R(R, V, d1), dev.bar_D_delta_R(R, V, d1)
        assert (a[0] is None) == (b[0] is None)
        if a[0] is not None:
            n_star += 1
            assert np.array_equal(a[0], b[0]) and np.allclose(a[1], b[1], atol=1e-6) and a[3] == b[3]
        outcome = []
        for table in (dev.table, ref.table):
            try:
                outcome.append(sequences.relevant_sequences(mpc, R[None], table=table)[0])
            except ValueError as e:              # NoIncumbent / TableTooLarge: on both or neither
                outcome.append(type(e).__name__)
        assert outcome[0] == outcome[1]
        n_tables += not isinstance(outcome[0], str)
    assert n_star >= 1 and n_tables >= 1
    th = rng.uniform(-0.8, 0.8, (5, 2)) * examples.theta_box(mpc)
    for (u1, dd1, J1), (u2, dd2, J2) in zip(bnb_frontier.p_theta_many(dev, th),
                                            bnb_frontier.p_theta_many(ref, th)):
        assert (dd1 is None) == (dd2 is None)
        if dd1 is not None:
            assert np.array_equal(dd1, dd2) and abs(J1 - J2) <= 1e-8 * (1 + abs(J2))
    # the frontier-wide driver on the device-table class: the enumerating CPU partition's tree
    roots, locs = helpers.roots_of(mpc)
    cpu = PartitionCPU(OracleCPU(mpc, eps_a, 0.2))
    cpu.run(roots, locs, 'ecc')
    trees = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    stats = bnb_frontier.grow_frontier(dev, trees, 'ecc', handoff=False,
                                       split_batch=_host_split_batch)
    assert not stats['truncated']
    n = 0
    for t, loc0 in zip(trees, locs):
        for nd, loc in t.walk(loc0):
            r = cpu.nodes[loc]
            n += 1
            assert nd.is_leaf() == r['leaf']
            assert nd.data.is_epsilon_suboptimal == r['is_epsilon_suboptimal']
            if r['commutation'] is not None:
                assert np.array_equal(nd.data.commutation.astype(int), r['commutation'].astype(int))
                assert np.allclose(nd.data.vertex_costs, r['vertex_costs'], rtol=1e-7, atol=1e-8)
    assert n == len(cpu.nodes)
    dev.close()


class _StallingStub(_StubDeviceProblem):
    """The stub with a solver that reports chosen problems as stalled: status word
    1 | (decade << 8) of csrc/ehm_dev.h -- stalled at merit <= 10^decade."""
    stall_points = stall_phase_one = stall_slack = False
    infeasible_sliver = False
    stall_decade = 8                    # residuals / gap of 1e-2: nothing usable

    def point_idx(self, theta, slot, feas=False):
        J, u0, st = super().point_idx(theta, slot, feas)
        if (feas and self.stall_phase_one) or (not feas and self.stall_points):
            st = st.copy()
            st[0] = 1
            if feas:
                J = J.copy()
                J[0] = 1e-3             # a last iterate that is NOT feasible
        return J, u0, st

    def simplex_idx(self, R, slot, mode, Vbar=None):
        obj, alpha, st = super().simplex_idx(R, slot, mode, Vbar)
        if mode == 1 and self.stall_slack:
            st = st.copy()
            st[0] = 1 | (self.stall_decade << 8)
        if mode == 2 and self.infeasible_sliver:
            obj = obj.copy()
            obj[0] = 1e-3
        return obj, alpha, st


def test_stalled_device_solves_are_never_taken_for_answers(monkeypatch):
    """
    sequences.PrefixTable honours the solver status of every launch: a stalled optimum is +inf (a
    failed vertex solve for the callers' blacklist / retry paths, lib/oracle.py:214-218, 406-414),
    a stalled phase one that has not reached the tolerance raises instead of pruning its prefix,
    a stalled suboptimality test is settled by phase one (interior-free sliver: infeasible) or
    raises -- never a silent verdict (the enumerating Oracle raises SolverError there too).
    """
    from explicit_hybrid_mpc_amd import sequences
    from explicit_hybrid_mpc_amd.oracle import SolverError
    monkeypatch.setattr(sequences.engine, 'GpuProblem', _StallingStub)
    mpc = helpers.make_instance('pwa_small', 0)
    table = sequences.PrefixTable(mpc, slots=8)
    rng = np.random.default_rng(5)
    R = np.array(helpers.random_simplices(mpc, rng, 1, scale_lo=-2.)[0])
    seq = table.first_feasible(R)
    assert seq is not None
    pairs, thetas = [seq, seq], np.array([R[0], R[1]])
    J0, _ = table.solve_points(pairs, thetas, known_feasible=True)
    assert np.isfinite(J0).all() and table.stalled == 0
    table.gp.stall_points = True
    J1, _ = table.solve_points(pairs, thetas, known_feasible=True)
    assert np.isinf(J1[0]) and J1[1] == J0[1] and table.stalled == 1
    table.gp.stall_points = False
    table.gp.stall_phase_one = True
    with pytest.raises(SolverError):
        table.solve_points(pairs, thetas)
    table.gp.stall_phase_one = False
    V = np.array([table.solve_points([seq], v[None], known_feasible=True)[0][0] for v in R])
    t0, _ = table.solve_slack([seq], R[None], V[None], known_feasible=[True])
    assert np.isfinite(t0[0])
    table.gp.stall_slack = True
    with pytest.raises(SolverError):            # feasible on the simplex, yet no value
        table.solve_slack([seq], R[None], V[None], known_feasible=[True])
    # a solve that stalled CLOSE to its optimum (merit 1e4: good to ~1e-6) answers where the value
    # is far from zero -- the sign is all the suboptimality test asks (lib/oracle.py:440-442
    # accepts OPTIMAL_INACCURATE the same way); within the error bar it still raises
    table.gp.stall_decade = 4
    assert abs(t0[0]) > 1e-3
    t2, _ = table.solve_slack([seq], R[None], V[None], known_feasible=[True])
    assert t2[0] == t0[0] and table.accepted_inaccurate == 1
    assert not sequences.decisive_inaccurate([1 | (4 << 8)], [5e-5])[0]
    assert not sequences.decisive_inaccurate([1 | (7 << 8)], [0.5])[0]
    assert not sequences.decisive_inaccurate([0], [0.5])[0]
    table.gp.stall_decade = 8
    table.gp.infeasible_sliver = True           # phase one says: nothing of it on this simplex
    t1, _ = table.solve_slack([seq], R[None], V[None], known_feasible=[True])
    assert t1[0] == -np.inf
    table.close()


def test_deepest_first_order_grows_the_same_tree_and_finishes_regions_first():
    """
    grow_frontier(order='deepest') -- the order of the reference's recursive workers
    (lib/worker.py:403-417) -- ends with the tree of the level-by-level order; stopped early it
    has closed more regions for the same number of node visits.
    """
    from explicit_hybrid_mpc_amd import bnb_frontier
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)

    def run(**kw):
        orc = bnb.PrefixOracle(mpc, eps_a, 0.2, table=prefix_bb.CpuPrefixTable(mpc))
        trees = [Tree(NodeData(vertices=np.array(R))) for R in roots]
        stats = bnb_frontier.grow_frontier(orc, trees, 'ecc', handoff=False,
                                           split_batch=_host_split_batch, **kw)
        cells = {}
        for t, loc0 in zip(trees, locs):
            for nd, loc in t.walk(loc0):
                cells[loc] = (nd.is_leaf(), bool(nd.data.is_epsilon_suboptimal))
        return stats, cells
    s_fifo, fifo = run()
    s_deep, deep = run(order='deepest', round_cap=3)
    assert not s_fifo['truncated'] and not s_deep['truncated']
    assert fifo == deep and s_deep['regions'] == s_fifo['regions'] == sum(
        1 for leaf, closed in fifo.values() if leaf and closed)
    budget = s_fifo['host_visits'] // 2
    a, _ = run(max_visits=budget, round_cap=3)
    b, _ = run(max_visits=budget, round_cap=3, order='deepest')
    assert a['truncated'] and b['truncated'] and b['regions'] >= a['regions']
    c, _ = run(min_regions=3, round_cap=3, order='deepest')
    assert c['truncated'] and 3 <= c['regions'] < s_fifo['regions']


def test_short_prefixes_on_a_table_of_the_short_horizon(monkeypatch):
    """
    sequences.SplitPrefixTable: prefixes of at most ``short`` steps are solved as blocks of the
    same law with horizon ``short`` (PWAMPC.with_horizon), the rest as blocks of the full model --
    the same optima, slacks, minima and verdicts as the one-table form, pair for pair, and the
    frontier-wide driver on it grows the enumerating CPU partition's tree.  (HiGHS stub of the
    device problem on the condensed blocks.)
    """
    from explicit_hybrid_mpc_amd import sequences, bnb_frontier
    monkeypatch.setattr(sequences.engine, 'GpuProblem', _StubDeviceProblem)
    mpc = helpers.make_instance('pwa_small', 0)                  # 2 modes, N = 3
    assert sequences.short_horizon(mpc) == 0                     # it fits as it is: force the split
    one = sequences.PrefixTable(mpc, slots=16)
    two = sequences.SplitPrefixTable(mpc, 2, slots=16)
    assert two.short.mpc.N == 2 and two.long.mpc.N == 3 and two.short.full_length is None
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(17)
    prefixes = [(), (0,), (1,), (0, 1), (1, 0), (1, 1), (1, 1, 0), (0, 0, 0), (1, 0, 1), (0, 1, 1)]
    pairs = [prefixes[k] for k in rng.integers(0, len(prefixes), 60)]
    thetas = rng.uniform(-1, 1, (60, 2)) * half
    J1, u1 = one.solve_points(pairs, thetas)
    J2, u2 = two.solve_points(pairs, thetas)
    assert np.array_equal(np.isfinite(J1), np.isfinite(J2))
    fin = np.isfinite(J1)
    assert np.allclose(J1[fin], J2[fin], rtol=1e-9, atol=1e-10)
    full = np.array([len(q) == 3 for q in pairs])
    assert np.allclose(u1[fin & full], u2[fin & full], atol=1e-8)     # full sequences: same table
    assert two.short.lp_solves > 0 and two.long.lp_solves > 0
    simplices = np.array(helpers.random_simplices(mpc, rng, 60, scale_lo=-1.2))
    m1, m2 = one.solve_min(pairs, simplices), two.solve_min(pairs, simplices)
    assert np.array_equal(np.isfinite(m1), np.isfinite(m2))
    assert np.allclose(m1[np.isfinite(m1)], m2[np.isfinite(m1)], rtol=1e-9, atol=1e-10)
    V = rng.uniform(0.5, 2., (60, 3))
    t1, _ = one.solve_slack(pairs, simplices, V)
    t2, _ = two.solve_slack(pairs, simplices, V)
    assert np.array_equal(np.isfinite(t1), np.isfinite(t2))
    assert np.allclose(t1[np.isfinite(t1)], t2[np.isfinite(t1)], rtol=1e-9, atol=1e-10)
    # counters: the split table reports both halves
    assert two.lp_solves == two.short.lp_solves + two.long.lp_solves
    assert two.by_length.sum() == two.short.by_length.sum() + two.long.by_length.sum()
    assert two.by_length[:, 3].sum() == two.long.by_length[:, 3].sum()
    assert two.short.by_length[:, :3].sum() == two.short.by_length.sum()     # nothing longer there
    one.close()
    # the searches and the driver on the split table
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    dev = bnb.PrefixOracle(mpc, eps_a, 0.2, table=two)
    roots, locs = helpers.roots_of(mpc)
    cpu = PartitionCPU(OracleCPU(mpc, eps_a, 0.2))
    cpu.run(roots, locs, 'ecc')
    trees = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    stats = bnb_frontier.grow_frontier(dev, trees, 'ecc', handoff=False,
                                       split_batch=_host_split_batch)
    assert not stats['truncated']
    n = 0
    for t, loc0 in zip(trees, locs):
        for nd, loc in t.walk(loc0):
            r = cpu.nodes[loc]
            n += 1
            assert nd.is_leaf() == r['leaf']
            assert nd.data.is_epsilon_suboptimal == r['is_epsilon_suboptimal']
            if r['commutation'] is not None:
                assert np.array_equal(nd.data.commutation.astype(int), r['commutation'].astype(int))
                assert np.allclose(nd.data.vertex_costs, r['vertex_costs'], rtol=1e-7, atol=1e-8)
    assert n == len(cpu.nodes)
    two.close()


@pytest.mark.parametrize('kind', ['cpu', 'stub', 'split'])
def test_native_queues_of_bar_e_equal_the_python_statement(kind, monkeypatch):
    """
    bnb_frontier.bar_e_many keeps its best-first queues natively (include/ehm_search.h:
    ehm_search_bare_*) and asks the table for (prefix code, node) pairs; the plain-Python statement
    of the same search (tests/prefix_search_py.bar_e_many_py: heapq on (-t, prefix), dictionaries,
    per-pair lists) must take the same decisions, report the same margins, expand the same
    prefixes, solve the SAME problems and leave the same optima for the nodes that stay open --
    on the HiGHS table of the uncondensed relaxations ('cpu': the generic pair path), on the
    device-table class over a HiGHS stub ('stub': codes -> slots with numpy) and on the two-table
    form ('split').  Nodes with and without an incumbent, with and without inherited bounds.
    """
    from explicit_hybrid_mpc_amd import sequences, bnb_frontier
    from tests.prefix_search_py import bar_e_many_py
    monkeypatch.setattr(sequences.engine, 'GpuProblem', _StubDeviceProblem)
    mpc = helpers.make_instance('pwa_small', 0)                  # 2 modes, N = 3
    eps_a = helpers.eps_a_rule(mpc, 0.25)

    def make():
        if kind == 'cpu':
            table = prefix_bb.CpuPrefixTable(mpc, eps_a, 0.05)
        elif kind == 'stub':
            table = sequences.PrefixTable(mpc, slots=16, eps_a=eps_a, eps_r=0.05)
        else:
            table = sequences.SplitPrefixTable(mpc, 2, slots=16, eps_a=eps_a, eps_r=0.05)
        return bnb.PrefixOracle(mpc, eps_a, 0.05, table=table)
    ref = bnb.PrefixOracle(mpc, eps_a, 0.05, table=prefix_bb.CpuPrefixTable(mpc, eps_a, 0.05))
    rng = np.random.default_rng(23)
    Rs, Vs, incs, bounds = [], [], [], []
    every = [q for k in range(1, mpc.N + 1)
             for q in itertools.product(range(mpc.delta_size), repeat=k)]
    for R in list(helpers.random_simplices(mpc, rng, 40, scale_lo=-1.3)) + \
            [np.array(r) for r in helpers.roots_of(mpc)[0]]:
        R = np.array(R)
        delta, vx = ref.V_R(R)
        if delta is None:
            continue
        V = np.array([v[1] for v in vx])
        Rs.append(R)
        Vs.append(V)
        incs.append(ref.sequence_of(delta) if rng.random() < 0.5 else None)
        if rng.random() < 0.5:          # inherited upper bounds: some refute, some do not
            some = [every[i] for i in rng.choice(len(every), size=6, replace=False)]
            bounds.append({q: float(rng.choice([-1., -1e-9, 0.3])) for q in some})
        else:
            bounds.append(None)
    assert len(Rs) >= 12
    a, b = make(), make()
    la, lb = [dict() for _ in Rs], [dict() for _ in Rs]
    ca, ma = bnb_frontier.bar_e_many(a, Rs, Vs, bounds, la, incs)
    cb, mb = bar_e_many_py(b, Rs, Vs, bounds, lb, incs)
    assert ca == cb and any(ca) and not all(ca)
    assert np.allclose(ma, mb, rtol=1e-12, atol=0.)
    assert a.n_expanded == b.n_expanded and a.n_inherited == b.n_inherited
    assert a.table.lp_solves == b.table.lp_solves and a.calls == b.calls
    for j, closed in enumerate(ca):
        if closed:
            continue                    # a closed node's optima have no reader
        assert set(la[j]) == set(lb[j]), j
        for q in la[j]:
            assert la[j][q][0] == lb[j][q][0] or abs(la[j][q][0] - lb[j][q][0]) <= 1e-12
            assert (la[j][q][1] is None) == (lb[j][q][1] is None)
    # without incumbents, bounds or a place for the optima
    c0, m0 = bnb_frontier.bar_e_many(a, Rs[:5], Vs[:5])
    c1, m1 = bar_e_many_py(b, Rs[:5], Vs[:5])
    assert c0 == c1 and np.allclose(m0, m1, rtol=1e-12, atol=0.)
    assert bnb_frontier.bar_e_many(a, [], []) == ([], [])
    a.close()
    b.close()
    ref.close()
