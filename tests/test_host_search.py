"""
The native bookkeeping of the prefix searches (include/ehm_search.h, csrc/ehm_search.cpp: point
ids, the memo of phase-one verdicts with midpoint inference, one problem per pair and launch, the
lockstep descents of V_R) -- host code, so it is tested here without a device:

  * on a synthetic solver whose feasible sets are intersections of half-spaces (convex, shrinking
    with the prefix, like the relaxations'), against brute-force enumeration;
  * against the plain-Python statement it replaced (tests/prefix_search_py.py) on the HiGHS
    stand-in of the table, search by search and on the tree the driver grows.
"""

import ctypes
import itertools

import numpy as np
import pytest

from explicit_hybrid_mpc_amd import _capi, bnb, bnb_frontier, examples, sequences
from explicit_hybrid_mpc_amd._capi import ptr
from explicit_hybrid_mpc_amd.tree import Tree, NodeData
from oracle import prefix_bb
from tests import helpers
from tests.prefix_search_py import PyPrefixSearch
from tests.test_host_bnb import _host_split_batch


class _Shape:
    def __init__(self, n_x, delta_size, N):
        self.n_x, self.delta_size, self.N, self.n_u = n_x, delta_size, N, 1


class HalfSpaceTable(sequences.PrefixSearch):
    """Prefix q is feasible at theta iff  a[i, q_i] . theta <= b[i, q_i]  for every i < len(q)."""

    def __init__(self, n_x=3, delta_size=3, N=4, seed=0):
        self.mpc = _Shape(n_x, delta_size, N)
        rng = np.random.default_rng(seed)
        self.a = rng.normal(size=(N, delta_size, n_x))
        self.b = rng.uniform(-0.2, 1.2, size=(N, delta_size))
        self.asked = []
        self.init_search()

    def feasible(self, q, theta):
        return all(self.a[i, d] @ theta <= self.b[i, d] for i, d in enumerate(q))

    def solve_points(self, prefixes, thetas, feasibility_only=False, known_feasible=False):
        thetas = np.asarray(thetas).reshape(len(prefixes), -1)
        J = np.full(len(prefixes), np.inf)
        for k, q in enumerate(prefixes):
            self.asked.append((tuple(q), thetas[k].tobytes()))
            if self.feasible(q, thetas[k]):
                J[k] = 0.
        return J, np.zeros((len(prefixes), 1))

    def brute_first(self, points, exclude=()):
        for q in itertools.product(range(self.mpc.delta_size), repeat=self.mpc.N):
            if q not in exclude and all(self.feasible(q, th) for th in points):
                return q
        return None


def test_descents_against_enumeration_and_every_pair_solved_once():
    tab = HalfSpaceTable()
    rng = np.random.default_rng(1)
    sets = [rng.uniform(-0.5, 0.5, (4, 3)) * rng.uniform(0.05, 1.) + rng.uniform(-0.3, 0.3, 3)
            for _ in range(60)]
    sets += [sets[0].copy(), sets[1][:2].copy()]          # shared points, a ragged set
    got = tab.first_feasible_many(sets)
    want = [tab.brute_first(s) for s in sets]
    assert got == want
    assert sum(q is not None for q in want) >= 10 and any(q is None for q in want)
    # nothing was solved twice, although set 0 and its copy descend side by side
    assert len(tab.asked) == len(set(tab.asked))
    _, points, handed, shared = tab.search_counts()
    assert handed == len(tab.asked) and shared > 0 and points == len({p.tobytes() for s in sets for p in s})
    # a second pass costs nothing
    n = len(tab.asked)
    assert tab.first_feasible_many(sets) == want and len(tab.asked) == n
    # the blacklist (lib/oracle.py:198): skip the first answers
    have = [k for k, q in enumerate(want) if q is not None][:8]
    excl = [[want[k]] if k in have else [] for k in range(len(sets))]
    again = tab.first_feasible_many(sets, excl)
    for k in range(len(sets)):
        assert again[k] == tab.brute_first(sets[k], exclude=set(excl[k]))
    assert all(again[k] != want[k] for k in have)
    tab.close_search()


def test_feasible_sets_and_midpoint_inference():
    tab = HalfSpaceTable(seed=3)
    rng = np.random.default_rng(4)
    prefixes = [q for k in range(1, 4) for q in itertools.product(range(3), repeat=k)]
    ends_a = rng.uniform(-0.4, 0.4, (30, 3))
    ends_b = rng.uniform(-0.4, 0.4, (30, 3))
    pick = [prefixes[k] for k in rng.integers(0, len(prefixes), 200)]
    sets = [np.array([ends_a[k], ends_b[k]]) for k in rng.integers(0, 30, 200)]
    flags = tab.feasible_sets(pick, sets)
    assert np.array_equal(flags, [all(tab.feasible(q, th) for th in s) for q, s in zip(pick, sets)])
    assert flags.any() and not flags.all()
    # midpoints of edges whose ends are both known feasible need no problem
    mids = 0.5 * (ends_a + ends_b)
    tab.register_midpoints(mids, ends_a, ends_b)
    n = len(tab.asked)
    mid_sets = [0.5 * (s[0] + s[1])[None] for s in sets]
    f2 = tab.feasible_sets(pick, mid_sets)
    assert np.array_equal(f2, [tab.feasible(q, s[0]) for q, s in zip(pick, mid_sets)])
    solved = {(q, th) for q, th in tab.asked[n:]}
    for q, s, ms, ok in zip(pick, sets, mid_sets, flags):
        if ok:                                          # both ends feasible: inferred
            assert (q, ms[0].tobytes()) not in solved
    assert all(f2[k] for k in np.flatnonzero(flags))
    # ids given by the caller are the ids of the values
    ids = [tab.point_ids(s) for s in sets]
    assert np.array_equal(tab.feasible_sets(pick, None, ids), flags) and len(tab.asked) == n + len(solved)
    # forgetting drops the verdicts, not the ids or the midpoints
    tab.FEAS_MEMO_LIMIT = -1
    assert np.array_equal(tab.feasible_sets(pick, None, ids), flags)
    assert tab.search_counts()[0] > 0 and len(tab.asked) > n + len(solved)
    tab.close_search()


def test_protocol_errors_are_reported():
    lib = _capi.load()
    h = ctypes.c_void_p()
    assert lib.ehm_search_create(8, 5, 12, ctypes.byref(h)) == _capi.EHM_E_INVALID     # 6^12 > 2^26
    assert b'2^26' in lib.ehm_search_last_error()
    assert lib.ehm_search_create(9, 2, 3, ctypes.byref(h)) == _capi.EHM_E_INVALID      # p > EHM_MAX_P
    _capi.check_search(lib.ehm_search_create(2, 2, 3, ctypes.byref(h)))
    pts = np.array([[0., 0.], [1., 0.], [0., 0.]])
    ids = np.empty(3, dtype=np.int64)
    _capi.check_search(lib.ehm_search_point_ids(h, 3, ptr(pts), ptr(ids)))
    assert list(ids) == [0, 1, 0]
    codes = np.array([1, 2], dtype=np.uint64)
    begin = np.array([0, 2, 3], dtype=np.int64)
    pid = np.array([0, 1, 1], dtype=np.int64)
    flags = np.zeros(2, dtype=np.uint8)
    na, npre = ctypes.c_int64(), ctypes.c_int64()
    bad = np.array([0, 7, 1], dtype=np.int64)
    assert lib.ehm_search_query(h, 2, ptr(codes), ptr(begin), ptr(bad), ptr(flags),
                                ctypes.byref(na), ctypes.byref(npre)) == _capi.EHM_E_INVALID
    big = np.array([1, 27], dtype=np.uint64)                                            # 27 = 3^3
    assert lib.ehm_search_query(h, 2, ptr(big), ptr(begin), ptr(pid), ptr(flags),
                                ctypes.byref(na), ctypes.byref(npre)) == _capi.EHM_E_INVALID
    _capi.check_search(lib.ehm_search_query(h, 2, ptr(codes), ptr(begin), ptr(pid), ptr(flags),
                                            ctypes.byref(na), ctypes.byref(npre)))
    assert (na.value, npre.value) == (3, 2) and list(flags) == [1, 1]
    # a second question, a descent or forgetting while the pairs are pending: refused
    assert lib.ehm_search_query(h, 2, ptr(codes), ptr(begin), ptr(pid), ptr(flags),
                                ctypes.byref(na), ctypes.byref(npre)) == _capi.EHM_E_INVALID
    assert lib.ehm_search_forget(h) == _capi.EHM_E_INVALID
    assert lib.ehm_search_descent_begin(h, 1, ptr(begin), ptr(pid), None, None) == _capi.EHM_E_INVALID
    pc, pi, th = np.empty(2, dtype=np.uint64), np.empty(3, dtype=np.int64), np.empty((3, 2))
    _capi.check_search(lib.ehm_search_asks(h, ptr(pc), ptr(pi), ptr(th)))
    assert list(pc) == [1, 2] and list(pi) == [0, 0, 1] and np.array_equal(th, pts[[0, 1, 1]])
    ok = np.array([1, 0, 1], dtype=np.uint8)
    _capi.check_search(lib.ehm_search_answer(h, ptr(ok), ptr(flags)))
    assert list(flags) == [0, 1]
    assert lib.ehm_search_answer(h, ptr(ok), ptr(flags)) == _capi.EHM_E_INVALID          # nothing pending
    # descents: a result before they are finished is refused
    _capi.check_search(lib.ehm_search_descent_begin(h, 1, ptr(begin), ptr(pid), None, None))
    _capi.check_search(lib.ehm_search_descent_step(h, None, ctypes.byref(na), ctypes.byref(npre)))
    assert na.value > 0
    seq = np.empty((1, 3), dtype=np.int32)
    assert lib.ehm_search_descent_result(h, ptr(seq), None) == _capi.EHM_E_INVALID
    assert lib.ehm_search_descent_step(h, None, ctypes.byref(na), ctypes.byref(npre)) == _capi.EHM_E_INVALID
    while na.value:
        v = np.ones(na.value, dtype=np.uint8)
        _capi.check_search(lib.ehm_search_descent_step(h, ptr(v), ctypes.byref(na), ctypes.byref(npre)))
    _capi.check_search(lib.ehm_search_descent_result(h, ptr(seq), None))
    # prefix (0,) was answered infeasible at point 1 above and is remembered: mode 1 first
    assert list(seq[0]) == [1, 0, 0]
    lib.ehm_search_destroy(h)


class _PyTable(PyPrefixSearch, prefix_bb.CpuPrefixTable):
    """The HiGHS stand-in of the table with the plain-Python memo and descents."""


def test_native_bookkeeping_equals_the_python_statement_on_the_cpu_table():
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    nat = bnb.PrefixOracle(mpc, eps_a, 0.2, table=prefix_bb.CpuPrefixTable(mpc))
    py = bnb.PrefixOracle(mpc, eps_a, 0.2, table=_PyTable(mpc))
    assert not hasattr(py.table, '_search') and nat.table._search
    rng = np.random.default_rng(5)
    Rs = [np.array(R) for R in helpers.random_simplices(mpc, rng, 12, scale_lo=-1.5)]
    a = nat.table.first_feasible_many(Rs + Rs[:3])
    b = py.table.first_feasible_many(Rs + Rs[:3])
    assert a == b and any(q is not None for q in a)
    assert nat.table.lp_solves < py.table.lp_solves          # shared pairs are solved once
    # the frontier-wide driver on both: the same tree
    roots, _ = helpers.roots_of(mpc)
    trees = []
    for orc in (nat, py):
        ts = [Tree(NodeData(vertices=np.array(R))) for R in roots]
        stats = bnb_frontier.grow_frontier(orc, ts, 'ecc', handoff=False,
                                           split_batch=_host_split_batch)
        assert not stats['truncated']
        trees.append(ts)
    n = 0
    for t1, t2 in zip(*trees):
        w1, w2 = list(t1.walk()), list(t2.walk())
        assert [loc for _, loc in w1] == [loc for _, loc in w2]
        for (n1, _), (n2, _) in zip(w1, w2):
            n += 1
            assert np.array_equal(n1.data.vertices, n2.data.vertices)
            assert n1.data.is_epsilon_suboptimal == n2.data.is_epsilon_suboptimal
            c1, c2 = getattr(n1.data, 'commutation', None), getattr(n2.data, 'commutation', None)
            assert (c1 is None) == (c2 is None)
            if c1 is not None:
                assert np.array_equal(c1, c2)
                assert np.allclose(n1.data.vertex_costs, n2.data.vertex_costs, rtol=1e-9, atol=1e-10)
    assert n > len(roots)


def test_warm_started_searches_grow_the_enumerating_partition():
    """
    bnb_frontier with everything it inherits -- bounds of t* proven on the ancestors, the
    parent's best-slack sequence tried first, bar_E's optima reused by bar_D, the barycentre
    witness, remembered vertex optima -- at a tolerance where lcss refines over several levels:
    the tree of the ENUMERATING CPU partition (every commutation, every node from scratch), and
    the same answers from bar_e_many / bar_d_many with and without what they are handed.
    """
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.2)     # (tighter, this instance refines along a cost jump
    eps_r = 0.1                              # for ever -- DESIGN.md section 6, config 3)
    roots, locs = helpers.roots_of(mpc)
    cpu = PartitionCPU(OracleCPU(mpc, eps_a, eps_r))
    cpu.run(roots, locs, 'ecc')
    orc = bnb.PrefixOracle(mpc, eps_a, eps_r, table=prefix_bb.CpuPrefixTable(mpc))
    trees = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    stats = bnb_frontier.grow_frontier(orc, trees, 'ecc', handoff=False,
                                       split_batch=_host_split_batch)
    assert not stats['truncated']
    assert orc.n_inherited > 20                      # the searches did lean on their ancestors
    n = depth = 0
    for t, loc0 in zip(trees, locs):
        for nd, loc in t.walk(loc0):
            r = cpu.nodes[loc]
            n += 1
            depth = max(depth, len(loc) - len(loc0))
            assert np.array_equal(nd.data.vertices, r['vertices'])
            assert nd.is_leaf() == r['leaf']
            assert nd.data.is_epsilon_suboptimal == r['is_epsilon_suboptimal']
            if r['commutation'] is not None:
                assert np.array_equal(nd.data.commutation.astype(int), r['commutation'].astype(int))
                assert np.allclose(nd.data.vertex_costs, r['vertex_costs'], rtol=1e-7, atol=1e-8)
    assert n == len(cpu.nodes) and depth >= 4
    # the two searches on inner nodes, with and without the parent's knowledge
    inner = [(nd, loc) for t, loc0 in zip(trees, locs) for nd, loc in t.walk(loc0)
             if not nd.is_leaf() and hasattr(nd.data, 'commutation')][:12]
    assert len(inner) >= 4
    compared = 0
    for nd, _ in inner:
        for kid in (nd.left, nd.right):
            if not hasattr(kid.data, 'commutation'):
                continue
            R, V = np.asarray(nd.data.vertices), np.asarray(nd.data.vertex_costs)
            learned, stars = [dict()], []
            bnb_frontier.bar_e_many(orc, [R], [V], None, learned)
            bnb_frontier.bar_d_many(orc, [R], [V], [nd.data.commutation], None, learned, None, stars)
            Rk, Vk = np.asarray(kid.data.vertices), np.asarray(kid.data.vertex_costs)
            if not np.array_equal(kid.data.commutation, nd.data.commutation) or \
                    not np.all(np.delete(Vk, np.flatnonzero(np.any(Rk != R, axis=1))) <=
                               np.delete(V, np.flatnonzero(np.any(Rk != R, axis=1)))):
                continue                              # the bounds do not carry over (grow_frontier)
            bound = {q: v[0] for q, v in learned[0].items()}
            cold = bnb_frontier.bar_e_many(orc, [Rk], [Vk])[0]
            warm = bnb_frontier.bar_e_many(orc, [Rk], [Vk], [bound], [dict()], [stars[0]])[0]
            assert cold == warm
            compared += 1
            a = bnb_frontier.bar_d_many(orc, [Rk], [Vk], [kid.data.commutation])[0]
            b = bnb_frontier.bar_d_many(orc, [Rk], [Vk], [kid.data.commutation], [bound], [dict()],
                                        [stars[0]])[0]
            assert (a[0] is None) == (b[0] is None)
            if a[0] is not None:
                assert np.array_equal(a[0], b[0]) and np.allclose(a[1], b[1], atol=1e-9) and a[3] == b[3]
    assert compared >= 3


def test_tables_grow_past_their_first_sizes():
    """Enough points (> 4096 x 0.6) and verdicts (> 65536 x 0.6) to rehash both tables several
    times: ids stay ids of values, verdicts stay verdicts."""
    tab = HalfSpaceTable(n_x=3, delta_size=3, N=4, seed=7)
    rng = np.random.default_rng(8)
    pts = rng.uniform(-0.6, 0.6, (12000, 3))
    ids = tab.point_ids(pts)
    assert np.array_equal(ids, np.arange(12000))
    assert np.array_equal(tab.point_ids(pts[::-1]), ids[::-1])           # by value, any order
    prefixes = [q for k in range(1, 5) for q in itertools.product(range(3), repeat=k)]
    pick = [prefixes[k] for k in rng.integers(0, len(prefixes), 60000)]
    which = rng.integers(0, 12000, (60000, 2))
    idl = [ids[w] for w in which]
    flags = tab.feasible_sets(pick, None, idl)
    want = np.array([tab.feasible(q, pts[w[0]]) and tab.feasible(q, pts[w[1]])
                     for q, w in zip(pick, which)])
    assert np.array_equal(flags, want)
    held = tab.search_counts()[0]
    assert held > 65536 and len(tab.asked) == len(set(tab.asked)) == held
    n = len(tab.asked)
    assert np.array_equal(tab.feasible_sets(pick, None, idl), want) and len(tab.asked) == n
    tab.close_search()


def test_a_failed_launch_leaves_the_search_state_usable():
    tab = HalfSpaceTable(seed=9)
    rng = np.random.default_rng(10)
    sets = [rng.uniform(-0.4, 0.4, (3, 3)) for _ in range(20)]
    good = tab.solve_points
    calls = {'n': 0}

    def flaky(prefixes, thetas, feasibility_only=False, known_feasible=False):
        calls['n'] += 1
        if calls['n'] == 3:
            raise RuntimeError('device lost')
        return good(prefixes, thetas, feasibility_only, known_feasible)
    tab.solve_points = flaky
    with pytest.raises(RuntimeError):
        tab.first_feasible_many(sets)
    # nothing is pending, what the failed launch was to decide is simply unknown again
    assert tab.first_feasible_many(sets) == [tab.brute_first(s) for s in sets]
    prefixes = [(0,), (1,), (2, 0)]
    assert np.array_equal(tab.feasible_sets(prefixes, [sets[0]] * 3),
                          [all(tab.feasible(q, th) for th in sets[0]) for q in prefixes])
    tab.close_search()


class _PyHalfSpace(PyPrefixSearch, HalfSpaceTable):
    """The synthetic solver under the plain-Python memo and descents."""


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_random_operation_sequences_against_the_python_statement(seed):
    """The same random sequence of questions, midpoint registrations and memo resets put to the
    native state and to the Python statement: the same answers throughout."""
    nat, py = HalfSpaceTable(seed=20 + seed), _PyHalfSpace(seed=20 + seed)
    rng = np.random.default_rng(seed)
    pool = rng.uniform(-0.5, 0.5, (40, 3))
    prefixes = [q for k in range(1, 5) for q in itertools.product(range(3), repeat=k)]
    for step in range(120):
        op = rng.integers(0, 10)
        if op < 4:                                   # feasible_sets on ragged sets
            n = int(rng.integers(1, 12))
            pick = [prefixes[k] for k in rng.integers(0, len(prefixes), n)]
            sets = [pool[rng.integers(0, len(pool), int(rng.integers(1, 5)))] for _ in range(n)]
            assert np.array_equal(nat.feasible_sets(pick, sets), py.feasible_sets(pick, sets)), step
        elif op < 7:                                 # descents, some with a blacklist
            n = int(rng.integers(1, 8))
            sets = [pool[rng.integers(0, len(pool), int(rng.integers(1, 4)))] for _ in range(n)]
            excl = None
            if op == 6:
                excl = [[prefixes[k] for k in rng.integers(len(prefixes) - 81, len(prefixes), 3)]
                        for _ in range(n)]
            assert nat.first_feasible_many(sets, excl) == py.first_feasible_many(sets, excl), step
        elif op < 9:                                 # new points: midpoints of known ones
            i, j = rng.integers(0, len(pool), (2, 5))
            mids = 0.5 * (pool[i] + pool[j])
            nat.register_midpoints(mids, pool[i], pool[j])
            py.register_midpoints(mids, pool[i], pool[j])
            pool = np.vstack([pool, mids])
        else:                                        # the memo is dropped
            nat.FEAS_MEMO_LIMIT = py.FEAS_MEMO_LIMIT = -1
            q = prefixes[int(rng.integers(0, len(prefixes)))]
            assert np.array_equal(nat.feasible_sets([q], [pool[:2]]), py.feasible_sets([q], [pool[:2]]))
            nat.FEAS_MEMO_LIMIT = py.FEAS_MEMO_LIMIT = 3000000
    assert len(nat.asked) <= len(py.asked)           # never more problems than the Python form
    nat.close_search()


def test_region_tables_with_the_parents_minima():
    """region_tables_many with ``above``: a child's minimum of a prefix relaxation is at least its
    parent's, so what the parent's region prices above the child's bound is not solved again --
    the same tables from fewer problems; the level that breaks a table is reported."""
    mpc = examples.pwa4_mpc()                            # 4 modes, N = 4: 256 sequences
    half = examples.theta_box(mpc)
    E = np.vstack([np.zeros(8), np.eye(8)]) - 1. / 9.
    V = examples.box_vertices(half)
    table = prefix_bb.CpuPrefixTable(mpc)
    bo = bnb.PrefixOracle(mpc, 0.01, 0.02, table=table)
    parents = [0.8 * V[37] + 0.08 * half * E, 0.7 * V[100] + 0.05 * half * E]
    kids = [0.5 * (R + R[k]) for R, k in zip(parents, (0, 3))]          # inside their parents
    def record(Rs):
        comms, tops = [], []
        for R in Rs:
            delta, vx = bo.V_R(R)
            comms.append(delta)
            tops.append(max(v[1] for v in vx))
        return comms, tops
    pc, pt = record(parents)
    minima = [dict(), dict()]
    ptab = bnb_frontier.region_tables_many(bo, parents, pc, pt, 256, None, minima)
    assert all(t is not None for t in ptab) and all(len(m) > 50 for m in minima)
    kc, kt = record(kids)
    n0 = table.lp_solves
    cold = bnb_frontier.region_tables_many(bo, kids, kc, kt, 256)
    n1 = table.lp_solves
    own = [dict(), dict()]
    warm = bnb_frontier.region_tables_many(bo, kids, kc, kt, 256, minima, own)
    n2 = table.lp_solves
    assert warm == cold and all(t is not None for t in cold)
    assert n2 - n1 < n1 - n0
    for m_kid, m_par in zip(own, minima):                # the bound itself
        for q, c in m_kid.items():
            if q in m_par and np.isfinite(m_par[q]):
                assert c >= m_par[q] - 1e-7 * (1 + abs(c))
    excess = [0., 0.]
    assert bnb_frontier.region_tables_many(bo, kids, kc, kt, 4, None, None, excess) == [None, None]
    assert all(f > 1. for f in excess)


def test_driver_bookkeeping_when_some_nodes_are_handed_off(monkeypatch):
    """The hand-off removes nodes from a round between bar_E and bar_D; what the remaining ones
    inherit and hand down must stay aligned with them.  The device engine is replaced by a stand-in
    that takes every other open node (and leaves it to the enumerating CPU partition to finish),
    with and without the back-off of table attempts: the rest of the tree is the enumerating
    partition's."""
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a, eps_r = helpers.eps_a_rule(mpc, 0.2), 0.1
    roots, locs = helpers.roots_of(mpc)
    cpu = PartitionCPU(OracleCPU(mpc, eps_a, eps_r))
    cpu.run(roots, locs, 'ecc')
    for backoff in (False, True):
        orc = bnb.PrefixOracle(mpc, eps_a, eps_r, table=prefix_bb.CpuPrefixTable(mpc))
        taken = []

        def fake_hand_off(oracle, nodes, table_max, engine_opts, stats, above=None, costs=None,
                          excess=None):
            Rs = [np.asarray(nd.data.vertices) for nd in nodes]
            tabs = bnb_frontier.region_tables_many(
                oracle, Rs, [nd.data.commutation for nd in nodes],
                [float(np.max(nd.data.vertex_costs)) for nd in nodes], table_max, above, costs,
                excess)
            keep = []
            for k, nd in enumerate(nodes):
                if tabs[k] is not None and len(taken) % 2 == 0:
                    taken.append(nd)                     # "the engine grows this subtree"
                    stats['handoffs'] += 1
                else:
                    if tabs[k] is not None:
                        taken.append(None)
                    keep.append(k)
            return keep
        monkeypatch.setattr(bnb_frontier, '_hand_off', fake_hand_off)
        trees = [Tree(NodeData(vertices=np.array(R))) for R in roots]
        stats = bnb_frontier.grow_frontier(orc, trees, 'ecc', handoff=True, table_max=8,
                                           split_batch=_host_split_batch, table_backoff=backoff)
        assert not stats['truncated']
        handed = {id(nd) for nd in taken if nd is not None}
        n = 0
        for t, loc0 in zip(trees, locs):
            for nd, loc in t.walk(loc0):
                r = cpu.nodes[loc]
                assert np.array_equal(nd.data.vertices, r['vertices'])
                if id(nd) in handed:
                    assert nd.is_leaf() and not r['is_epsilon_suboptimal']     # left to the engine
                    continue
                n += 1
                assert nd.is_leaf() == r['leaf']
                assert nd.data.is_epsilon_suboptimal == r['is_epsilon_suboptimal']
                if r['commutation'] is not None:
                    assert np.array_equal(nd.data.commutation.astype(int),
                                          r['commutation'].astype(int))
        assert n > len(roots) and stats['handoffs'] == len(handed) >= 1


def test_best_first_queue_protocol_and_its_error_paths():
    """ehm_search_bare_* (include/ehm_search.h) on a synthetic slack function: two searches over 2
    modes x 3 steps advance in lockstep; seeds, refuting bounds, the ask / answer order and the
    misuse of the protocol (answer without a step, result before the end, bad indices)."""
    import ctypes
    from explicit_hybrid_mpc_amd import _capi
    from explicit_hybrid_mpc_amd._capi import ptr
    lib = _capi.load()
    n_modes, N, base = 2, 3, 3

    def code(q):
        return sum((d + 1) * base ** i for i, d in enumerate(q))

    def prefix(c):
        out = []
        while c:
            out.append(c % base - 1)
            c //= base
        return tuple(out)
    # slack of a prefix: search 0 is refuted everywhere (closed), search 1 is open through (1, 0, 1)
    def slack(j, q):
        if j == 0:
            return -0.5 - 0.1 * len(q)
        good = (1, 0, 1)
        return 1.0 - 0.1 * len(q) if q == good[:len(q)] else -0.2
    guard = np.array([1e-6, 1e-6])
    h = ctypes.c_void_p()
    assert lib.ehm_search_bare_create(2, n_modes, N, ptr(guard), ctypes.byref(h)) == 0
    n_ask, left = ctypes.c_int64(), ctypes.c_int64()
    # misuse before any step
    assert lib.ehm_search_bare_answer(h, None, ctypes.byref(left)) != 0
    assert b'no step in flight' in lib.ehm_search_last_error()
    closed, margin = np.empty(2, dtype=np.int8), np.empty(2)
    assert lib.ehm_search_bare_result(h, ptr(closed), ptr(margin), None) != 0
    assert b'not finished' in lib.ehm_search_last_error()
    assert lib.ehm_search_bare_seed(h, 5, 0, 0., 0) != 0             # no such search
    # an inherited bound that refutes (1, 1) for search 1, and one that does not count
    codes = np.array([code((1, 1)), code((1, 0))], dtype=np.uint64)
    tb = np.array([-1., -1e-9])
    assert lib.ehm_search_bare_bounds(h, 1, 2, ptr(codes), ptr(tb)) == 0
    asked, steps = [], 0
    while True:
        assert lib.ehm_search_bare_step(h, 4, ctypes.byref(n_ask), ctypes.byref(left)) == 0
        if not n_ask.value:
            break
        steps += 1
        c = np.empty(n_ask.value, dtype=np.uint64)
        o = np.empty(n_ask.value, dtype=np.int32)
        assert lib.ehm_search_bare_asks(h, ptr(c), ptr(o)) == 0
        assert lib.ehm_search_bare_step(h, 4, ctypes.byref(n_ask), ctypes.byref(left)) != 0   # in flight
        asked += [(int(j), prefix(int(q))) for q, j in zip(c, o)]
        t = np.array([slack(int(j), prefix(int(q))) for q, j in zip(c, o)])
        if steps == 1:
            # a slack that is not a number (a failed solve) must not prune: refused, the step
            # stays in flight and takes the real answers
            bad = t.copy()
            bad[0] = np.nan
            assert lib.ehm_search_bare_answer(h, ptr(bad), ctypes.byref(left)) != 0
            assert b'not a number' in lib.ehm_search_last_error()
        assert lib.ehm_search_bare_answer(h, ptr(t), ctypes.byref(left)) == 0
    counts = np.zeros(2, dtype=np.int64)
    assert lib.ehm_search_bare_result(h, ptr(closed), ptr(margin), ptr(counts)) == 0
    assert closed.tolist() == [1, 0] and abs(margin[0] - 0.6) < 1e-12 and abs(margin[1] - 0.7) < 1e-12
    # search 0 ends after its two one-step relaxations; search 1 never solves the refuted (1, 1)
    assert (0, (0,)) in asked and (0, (1,)) in asked and not any(j == 0 and len(q) > 1 for j, q in asked)
    assert (1, (1, 1)) not in asked and (1, (1, 0)) in asked and (1, (1, 0, 1)) in asked
    assert counts[1] == 1 and steps == 3                     # one child answered by the bound
    cnt = ctypes.c_int64()
    assert lib.ehm_search_bare_learned(h, 1, ctypes.byref(cnt), None, None) == 0 and cnt.value >= 5
    lc, lt = np.empty(cnt.value, dtype=np.uint64), np.empty(cnt.value)
    assert lib.ehm_search_bare_learned(h, 1, ctypes.byref(cnt), ptr(lc), ptr(lt)) == 0
    assert dict(zip((prefix(int(c)) for c in lc), lt))[(1, 0, 1)] == 0.7
    assert lib.ehm_search_bare_destroy(h) == 0
    # (n_modes + 1)^N beyond the code range is refused
    assert lib.ehm_search_bare_create(1, 4, 12, ptr(guard), ctypes.byref(h)) != 0
