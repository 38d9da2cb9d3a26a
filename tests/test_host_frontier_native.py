"""
The native partition driver (include/ehm_frontier.h, csrc/ehm_frontier.cpp) without a device:

* its condensation of prefix relaxations against ``PWAMPC.condense_prefix`` (numpy);
* the driver on the CPU statement of the table (oracle/prefix_bb.CpuPrefixTable through
  ``frontier.TableSolvers``) grows the tree ``bnb_frontier.grow_frontier`` grows on the same table
  -- cell for cell: vertices, verdicts, commutations, vertex costs;
* the library exports every symbol include/ehm_frontier.h declares.
"""

import ctypes
import itertools
import os
import re

import numpy as np
import pytest

from explicit_hybrid_mpc_amd import _capi, bnb, bnb_frontier, examples, frontier
from explicit_hybrid_mpc_amd.tree import NodeData, Tree
from oracle import geometry, prefix_bb
from tests import helpers


def _host_split_batch(R):
    out = [geometry.split_along_longest_edge(r) for r in R]
    return (np.array([o[0] for o in out]), np.array([o[1] for o in out]),
            np.array([o[2] for o in out], dtype=np.int32))


def test_header_symbols_are_exported():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(here, 'include', 'ehm_frontier.h')).read()
    declared = set(re.findall(r'\b(ehm_frontier_\w+)\s*\(', text))
    assert declared == set(_capi.EXPORTED_FRONTIER)
    lib = ctypes.CDLL(_capi.library_path())
    for name in declared:
        assert hasattr(lib, name), name


@pytest.mark.parametrize('law', ['pwa4', 'pwa_small', 'pwa'])
def test_native_condensation_is_condense_prefix(law):
    mpc = examples.pwa4_mpc(N=5) if law == 'pwa4' else helpers.make_instance(law, 0)
    rng = np.random.default_rng(0)
    for horizon in (mpc.N, max(1, mpc.N - 2)):
        ref = mpc if horizon == mpc.N else mpc.with_horizon(horizon)
        for _ in range(12):
            k = int(rng.integers(0, horizon + 1))
            prefix = tuple(int(i) for i in rng.integers(0, mpc.delta_size, k))
            G, w, S = frontier.condense_native(mpc, prefix, horizon)
            G0, w0, S0 = ref.condense_prefix(prefix)
            assert G.shape == G0.shape and S.shape == S0.shape
            # (numpy's products go through BLAS: the same sums in another order)
            assert np.allclose(G, G0, rtol=0, atol=1e-13)
            assert np.allclose(w, w0, rtol=0, atol=1e-13)
            assert np.allclose(S, S0, rtol=0, atol=1e-13)
            assert np.array_equal(G == 0., G0 == 0.) and np.array_equal(S == 0., S0 == 0.)


def _same_trees(a, b):
    na = {loc: nd for nd, loc in a.walk()}
    nb = {loc: nd for nd, loc in b.walk()}
    assert set(na) == set(nb)
    for loc, x in na.items():
        y = nb[loc]
        assert np.array_equal(x.data.vertices, y.data.vertices)
        assert x.is_leaf() == y.is_leaf()
        assert x.data.is_epsilon_suboptimal == y.data.is_epsilon_suboptimal
        assert hasattr(x.data, 'commutation') == hasattr(y.data, 'commutation'), loc
        if hasattr(x.data, 'commutation'):
            assert np.array_equal(x.data.commutation, y.data.commutation)
            assert np.allclose(x.data.vertex_costs, y.data.vertex_costs, atol=1e-12)
            assert np.allclose(x.data.vertex_inputs, y.data.vertex_inputs, atol=1e-9)
    return len(na)


@pytest.mark.parametrize('eps_r', [0.2, 2.0])
def test_native_driver_grows_the_tree_of_grow_frontier(eps_r):
    """eps_r 2.0: almost every cell closes on its first lcss visit; 0.2: open cells go through
    bar_D, adopt commutations in place or bisect, their children inherit the best-slack sequence
    -- all of it native, the same tree either way."""
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, _ = helpers.roots_of(mpc)
    ref_orc = bnb.PrefixOracle(mpc, eps_a, eps_r, table=prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r))
    ref = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    bnb_frontier.grow_frontier(ref_orc, ref, 'ecc', handoff=False, split_batch=_host_split_batch,
                               round_cap=7)
    table = prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r)
    solvers = frontier.TableSolvers(table, _host_split_batch)
    nat = frontier.NativeFrontier(mpc, eps_a, eps_r, solvers=solvers)
    slow = bnb.PrefixOracle(mpc, eps_a, eps_r, table=prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r))
    got = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    st = frontier.grow_cells(nat, got, slow_oracle=lambda: slow, round_cap=5,
                             slow_opts=dict(handoff=False, split_batch=_host_split_batch))
    n = sum(_same_trees(a, b) for a, b in zip(ref, got))
    assert st['n_nodes'] <= n and st['rounds'] > 2 and not st['truncated']
    closed = sum(1 for t in got for nd, _ in t.walk() if nd.is_leaf())
    assert st['regions'] == closed
    assert st['slow_path_cells'] == 0           # nothing was handed back to the interpreter
    if eps_r == 0.2:
        assert st['calls_bar_d'] >= 10 and st['lcss_visits'] > st['calls_bar_d']
    # a second run on the same handle (reset drops the tree and the memo) gives the same tree
    again = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    frontier.grow_cells(nat, again, slow_oracle=lambda: slow, round_cap=64,
                        slow_opts=dict(handoff=False, split_batch=_host_split_batch))
    assert sum(_same_trees(a, b) for a, b in zip(ref, again)) == n
    nat.close()


def test_native_driver_respects_a_budget_of_visits():
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, _ = helpers.roots_of(mpc)
    table = prefix_bb.CpuPrefixTable(mpc, eps_a, 0.2)
    nat = frontier.NativeFrontier(mpc, eps_a, 0.2,
                                  solvers=frontier.TableSolvers(table, _host_split_batch))
    nat.add_roots(roots[:1])
    st = nat.run(round_cap=2, max_visits=3)
    assert st['truncated'] and st['visits'] == 3
    flat = nat.export()
    leaves = flat['left'] < 0
    assert np.any(flat['flags'][leaves] & frontier.FR_PENDING)
    st = nat.run(round_cap=8)                   # ... and goes on where it stopped
    assert not st['truncated']
    flat = nat.export()
    assert not np.any(flat['flags'][flat['left'] < 0] & frontier.FR_PENDING)
    nat.close()


def test_a_failing_solver_surfaces_as_its_own_exception():
    mpc = helpers.make_instance('pwa_small', 0)
    table = prefix_bb.CpuPrefixTable(mpc, 1., 1.)

    class Boom(RuntimeError):
        pass

    def broken(*a, **k):
        raise Boom('no launch today')
    table.solve_points = broken
    nat = frontier.NativeFrontier(mpc, 1., 1.,
                                  solvers=frontier.TableSolvers(table, _host_split_batch))
    nat.add_roots(helpers.roots_of(mpc)[0][:1])
    with pytest.raises(Boom):
        nat.run()
    nat.close()


def test_depth_limit_leaves_open_cells_flagged():
    """``max_depth``: a cell at the limit is not bisected -- it stays an open leaf flagged
    EHM_FR_DEPTH (a law whose optimal cost jumps across a mode boundary is refined without end
    along it; so is the reference's partition, lib/worker.py:403-417 has no limit either)."""
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, _ = helpers.roots_of(mpc)
    table = prefix_bb.CpuPrefixTable(mpc, eps_a, 0.2)
    nat = frontier.NativeFrontier(mpc, eps_a, 0.2,
                                  solvers=frontier.TableSolvers(table, _host_split_batch))
    nat.add_roots(roots)
    st = nat.run(max_depth=3)
    flat = nat.export()
    leaves = flat['left'] < 0
    limited = (flat['flags'] & frontier.FR_DEPTH) != 0
    assert st['depth_limited'] == int(limited.sum()) > 0 and st['depth'] == 3
    assert not np.any(limited & ~leaves)
    assert not np.any(flat['flags'][leaves] & frontier.FR_PENDING)
    # every leaf is closed or at the limit
    closed = (flat['flags'] & frontier.FR_CLOSED) != 0
    assert np.all(closed[leaves] | limited[leaves]) and st['regions'] == int(closed.sum())
    nat.close()


def test_peek_any_is_the_early_exit_form_of_peek():
    """``ehm_search_peek_any`` (known at a vertex => known on the cell; first vertex nothing is
    held about) against the per-pair ``ehm_search_peek`` it replaces in the native driver,
    including the midpoint inference (feasible at both ends of a bisected edge)."""
    lib = _capi.load()
    p, n_modes, N = 3, 2, 4
    h = ctypes.c_void_p()
    _capi.check_search(lib.ehm_search_create(p, n_modes, N, ctypes.byref(h)))
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((40, p))
    pts[30:] = 0.5 * (pts[0:10] + pts[10:20])           # midpoints of (k, k + 10)
    ids = np.empty(40, dtype=np.int64)
    _capi.check_search(lib.ehm_search_point_ids(h, 40, _capi.ptr(pts), _capi.ptr(ids)))
    mid, ea, eb = ids[30:].copy(), ids[0:10].copy(), ids[10:20].copy()   # (kept alive: raw pointers)
    _capi.check_search(lib.ehm_search_register_midpoints(h, 10, _capi.ptr(mid), _capi.ptr(ea),
                                                         _capi.ptr(eb)))
    codes = np.array([1, 2, 1 + 3 * 1, 2 + 3 * 2, 1 + 3 * 2 + 9 * 1], dtype=np.uint64)
    # verdicts for a random subset of (prefix, point) pairs among the first 30 points
    n_sets = 200
    qc = rng.choice(codes, n_sets)
    sets = [rng.choice(30, size=rng.integers(1, 5), replace=False) for _ in range(n_sets)]
    begin = np.zeros(n_sets + 1, dtype=np.int64)
    np.cumsum([len(s) for s in sets], out=begin[1:])
    flat = np.concatenate(sets).astype(np.int64)
    flags = np.ones(n_sets, dtype=np.uint8)
    n_ask, n_pre = ctypes.c_int64(), ctypes.c_int64()
    qp = ids[flat].copy()
    _capi.check_search(lib.ehm_search_query(h, n_sets, _capi.ptr(qc), _capi.ptr(begin),
                                            _capi.ptr(qp), _capi.ptr(flags),
                                            ctypes.byref(n_ask), ctypes.byref(n_pre)))
    ok = (rng.random(n_ask.value) < 0.6).astype(np.uint8)
    _capi.check_search(lib.ehm_search_answer(h, _capi.ptr(ok), _capi.ptr(flags)))
    # questions over sets that mix known, unknown and midpoint points
    m = 300
    c2 = rng.choice(codes, m)
    sets2 = [rng.choice(40, size=4, replace=False) for _ in range(m)]
    b2 = np.arange(0, 4 * m + 1, 4, dtype=np.int64)
    f2 = ids[np.concatenate(sets2)].copy()
    ver = np.empty(4 * m, dtype=np.int8)
    # (peek first on a COPY of the questions' order: both calls may store inferred midpoints,
    # which changes nothing either of them reports)
    known = np.empty(m, dtype=np.uint8)
    first = np.empty(m, dtype=np.int32)
    _capi.check_search(lib.ehm_search_peek_any(h, m, _capi.ptr(c2), _capi.ptr(b2), _capi.ptr(f2),
                                               _capi.ptr(known), _capi.ptr(first)))
    rep = np.repeat(c2, 4)
    _capi.check_search(lib.ehm_search_peek(h, 4 * m, _capi.ptr(rep), _capi.ptr(f2), _capi.ptr(ver)))
    ver = ver.reshape(m, 4)
    assert np.array_equal(known.astype(bool), (ver == 1).any(axis=1))
    open_q = ~known.astype(bool)
    has_unknown = (ver == -1).any(axis=1)
    assert np.array_equal(first[open_q & has_unknown], (ver == -1).argmax(axis=1)[open_q & has_unknown])
    assert np.all(first[open_q & ~has_unknown] == -1)
    assert known.any() and open_q.any() and (ver[:, :] == 1).any()
    lib.ehm_search_destroy(h)


def test_native_driver_on_a_cell_across_a_mode_boundary_32_sequences():
    """n_x = 4, n_u = 2, N = 5, two modes (32 sequences), a cell across the boundary x_1 = 0 at
    tolerances that make lcss do most of the work (342 bar_E and 156 bar_D searches, inherited
    bounds and best-slack sequences at every level): the native driver grows the tree of
    ``bnb_frontier.grow_frontier`` with the same visits and -- every search step asking for the
    same problems -- the same number of LPs."""
    mpc = helpers.make_instance('pwa', 0)
    half = examples.theta_box(mpc)
    eps_a, eps_r = helpers.eps_a_rule(mpc, 0.03), 0.02
    rng = np.random.default_rng(4)
    centre = np.array([0.0, 0.3, -0.2, 0.1]) * half
    R = np.clip(centre + 0.8 * half * rng.uniform(-1, 1, (5, 4)), -half, half)
    ref_table = prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r)
    ref_orc = bnb.PrefixOracle(mpc, eps_a, eps_r, table=ref_table)
    ref = Tree(NodeData(vertices=R.copy()))
    s_ref = bnb_frontier.grow_frontier(ref_orc, ref, 'ecc', handoff=False,
                                       split_batch=_host_split_batch, round_cap=16)
    table = prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r)
    nat = frontier.NativeFrontier(mpc, eps_a, eps_r,
                                  solvers=frontier.TableSolvers(table, _host_split_batch))
    got = Tree(NodeData(vertices=R.copy()))
    st = frontier.grow_cells(nat, got, round_cap=16)
    assert _same_trees(ref, got) > 300
    assert st['slow_path_cells'] == 0 and st['visits'] == s_ref['host_visits']
    assert st['calls_bar_d'] == ref_orc.calls['bar_D'] > 100
    assert st['regions'] == s_ref['regions']
    assert table.lp_solves == ref_table.lp_solves
    nat.close()


@pytest.mark.parametrize('law', ['pwa_small', 'pwa'])
def test_native_p_theta_is_p_theta_many(law):
    """``ehm_frontier_p_theta`` (best-first for the optimal value, lexicographic walk for the
    first sequence that attains it) against ``bnb_frontier.p_theta_many`` on the same CPU table:
    the same optimum, the same canonical sequence, the same first input -- also where the
    parameter is infeasible for every sequence."""
    mpc = helpers.make_instance(law, 0)
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(7)
    n = 14 if law == 'pwa' else 24
    thetas = rng.uniform(-1, 1, (n, mpc.n_x)) * half
    thetas[-1] = 3. * half                                   # outside the feasible set
    thetas[0] = 0.                                           # ties: every mode is admissible at 0
    ref_table = prefix_bb.CpuPrefixTable(mpc)
    ref = bnb_frontier.p_theta_many(bnb.PrefixOracle(mpc, 1., 1., table=ref_table), thetas)
    table = prefix_bb.CpuPrefixTable(mpc)
    nat = frontier.NativeFrontier(mpc, 1., 1.,
                                  solvers=frontier.TableSolvers(table, _host_split_batch))
    got = nat.p_theta(thetas)
    nat.close()
    assert ref[-1] == (None, None, None) and got[-1] == (None, None, None)
    feasible = 0
    for (u_r, d_r, J_r), (u_g, d_g, J_g) in zip(ref, got):
        assert (d_r is None) == (d_g is None)
        if d_r is None:
            continue
        feasible += 1
        assert J_g == J_r and np.array_equal(d_g, d_r) and np.array_equal(u_g, u_r)
    assert feasible >= n - 3
    assert table.lp_solves == ref_table.lp_solves           # the same problems, step for step


def test_a_failed_round_poisons_the_handle_until_reset():
    """
    A solver call that fails inside a round leaves cells flagged PENDING in no work list: the
    handle must refuse run / export (EHM_E_INVALID, "reset first") instead of returning a silently
    incomplete tree, and give the reference tree again after ehm_frontier_reset.
    """
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a, eps_r = helpers.eps_a_rule(mpc, 0.25), 0.2
    roots, _ = helpers.roots_of(mpc)
    table = prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r)
    calls = {'n': 0, 'fail_at': 3}
    real = table.solve_slack

    def flaky(*a, **kw):
        calls['n'] += 1
        if calls['n'] == calls['fail_at']:
            raise RuntimeError('injected solver failure')
        return real(*a, **kw)
    table.solve_slack = flaky
    solvers = frontier.TableSolvers(table, _host_split_batch)
    nat = frontier.NativeFrontier(mpc, eps_a, eps_r, solvers=solvers)
    nat.add_roots(np.array(roots))
    with pytest.raises(RuntimeError, match='injected'):
        nat.run(round_cap=5)
    for call in (lambda: nat.run(round_cap=5), nat.export):
        with pytest.raises(_capi.EhmError) as err:
            call()
        assert err.value.code == _capi.EHM_E_INVALID and 'reset' in str(err.value)
    # after a reset the same handle grows the whole tree (the failure does not repeat)
    calls['fail_at'] = -1
    slow = bnb.PrefixOracle(mpc, eps_a, eps_r, table=prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r))
    got = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    frontier.grow_cells(nat, got, slow_oracle=lambda: slow, round_cap=64,
                        slow_opts=dict(handoff=False, split_batch=_host_split_batch))
    ref_orc = bnb.PrefixOracle(mpc, eps_a, eps_r, table=prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r))
    ref = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    bnb_frontier.grow_frontier(ref_orc, ref, 'ecc', handoff=False, split_batch=_host_split_batch,
                               round_cap=7)
    assert sum(_same_trees(a, b) for a, b in zip(ref, got)) > 10
    nat.close()


def _flat_nodes(flat):
    """{location: node index} of a flat export (roots: their index as the first letter)."""
    loc = {}
    stack = [(r, 'r%d.' % r) for r in range(flat['n_roots'])]
    while stack:
        k, name = stack.pop()
        loc[name] = k
        if flat['left'][k] >= 0:
            stack.append((int(flat['left'][k]), name + '0'))
            stack.append((int(flat['right'][k]), name + '1'))
    return loc


@pytest.mark.parametrize('stop_after,share', [(4, 0.5), (9, 1.0), (1, 0.34)])
def test_cells_taken_from_one_handle_and_given_to_another_grow_the_same_tree(stop_after, share):
    """
    ehm_frontier_take / ehm_frontier_give (lib/scheduler.py:498-599, 633-639: any worker grows any
    leaf): a handle is stopped after a few visits, the shallowest of its pending cells -- some
    hold a commutation, some still look for one -- move to a second handle, both run to the end,
    and the two trees merged (frontier.merge_taken) are the tree one handle grows alone: same
    nodes, same verdicts, same commutations and vertex costs.
    """
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a, eps_r = helpers.eps_a_rule(mpc, 0.25), 0.2
    roots, _ = helpers.roots_of(mpc)

    def handle():
        table = prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r)
        return frontier.NativeFrontier(mpc, eps_a, eps_r,
                                       solvers=frontier.TableSolvers(table, _host_split_batch))
    alone = handle()
    alone.add_roots(np.array(roots))
    st0 = alone.run(round_cap=5)
    ref = alone.export()
    alone.close()

    a, b = handle(), handle()
    a.add_roots(np.array(roots))
    st = a.run(round_cap=3, max_visits=stop_after)
    assert st['truncated']
    pending = a.pending()
    assert pending >= 2
    cells = a.take(max(1, int(round(share * pending))))
    m = len(cells['node'])
    assert a.pending() == pending - m
    # shallowest first, and the giver keeps them as leaves flagged REMOTE
    assert np.all(np.diff(cells['depth']) >= 0)
    mid = a.export()
    assert np.all(mid['flags'][cells['node']] & frontier.FR_REMOTE)
    assert np.all(mid['left'][cells['node']] < 0)
    # the records travel as plain arrays (what goes through the store between two ranks)
    import pickle
    cells = pickle.loads(pickle.dumps(cells))
    b.give(cells)
    assert b.pending() == m
    st_a = a.run(round_cap=5)
    st_b = b.run(round_cap=5)
    assert not st_a['truncated'] and not st_b['truncated']
    assert st_a['regions'] + st_b['regions'] == st0['regions']
    fa, fb = a.export(), b.export()
    assert np.all(fa['flags'][cells['node']] & frontier.FR_REMOTE)      # still leaves over there
    merged = frontier.merge_taken(fa, cells, fb)
    assert merged['n_nodes'] == ref['n_nodes'] and merged['n_roots'] == ref['n_roots']
    lm, lr = _flat_nodes(merged), _flat_nodes(ref)
    assert set(lm) == set(lr)
    for name, k in lr.items():
        j = lm[name]
        assert np.array_equal(ref['vertices'][k], merged['vertices'][j]), name
        assert (ref['left'][k] < 0) == (merged['left'][j] < 0), name
        assert ref['flags'][k] == merged['flags'][j], name
        assert np.array_equal(ref['sequence'][k], merged['sequence'][j]), name
        if ref['flags'][k] & frontier.FR_HAS_RECORD:
            assert np.allclose(ref['vertex_costs'][k], merged['vertex_costs'][j], atol=1e-12), name
            assert np.allclose(ref['vertex_inputs'][k], merged['vertex_inputs'][j], atol=1e-9), name
    # ... and it grafts into the reference's tree objects like any export
    got = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    assert frontier.graft(merged, mpc, got) == []
    assert sum(1 for t in got for nd, _ in t.walk() if nd.is_leaf() and
               nd.data.is_epsilon_suboptimal) == st0['regions']
    # a handle that has grown refuses cells; a cell with half a record is refused too
    with pytest.raises(_capi.EhmError, match='grown already'):
        a.give(cells)
    b.reset()
    bad = {k: v.copy() for k, v in cells.items()}
    bad['sequence'][0, -1] = -1 if bad['sequence'][0, 0] >= 0 else 0
    with pytest.raises(_capi.EhmError, match='bad mode sequence'):
        b.give(bad)
    a.close()
    b.close()


def test_handles_of_one_process_share_a_root_through_the_local_exchange():
    """frontier.LocalExchange (bench.py --host-streams): three workers, ONE group of roots -- the
    worker that claims it answers the other two between its slices; the attached tree is the tree
    one handle grows."""
    import threading
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a, eps_r = helpers.eps_a_rule(mpc, 0.25), 0.2
    roots, _ = helpers.roots_of(mpc)

    def handle():
        table = prefix_bb.CpuPrefixTable(mpc, eps_a, eps_r)
        return frontier.NativeFrontier(mpc, eps_a, eps_r,
                                       solvers=frontier.TableSolvers(table, _host_split_batch))
    ref_nat = handle()
    ref = [Tree(NodeData(vertices=np.array(R))) for R in roots]
    st0 = frontier.grow_cells(ref_nat, ref, round_cap=4)
    ref_nat.close()
    nats = [handle() for _ in range(3)]
    ex = frontier.LocalExchange(len(nats))
    todo = [[Tree(NodeData(vertices=np.array(R))) for R in roots]]     # one group: one owner
    lock = threading.Lock()
    box = dict(given=[], adopted={}, regions=0, errors=[], owner=[])

    def work(i):
        try:
            with lock:
                part = todo.pop() if todo else None
            if part is not None:
                st = frontier.grow_cells(nats[i], part, round_cap=2, slice_visits=3,
                                         between_slices=ex.serve)
                with lock:
                    box['owner'].append((i, part))
                    box['given'] += [(pc['id'], leaves) for pc, leaves in st['given_away']]
                    box['regions'] += st['regions']
            while True:
                parcel = ex.wait_for_work()
                if parcel is None:
                    return
                sub = [Tree(NodeData(vertices=R.copy())) for R in parcel['vertices']]
                st = frontier.grow_cells(nats[i], sub, cells=parcel, round_cap=2, slice_visits=3,
                                         between_slices=ex.serve)
                with lock:
                    box['adopted'][parcel['id']] = dict(
                        trees=sub, given=[(pc['id'], lv) for pc, lv in st['given_away']])
                    box['regions'] += st['regions']
        except BaseException as e:
            box['errors'].append(e)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not box['errors'], box['errors']
    assert len(box['adopted']) >= 2 and box['regions'] == st0['regions']
    n = frontier.attach_adopted(box['given'], box['adopted'])
    assert n == sum(len(a['trees']) for a in box['adopted'].values())
    got = box['owner'][0][1]
    assert sum(_same_trees(a, b) for a, b in zip(ref, got)) == st0['n_nodes']
    assert not any(getattr(nd.data, 'remote', False) for t in got for nd, _ in t.walk())
    # a parcel nobody grew is an error unless the caller asks for the list
    with pytest.raises(KeyError):
        frontier.attach_adopted([(999, [])], {})
    left = []
    assert frontier.attach_adopted([(999, [got[0]])], {}, left) == 0 and left == [got[0]]
    for nat in nats:
        nat.close()
