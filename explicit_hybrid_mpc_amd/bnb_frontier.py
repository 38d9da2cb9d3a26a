"""
The searches of ``bnb.PrefixOracle`` for MANY nodes at once, and the partition driver on top of
them: all open nodes of the tree are visited together, and every step of every node's search --
one expansion of a best-first queue, one level of a lexicographic descent -- goes into the SAME
batched launch (``sequences.PrefixTable.solve_*`` take (prefix, point / simplex) pairs, so the
pairs of different nodes simply share a call).  ``bnb.grow`` pays ~40 launches per node visit;
here a visit round of thousands of nodes pays about as many in total.

Every function reproduces its one-node counterpart in ``bnb.py`` decision for decision (the CPU
tests run both on the same table); the driver visits the nodes in another order, which does not
matter -- a node's fate depends on its own record only (lib/worker.py:241-417).
"""

import ctypes
import heapq
import os
import time

import numpy as np

from . import _capi
from ._capi import ptr
from .bnb import BATCH, PLATEAU, TIE_TOL, SolverError, _rel, _set_record
from .tree import NodeData


def _kids(q, n_modes):
    return [q + (i,) for i in range(n_modes)]


def first_feasible_many(table, point_sets, excludes=None):
    """``PrefixSearch.first_feasible_many`` (sequences.py: all descents in lockstep, phase-one
    verdicts remembered per prefix and point)."""
    return table.first_feasible_many(point_sets, excludes)


def p_theta_many(oracle, thetas):
    """``PrefixOracle.P_theta`` for many parameters: list of (u0, delta, J) -- (None, None, None)
    where no sequence is feasible.  Both phases of every parameter's search run in lockstep.

    Phase two (the canonical tie-break: the first sequence in enumeration order whose cost is
    within TIE_TOL of the optimum) walks prefixes whose children phase one has mostly solved
    already -- every prefix on the way to the optimum was expanded there.  The values are kept
    per parameter, and the walk asks the device only for children it has not seen: without ties
    it needs no problem at all, where it used to re-solve 4 N of them in N dependent launches.
    Same values, same walk, same answer."""
    table, mpc = oracle.table, oracle.mpc
    n_modes, N = mpc.delta_size, mpc.N
    thetas = np.asarray(thetas, dtype=np.float64).reshape(-1, mpc.n_x)
    n = thetas.shape[0]
    oracle.calls['P_theta'] += n
    phase = [1] * n
    heaps = [[(0., ())] for _ in range(n)]
    best = [np.inf] * n
    limit = [np.inf] * n
    stacks = [None] * n
    seen = [dict() for _ in range(n)]       # child prefix -> (J, u0) as the device returned them
    out = [(None, None, None)] * n

    def cut(j):
        return best[j] - PLATEAU * _rel(best[j]) if np.isfinite(best[j]) else np.inf

    def walk(j):
        """Phase two of search j as far as the known values carry it: returns the children still
        to be solved ([] = the search has finished, ``out[j]`` is set)."""
        known = seen[j]
        while True:
            if not stacks[j]:
                raise SolverError('P_theta: the optimum found in phase one was not reproduced')
            kids = _kids(stacks[j][-1], n_modes)
            missing = [k for k in kids if k not in known]
            if missing:
                return missing
            stacks[j].pop()
            oracle.n_expanded += 1
            good = [k for k in kids if known[k][0] <= limit[j]]
            if good and len(good[0]) == N:
                jk, u = known[good[0]]
                out[j] = (u.copy(), oracle.delta_of(good[0]), float(jk))
                return []
            stacks[j].extend(reversed(good))
    active = list(range(n))
    while active:
        uniq, where, idx, owner, kid_of = [], {}, [], [], {}
        still = []
        for j in active:
            if phase[j] == 1:
                batch = []
                while heaps[j] and len(batch) < BATCH and heaps[j][0][0] < cut(j):
                    batch.append(heapq.heappop(heaps[j])[1])
                oracle.n_expanded += len(batch)
                kids = [k for q in batch for k in _kids(q, n_modes)]
            else:
                kids = walk(j)
                if not kids:
                    continue                        # finished on what phase one had solved
            still.append(j)
            kid_of[j] = kids
            for k in kids:
                u = where.get(k)
                if u is None:
                    u = where[k] = len(uniq)
                    uniq.append(k)
                idx.append(u)
            owner.extend([j] * len(kids))
        active = still
        if not active:
            break
        J, u0 = table.solve_points_idx(uniq, np.array(idx, dtype=np.int64),
                                       thetas[np.array(owner, dtype=np.int64)])
        pos, still = 0, []
        for j in active:
            kids = kid_of[j]
            Jj, uj = J[pos:pos + len(kids)], u0[pos:pos + len(kids)]
            pos += len(kids)
            known = seen[j]
            for q, jq, u in zip(kids, Jj, uj):
                known[q] = (jq, u)
            if phase[j] == 1:
                for q, jq in zip(kids, Jj):
                    if not np.isfinite(jq):
                        continue
                    if len(q) == N:
                        best[j] = min(best[j], jq)
                    else:
                        heapq.heappush(heaps[j], (jq, q))
                if not (heaps[j] and heaps[j][0][0] < cut(j)):
                    if not np.isfinite(best[j]):
                        continue                    # infeasible parameter
                    phase[j] = 2
                    limit[j] = best[j] + TIE_TOL * _rel(best[j])
                    stacks[j] = [()]
            still.append(j)
        active = still
    return out


INHERIT_GUARD = 1e-6     # an inherited bound counts only by more than INHERIT_GUARD (1 + max |V|): ten
                         # times the accuracy the device solver's optima are compared at (1e-7)
# problems a best-first step aims at per launch.  One launch costs about the latency of one LP
# whatever it holds up to a few thousand problems, so the steps of a round are wide: on config 5
# 65536 against 4096 takes the cell from 1108 launches / 4.1 s of kernels to 391 / 2.5 s at the
# same number of LPs (profiles/r4/config5_round_and_launch_sizes.txt)
LAUNCH_TARGET = int(os.environ.get('EHM_LAUNCH_TARGET', '65536'))
INHERIT_MAX = 8192       # bounds a node hands down before the non-refuting ones are dropped


def _batch_size(n_active, n_modes):
    """Prefixes one search expands per step: BATCH (bnb.py) when few searches share the launch --
    latency counts, the launch is far from full --, fewer when thousands do: the launch is full
    anyway and every prefix expanded ahead of its turn is problems solved for nothing."""
    return max(1, min(BATCH, LAUNCH_TARGET // max(1, n_active * n_modes)))



def _lookup(kids, exact, bound, guard, N):
    """
    What is already known about the suboptimality-test optimum t* of the prefixes ``kids`` on a
    node: ``exact`` holds values solved on THIS node (same simplex, same vertex costs: (t, alpha
    or None)), ``bound`` upper bounds inherited from an ancestor (docstring of ``grow_frontier``) --
    those only refute.  Returns the list of (t, alpha) or None per kid.
    """
    out = [None] * len(kids)
    for i, k in enumerate(kids):
        hit = exact.get(k) if exact else None
        if hit is not None and (hit[0] < 0. or len(k) < N or hit[1] is not None):
            out[i] = hit
        elif bound:
            tb = bound.get(k)
            if tb is not None and tb < -guard:
                out[i] = (tb, None)
    return out


def bar_e_many(oracle, Rs, Vs, bound=None, learned=None, incumbents=None):
    """
    ``PrefixOracle.bar_E_delta_R`` for many nodes: (list of bool, list of margins).
    ``bound[j]``: upper bounds of t* per prefix inherited from node j's ancestors (a prefix they
    refute needs no problem); ``learned[j]`` (a dict, filled here for the nodes that stay OPEN --
    what bar_D and the children go on with; a closed node's values have no reader): the optima
    solved on node j; ``incumbents[j]``: a full sequence tried first (the parent's best-slack
    sequence) -- where its slack is not negative the node is open and the search is not run.

    The best-first queues, the expansion of prefixes, the lookup of known values and the
    verdicts are native (include/ehm_search.h: ehm_search_bare_*); a step of all searches is one
    call that returns the (prefix code, node) pairs of ONE launch, ``solve_slack_codes`` turns
    them into slots and gathered simplices with numpy, and the answers go back in one call.  The
    plain-Python statement of the same search is tests/prefix_search_py.bar_e_many_py.
    """
    table, mpc = oracle.table, oracle.mpc
    n_modes, N = mpc.delta_size, mpc.N
    n = len(Rs)
    oracle.calls['bar_E'] += n
    if not n:
        return [], []
    lib = _capi.load()          # (the queues are native whatever keeps the table's memo)
    Rs_arr = np.ascontiguousarray(np.array(Rs, dtype=np.float64))
    Vs_arr = np.ascontiguousarray(np.array(Vs, dtype=np.float64))
    guard = np.ascontiguousarray(INHERIT_GUARD * (1. + np.max(np.abs(Vs_arr), axis=1)))
    handle = ctypes.c_void_p()
    _capi.check_search(lib.ehm_search_bare_create(n, n_modes, N, ptr(guard), ctypes.byref(handle)))
    try:
        n_active = n
        seeded = [j for j in range(n) if incumbents and incumbents[j] is not None]
        if seeded:
            Rw = Rs_arr[seeded]
            tw, aw = table.solve_slack([incumbents[j] for j in seeded], Rw, Vs_arr[seeded],
                                       table.feasible_somewhere([incumbents[j] for j in seeded],
                                                                Rw))
            for w, j in enumerate(seeded):
                if learned is not None:
                    learned[j][incumbents[j]] = (float(tw[w]), aw[w].copy())
                is_open = bool(tw[w] >= 0.)
                n_active -= is_open
                _capi.check_search(lib.ehm_search_bare_seed(
                    handle, j, int(table._code(tuple(incumbents[j]))), float(tw[w]), int(is_open)))
        for j in range(n):
            if bound and bound[j]:
                codes = np.fromiter((table._code(q) for q in bound[j]), dtype=np.uint64,
                                    count=len(bound[j]))
                tb = np.fromiter(bound[j].values(), dtype=np.float64, count=len(bound[j]))
                _capi.check_search(lib.ehm_search_bare_bounds(handle, j, codes.size, ptr(codes),
                                                              ptr(tb)))
        pids = np.asarray(table.point_ids(Rs_arr.reshape(-1, Rs_arr.shape[-1]))).reshape(
            n, Rs_arr.shape[1])
        n_ask, left = ctypes.c_int64(), ctypes.c_int64(n_active)
        while left.value > 0:
            width = _batch_size(int(left.value), n_modes)
            _capi.check_search(lib.ehm_search_bare_step(handle, width, ctypes.byref(n_ask),
                                                        ctypes.byref(left)))
            if not n_ask.value:
                break
            codes = np.empty(n_ask.value, dtype=np.uint64)
            owner = np.empty(n_ask.value, dtype=np.int32)
            _capi.check_search(lib.ehm_search_bare_asks(handle, ptr(codes), ptr(owner)))
            t = np.ascontiguousarray(table.solve_slack_codes(codes, owner.astype(np.int64), Rs_arr,
                                                             Vs_arr, pids), dtype=np.float64)
            _capi.check_search(lib.ehm_search_bare_answer(handle, ptr(t), ctypes.byref(left)))
        closed = np.empty(n, dtype=np.int8)
        margin = np.empty(n, dtype=np.float64)
        counts = np.zeros(2, dtype=np.int64)
        _capi.check_search(lib.ehm_search_bare_result(handle, ptr(closed), ptr(margin),
                                                      ptr(counts)))
        oracle.n_expanded += int(counts[0])
        oracle.n_inherited += int(counts[1])
        if learned is not None:
            cnt = ctypes.c_int64()
            for j in np.flatnonzero(closed == 0):
                _capi.check_search(lib.ehm_search_bare_learned(handle, int(j), ctypes.byref(cnt),
                                                               None, None))
                if not cnt.value:
                    continue
                codes = np.empty(cnt.value, dtype=np.uint64)
                vals = np.empty(cnt.value, dtype=np.float64)
                _capi.check_search(lib.ehm_search_bare_learned(handle, int(j), ctypes.byref(cnt),
                                                               ptr(codes), ptr(vals)))
                dst = learned[int(j)]
                for c, v in zip(codes, vals):
                    dst[table._prefix(int(c))] = (float(v), None)
        return [bool(c) for c in closed], [float(m) for m in margin]
    finally:
        lib.ehm_search_bare_destroy(handle)


def bar_d_many(oracle, Rs, Vs, deltas_ref, bound=None, learned=None, incumbents=None,
               stars=None):
    """
    ``PrefixOracle.bar_D_delta_R`` for many nodes: list of its 4-tuples.  ``bound`` / ``learned``
    as in ``bar_e_many`` (``learned[j]`` may come in holding what bar_E solved on the node: the
    same problems); a prefix solved in the first phase is not solved again in the second.
    ``incumbents[j]``: a full sequence to evaluate first (the best-slack sequence of node j's
    parent): its slack is a value the maximum cannot fall below, so every prefix whose inherited
    bound lies under it is never solved.  ``stars`` (a list, filled here): the best-slack
    sequence found per node, None where there is none.
    """
    table, mpc = oracle.table, oracle.mpc
    n_modes, N = mpc.delta_size, mpc.N
    n = len(Rs)
    oracle.calls['bar_D'] += n
    refs = [oracle.sequence_of(d) for d in deltas_ref]
    phase = [1] * n
    heaps = [[(-np.inf, ())] for _ in range(n)]
    best = [-np.inf] * n
    limit = [0.] * n
    stacks = [None] * n
    star = [None] * n                    # (sequence, alpha) once found; False = none exists
    if learned is None:
        learned = [dict() for _ in range(n)]
    guard = [INHERIT_GUARD * (1. + float(np.max(np.abs(V)))) for V in Vs]

    def floor(j):
        return max(0., best[j] + PLATEAU * _rel(best[j])) if np.isfinite(best[j]) else 0.
    vid = [table.point_ids(R) for R in Rs]                # the vertices' ids, once per node
    # warm start: the parent's best-slack sequence, evaluated on the node itself
    seeded = [j for j in range(n) if incumbents and incumbents[j] is not None
              and incumbents[j] not in learned[j]]
    if seeded:
        Rw = np.array([Rs[j] for j in seeded])
        tw, aw = table.solve_slack([incumbents[j] for j in seeded], Rw,
                                   np.array([Vs[j] for j in seeded]),
                                   table.feasible_somewhere([incumbents[j] for j in seeded], Rw))
        for w, j in enumerate(seeded):
            learned[j][incumbents[j]] = (float(tw[w]), aw[w].copy())
    warm = [j for j in range(n) if incumbents and incumbents[j] is not None
            and learned[j][incumbents[j]][0] >= 0.]
    if warm:
        ok = table.feasible_sets([incumbents[j] for j in warm], [Rs[j] for j in warm],
                                 [vid[j] for j in warm])
        for j, good in zip(warm, ok):
            if good:
                best[j] = learned[j][incumbents[j]][0]
    active = list(range(n))
    while active:
        pre, Rp, Vp, kid_of, val_of, ask_of = [], [], [], {}, {}, {}
        width = _batch_size(len(active), n_modes)
        for j in active:
            if phase[j] == 1:
                batch = []
                while heaps[j] and len(batch) < width and -heaps[j][0][0] >= floor(j):
                    batch.append(heapq.heappop(heaps[j])[1])
                oracle.n_expanded += len(batch)
                kids = [k for q in batch for k in _kids(q, n_modes)]
                need = floor(j)
            else:
                oracle.n_expanded += 1
                kids = _kids(stacks[j].pop(), n_modes)
                need = limit[j]
            kid_of[j] = kids
            val_of[j] = _lookup(kids, learned[j], bound[j] if bound else None, guard[j], N)
            if bound and bound[j]:
                # an inherited upper bound below what this phase still accepts: never solved
                for i, k in enumerate(kids):
                    if val_of[j][i] is None:
                        tb = bound[j].get(k)
                        if tb is not None and tb < need - guard[j]:
                            val_of[j][i] = (-np.inf, None)
            ask_of[j] = [i for i, v in enumerate(val_of[j]) if v is None]
            pre.extend(kids[i] for i in ask_of[j])
            Rp.extend([Rs[j]] * len(ask_of[j]))
            Vp.extend([Vs[j]] * len(ask_of[j]))
        oracle.n_inherited += sum(len(kid_of[j]) - len(ask_of[j]) for j in active)
        if pre:
            Ra = np.array(Rp)
            t, alpha = table.solve_slack(pre, Ra, np.array(Vp), table.feasible_somewhere(pre, Ra))
        pos = 0
        for j in active:
            kids, vals = kid_of[j], val_of[j]
            for i in ask_of[j]:
                full = len(kids[i]) == N
                vals[i] = learned[j][kids[i]] = (float(t[pos]), alpha[pos].copy() if full else None)
                pos += 1
        # feasibility at every vertex, for the prefixes whose slack bound is not negative
        live = [(j, i) for j in active for i, v in enumerate(val_of[j]) if v[0] >= 0.]
        dead = set()
        if live:
            ok = table.feasible_sets([kid_of[j][i] for j, i in live], [Rs[j] for j, _ in live],
                                     [vid[j] for j, _ in live])
            dead = {ji for ji, good in zip(live, ok) if not good}
        still = []
        for j in active:
            kids = kid_of[j]
            tj = [(-np.inf if (j, i) in dead else v[0]) for i, v in enumerate(val_of[j])]
            aj = [v[1] for v in val_of[j]]
            if phase[j] == 1:
                for q, tq in zip(kids, tj):
                    if not tq >= 0.:
                        continue
                    if len(q) == N:
                        best[j] = max(best[j], tq)
                    else:
                        heapq.heappush(heaps[j], (-tq, q))
                if not (heaps[j] and -heaps[j][0][0] >= floor(j)):
                    if not np.isfinite(best[j]):
                        star[j] = False
                        continue
                    phase[j] = 2
                    limit[j] = max(0., best[j] - TIE_TOL * _rel(best[j]))
                    stacks[j] = [()]
                still.append(j)
            else:
                good = [(k, a) for k, tk, a in zip(kids, tj, aj) if tk >= limit[j]]
                if good and len(good[0][0]) == N:
                    star[j] = good[0]
                    continue
                stacks[j].extend(k for k, _ in reversed(good))
                if not stacks[j]:
                    raise SolverError('bar_D: the slack found in phase one was not reproduced')
                still.append(j)
        active = still
    if stars is not None:
        stars[:] = [st[0] if st else None for st in star]
    # the winners' vertex solves and variability checks, one call each
    out = [(None, None, None, None)] * n
    win = [j for j in range(n) if star[j] and star[j][0] != refs[j]]
    if not win:
        return out
    nv = Rs[0].shape[0]
    Jv, uv = table.solve_points([star[j][0] for j in win for _ in range(nv)],
                                np.vstack([Rs[j] for j in win]))
    Jv, uv = Jv.reshape(len(win), nv), uv.reshape(len(win), nv, -1)
    thetas = np.array([star[j][1] @ Rs[j] for j in win])
    # (the node's own commutation is feasible at every vertex: no phase one over the simplex)
    Jmin = table.solve_min([refs[j] for j in win], np.array([Rs[j] for j in win]),
                           np.ones(len(win), dtype=bool))
    Jth = table.solve_points([star[j][0] for j in win], thetas)[0]
    for w, j in enumerate(win):
        if not (np.all(np.isfinite(Jv[w])) and np.isfinite(Jmin[w]) and np.isfinite(Jth[w])):
            # a failed solve: the blacklist-and-retry path of the one-node oracle
            oracle.calls['bar_D'] -= 1
            out[j] = oracle.bar_D_delta_R(Rs[j], Vs[j], deltas_ref[j])
            continue
        vx = [(uv[w, i].copy(), float(Jv[w, i]), 0.) for i in range(nv)]
        rhs = max(oracle.eps_a, oracle.eps_r * float(Jth[w]))
        small = bool(np.max(Vs[j]) - float(Jmin[w]) < rhs)
        out[j] = (oracle.delta_of(star[j][0]), thetas[w], vx, small)
    return out


def region_tables_many(oracle, Rs, commutations, Us, table_max, above=None, costs=None,
                       excess=None):
    """
    Region tables (``sequences.relevant_sequences``) of many lcss nodes: the bound U of a node
    is the largest vertex cost of ITS commutation (feasible at every vertex, so V* <= U on the
    node).  Returns a list of sorted sequence lists, None where more than ``table_max`` survive.
    ``above[j]``: minimum costs of prefix relaxations over a region that CONTAINS node j (its
    parent's, dict prefix -> cost): the minimum over the node is at least that, so a prefix the
    larger region already prices above U needs no problem; ``costs[j]`` (a dict, filled here):
    the minima found on node j, for its children; ``excess[j]`` (a list of n zeros, filled
    here): where the table did not fit, survivors / table_max at the level that broke it.
    """
    table, mpc = oracle.table, oracle.mpc
    n_modes = mpc.delta_size
    n = len(Rs)
    alive = [[()] for _ in range(n)]
    bounds = [u + TIE_TOL * (1. + abs(u)) for u in Us]
    active = list(range(n))
    for _ in range(mpc.N):
        if not active:
            break
        pre, Rp = [], []
        for j in active:
            cand = [q + (i,) for q in alive[j] for i in range(n_modes)]
            if above is not None and above[j]:
                known = above[j]
                cand = [q for q in cand if not known.get(q, -np.inf) > bounds[j]]
                oracle.n_inherited += n_modes * len(alive[j]) - len(cand)
            alive[j] = cand
            pre.extend(cand)
            Rp.extend([Rs[j]] * len(cand))
        if not pre:
            break
        Ra = np.array(Rp)
        cost = table.solve_min(pre, Ra, table.feasible_somewhere(pre, Ra))
        pos, still = 0, []
        for j in active:
            c = cost[pos:pos + len(alive[j])]
            pos += len(alive[j])
            if costs is not None:
                costs[j].update(zip(alive[j], (float(v) for v in c)))
            alive[j] = [q for q, cq in zip(alive[j], c) if cq <= bounds[j]]
            if len(alive[j]) > table_max:
                if excess is not None:
                    excess[j] = len(alive[j]) / float(table_max)
                alive[j] = None
            else:
                still.append(j)
        active = still
    out = []
    for j in range(n):
        if alive[j] is None:
            out.append(None)
            continue
        seqs = sorted(set(alive[j]) | {oracle.sequence_of(commutations[j])})
        out.append(seqs if len(seqs) <= table_max else None)
    return out


def _hand_off(oracle, nodes, table_max, engine_opts, stats, above=None, costs=None, excess=None):
    """Nodes (lcss, open) whose region table fits go to the device engine; returns the rest.
    ``above`` / ``costs``: see ``region_tables_many``."""
    from . import engine, partition
    mpc = oracle.mpc
    Rs = [np.asarray(nd.data.vertices, dtype=np.float64) for nd in nodes]
    tables = region_tables_many(oracle, Rs, [nd.data.commutation for nd in nodes],
                                [float(np.max(nd.data.vertex_costs)) for nd in nodes], table_max,
                                above, costs, excess)
    keep = []
    for k, (nd, R, seqs) in enumerate(zip(nodes, Rs, tables)):
        if seqs is None:
            stats['tables_too_large'] += 1
            keep.append(k)
            continue
        gp = engine.GpuProblem(mpc.restrict(seqs).compile(), oracle.eps_a, oracle.eps_r,
                               device=getattr(oracle.table, 'device', 0))
        try:
            init = dict(delta=np.asarray(nd.data.commutation, dtype=np.float64)[None],
                        vertex_costs=np.asarray(nd.data.vertex_costs)[None],
                        vertex_inputs=np.asarray(nd.data.vertex_inputs)[None])
            flat = gp.partition(R[None], action='lcss', init=init, **(engine_opts or {}))
        finally:
            gp.close()
        partition.graft_flat(flat, [nd])
        stats['handoffs'] += 1
        stats['handoff_nodes'] += flat.n_nodes
        stats['handoff_leaves'] += int(flat.info['n_leaves'])
        stats['regions'] = stats.get('regions', 0) + int(flat.info['n_closed'])
        stats['table_sizes'].append(len(seqs))
    return keep


def grow_frontier(oracle, branch, action='ecc', table_max=256, max_visits=None, handoff=True,
                  engine_opts=None, split_batch=None, round_cap=4096, log=None,
                  table_backoff=False, order='fifo', min_regions=None, deadline=None,
                  created_log=None):
    """
    ``bnb.grow`` with all pending nodes visited together (module docstring).  Same arguments and
    the same tree; ``round_cap`` bounds the nodes of one round, ``split_batch(R (n,p+1,p)) ->
    (S1, S2, ij)`` replaces the device bisection kernel (``engine.split_batch``).  Hand-off: an
    lcss node that bar_E leaves OPEN goes to the device engine when its region table -- bounded
    with its own commutation's vertex costs -- has at most ``table_max`` sequences (a node that
    closes at once needs no table).  ``table_backoff``: a node whose table did not fit by a
    factor f lets floor(log2 f) generations of its descendants pass before a table is tried
    again (a failed attempt costs more problems than the node's searches; the tree does not
    depend on it -- off by default until it has been measured on the device).

    ``order``: 'fifo' visits the tree level by level; 'lcss-first' takes the cells that hold a
    commutation first, deepest first (the feasibility bisection of ecc is driven only where no
    lcss work is pending); 'deepest' takes the deepest pending nodes
    first (rounds of up to ``round_cap`` nodes) -- the order of the reference's workers, whose
    ``lcss`` recursion finishes the left subtree before it touches the right one
    (lib/worker.py:403-417): subtrees are COMPLETED, so a run that is stopped early
    (``max_visits``, or ``min_regions`` closed leaves reached) has spent its visits on regions
    that are final instead of on an ever wider open frontier.  The tree of a run to completion
    does not depend on the order.  ``deadline``: a ``time.perf_counter()`` value after which no
    further round is started.

    What a node hands to its children (none of it changes a verdict, all of it saves problems):
    an ecc node the sequence that was feasible at its barycentre (tried first at theirs); an lcss
    node the optima its suboptimality-test searches solved -- a child lies inside its parent and,
    the optimal cost of a commutation being convex, the interpolant of its vertex costs lies
    below the parent's, so for every prefix t*(child) <= t*(parent) (+ the largest increase of a
    vertex cost where a commutation with larger costs is adopted): upper bounds that refute, or
    keep below the incumbent, without a solve -- and its best-slack sequence, which the
    children's searches evaluate first.
    """
    from . import engine
    split_batch = split_batch or (lambda R: engine.split_batch(R, device=getattr(
        oracle.table, 'device', 0)))
    mpc = oracle.mpc
    stats = dict(host_visits=0, rounds=0, handoffs=0, handoff_nodes=0, handoff_leaves=0,
                 table_sizes=[], tables_too_large=0, truncated=False, witness_hits=0,
                 table_attempts_skipped=0, regions=0)
    # one Tree or a list of them (the Delaunay roots of the set: their nodes share the rounds)
    # work items: (node, action, witness) -- the witness of an ecc node is a mode sequence that was
    # feasible at its parent's barycentre (None at a root)
    work = [(b, action, None) for b in (branch if isinstance(branch, (list, tuple)) else [branch])]
    depth_of = {id(nd): 0 for nd, _, _ in work}       # pending nodes only
    while work:
        if max_visits is not None and stats['host_visits'] >= max_visits:
            stats['truncated'] = True
            break
        if min_regions is not None and stats['regions'] >= min_regions:
            stats['truncated'] = True
            break
        if deadline is not None and time.perf_counter() >= deadline:    # checked between rounds
            stats['truncated'] = True
            break
        cap = round_cap if max_visits is None else min(round_cap,
                                                       max_visits - stats['host_visits'])
        if order == 'deepest':
            work.sort(key=lambda item: -depth_of[id(item[0])])      # stable: ties keep their order
        elif order == 'lcss-first':
            # cells that hold a commutation before cells that still look for one, deepest first
            work.sort(key=lambda item: (item[1] == 'ecc', -depth_of[id(item[0])]))
        batch, work = work[:cap], work[cap:]
        depth_in = {id(item[0]): depth_of.pop(id(item[0])) for item in batch}

        def push(node, act, extra, parent):
            # a node visited again (lcss after ecc, a swap in place) keeps its depth; a child is
            # one level below its parent
            depth_of[id(node)] = depth_in[id(parent)] + (0 if node is parent else 1)
            work.append((node, act, extra))
        stats['rounds'] += 1
        ecc = [nd for nd, act, _ in batch if act == 'ecc']
        witness = [wit for _, act, wit in batch if act == 'ecc']
        lcss = [nd for nd, act, _ in batch if act != 'ecc']
        # an lcss node's third entry: (upper bounds of t* per prefix proven on its ancestors, the
        # parent's best-slack sequence), or None
        bounds = [b[0] if b else None for _, act, b in batch if act != 'ecc']
        incumbents = [b[1] if b else None for _, act, b in batch if act != 'ecc']
        region_costs = [b[2] if b else None for _, act, b in batch if act != 'ecc']
        waits = [b[3] if b else 0 for _, act, b in batch if act != 'ecc']
        stats['host_visits'] += len(ecc) + len(lcss)
        if log:
            log('round %d: %d ecc + %d lcss nodes, %d waiting' %
                (stats['rounds'], len(ecc), len(lcss), len(work)))
        # (node, commutation, costs, inputs, inherited bounds) or (ecc node, None, None, witness, None)
        to_split = []
        if ecc:                             # lib/worker.py:241-283
            Rs = [np.asarray(nd.data.vertices, dtype=np.float64) for nd in ecc]
            oracle.calls['V_R'] += len(ecc)
            found = first_feasible_many(oracle.table, Rs)
            # lib/worker.py:264-266 checks the barycentre first; a sequence feasible at every
            # vertex is feasible there too (its feasible set is convex), so only the cells V_R
            # finds nothing for need the check.  It asks whether ANY sequence is feasible at the
            # barycentre: the one that was at the parent's barycentre is tried first (one
            # problem), the lexicographic descent runs only where that one fails.
            none = [k for k, s in enumerate(found) if s is None]
            oracle.calls['P_theta'] += len(none)
            centres = np.mean(np.array([Rs[k] for k in none]), axis=1) if none else np.zeros((0, 1))
            bary = [centres[i:i + 1] for i in range(len(none))]
            tried = [i for i, k in enumerate(none) if witness[k] is not None]
            held = np.zeros(len(none), dtype=bool)
            if tried:
                held[tried] = oracle.table.feasible_sets([witness[none[i]] for i in tried],
                                                         [bary[i] for i in tried])
            stats['witness_hits'] += int(held.sum())
            miss = [i for i in range(len(none)) if not held[i]]
            fresh = first_feasible_many(oracle.table, [bary[i] for i in miss])
            if any(s is None for s in fresh):
                raise RuntimeError('STOP, Theta contains infeasible regions')
            for i, s in zip(miss, fresh):
                witness[none[i]] = s
            have = [k for k, s in enumerate(found) if s is not None]
            if have:
                nv = Rs[0].shape[0]
                # the descents have just established feasibility at every vertex: no phase one;
                # an optimum a neighbouring cell has computed already is not computed again
                Jv, uv = oracle.table.optima_at([found[k] for k in have], [Rs[k] for k in have])
            for w, k in enumerate(have):
                if not np.all(np.isfinite(Jv[w])):      # lib/oracle.py:214-218: blacklist, retry
                    oracle.calls['V_R'] -= 1
                    delta, vx = oracle.V_R(Rs[k])
                    if delta is None:
                        found[k] = None
                        continue
                else:
                    delta = oracle.delta_of(found[k])
                    vx = [(uv[w, i].copy(), float(Jv[w, i]), 0.) for i in range(nv)]
                _set_record(ecc[k].data, delta, vx)
                push(ecc[k], 'lcss', None, ecc[k])
            to_split += [(ecc[k], None, None, witness[k], None)
                         for k, s in enumerate(found) if s is None]
        if lcss:                            # lib/worker.py:340-417
            Rs = [np.asarray(nd.data.vertices, dtype=np.float64) for nd in lcss]
            Vs = [np.asarray(nd.data.vertex_costs, dtype=np.float64) for nd in lcss]
            learned = [dict() for _ in lcss]
            closed, margins = bar_e_many(oracle, Rs, Vs, bounds, learned, incumbents)
            oracle.last_margin = min([oracle.last_margin] + margins)
            opened = [k for k, c in enumerate(closed) if not c]
            for k, c in enumerate(closed):
                if c:
                    lcss[k].data.is_epsilon_suboptimal = True
                    stats['regions'] += 1
            table_costs = {}
            next_wait = {k: max(0, waits[k] - 1) for k in opened}
            if handoff and opened:          # the open ones: to the engine where the table fits
                tried = [k for k in opened if not (table_backoff and waits[k] > 0)]
                stats['table_attempts_skipped'] += len(opened) - len(tried)
                table_costs = {k: dict() for k in tried}
                excess = [0.] * len(tried)
                keep = _hand_off(oracle, [lcss[k] for k in tried], table_max, engine_opts, stats,
                                 [region_costs[k] for k in tried],
                                 [table_costs[k] for k in tried], excess) if tried else []
                for k, f in zip(tried, excess):
                    if f >= 2.:
                        next_wait[k] = int(np.floor(np.log2(f)))
                kept = {tried[k] for k in keep}
                opened = [k for k in opened if k in kept or k not in table_costs]
            stars = []
            res = bar_d_many(oracle, [Rs[k] for k in opened], [Vs[k] for k in opened],
                             [lcss[k].data.commutation for k in opened],
                             [bounds[k] for k in opened], [learned[k] for k in opened],
                             [incumbents[k] for k in opened], stars) if opened else []
            for w, (k, (delta_star, theta_star, new_vx, small)) in enumerate(zip(opened, res)):
                data = lcss[k].data
                # what the children inherit: the optima solved on this node bound theirs from
                # above (on top of what the node inherited itself), and its best-slack sequence
                # is the first thing their searches evaluate
                down = dict(bounds[k]) if bounds[k] else {}
                down.update((q, v[0]) for q, v in learned[k].items())
                if len(down) > INHERIT_MAX:     # keep what refutes; the rest only orders bar_D
                    down = {q: t for q, t in down.items() if t < 0.}
                # the minima of prefix relaxations over this node, for the children's region
                # tables (a child's minimum is at least its parent's: geometry alone)
                below = dict(region_costs[k]) if region_costs[k] else {}
                below.update(table_costs.get(k, {}))
                if len(below) > INHERIT_MAX:
                    below = {}
                down = (down, stars[w], below, next_wait[k])
                if delta_star is None:
                    to_split.append((lcss[k], data.commutation, data.vertex_costs,
                                     data.vertex_inputs, down))
                    continue
                costs = np.array([v[1] for v in new_vx])
                inputs = np.array([v[0] for v in new_vx])
                # the bounds were proven against the interpolation of the OLD vertex costs; where
                # the adopted commutation's are larger the interpolant rises by at most the
                # largest increase, and so does every t*
                rise = float(np.max(costs - np.asarray(data.vertex_costs)))
                if rise > 0.:
                    down = ({q: t + rise for q, t in down[0].items()}, down[1], down[2], down[3])
                if small:                   # lib/worker.py:396-401
                    data.commutation, data.vertex_costs, data.vertex_inputs = (delta_star, costs,
                                                                                inputs)
                    push(lcss[k], 'lcss', down, lcss[k])
                else:
                    to_split.append((lcss[k], delta_star, costs, inputs, down))
        if to_split:
            Rsplit = np.array([np.asarray(nd.data.vertices, dtype=np.float64)
                               for nd, _, _, _, _ in to_split])
            S1, S2, ij = split_batch(Rsplit)
            rows = np.arange(len(to_split))
            oracle.table.register_midpoints(S1[rows, ij[:, 0]], Rsplit[rows, ij[:, 0]],
                                            Rsplit[rows, ij[:, 1]])
            with_data = [k for k, item in enumerate(to_split) if item[1] is not None]
            if with_data:
                mids = np.array([S1[k][ij[k][0]] for k in with_data])
                # feasible at both ends of the edge, so feasible at its midpoint: no phase one;
                # the cells around an edge that hold the same commutation share the optimum
                Jm, um = oracle.table.optima_at(
                    [oracle.sequence_of(to_split[k][1]) for k in with_data], mids[:, None, :])
                Jm, um = Jm[:, 0], um[:, 0]
            for w, k in enumerate(with_data):
                nd, delta, costs, inputs, down = to_split[k]
                i, j = int(ij[k][0]), int(ij[k][1])
                if not np.isfinite(Jm[w]):
                    raise SolverError('midpoint solve of the adopted commutation failed')
                in_1, in_2, co_1, co_2 = inputs.copy(), inputs.copy(), costs.copy(), costs.copy()
                in_1[i], in_2[j] = um[w], um[w]
                co_1[i], co_2[j] = Jm[w], Jm[w]
                nd.grow(NodeData(vertices=S1[k].copy(), commutation=delta, vertex_costs=co_1,
                                 vertex_inputs=in_1),
                        NodeData(vertices=S2[k].copy(), commutation=delta, vertex_costs=co_2,
                                 vertex_inputs=in_2))
                if created_log is not None:
                    # what the children were CREATED with (a later visit of a child may adopt
                    # another commutation in place, lib/worker.py:396-401): for parity tests
                    created_log[id(nd.left)] = (np.array(delta, copy=True), co_1.copy())
                    created_log[id(nd.right)] = (np.array(delta, copy=True), co_2.copy())
                push(nd.left, 'lcss', down, nd)
                push(nd.right, 'lcss', down, nd)
            for k, (nd, delta, _, wit, _) in enumerate(to_split):
                if delta is None:
                    nd.grow(NodeData(vertices=S1[k].copy()), NodeData(vertices=S2[k].copy()))
                    push(nd.left, 'ecc', wit, nd)
                    push(nd.right, 'ecc', wit, nd)
    # what the searches did not have to solve
    stats['prefixes_answered_by_inheritance'] = oracle.n_inherited
    stats['optima_asked_solved'] = (getattr(oracle.table, 'optima_asked', 0),
                                    getattr(oracle.table, 'optima_solved', 0))
    if hasattr(oracle.table, 'search_counts') and getattr(oracle.table, '_search', None):
        stats['memo_verdicts_points_pairs_shared'] = oracle.table.search_counts()
    return stats
