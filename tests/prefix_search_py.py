"""
The feasibility memo and the lockstep descents of ``sequences.PrefixSearch`` as plain Python dicts
and lists -- the statement the native bookkeeping (include/ehm_search.h, csrc/ehm_search.cpp)
replaced.  Test infrastructure: ``tests/test_host_search.py`` runs both on the same solver and
compares verdicts, sequences and the partition they grow.  (This form solves a pair once per
SEARCH that asks for it; the native state solves it once per launch.)
"""

import numpy as np


class PyPrefixSearch:
    """Mix in BEFORE a table class: overrides the memo methods of ``PrefixSearch``."""

    def init_search(self):
        """State of the feasibility memo (``feasible_sets``); every table calls it once."""
        self._pid = {}          # parameter point (bytes) -> id
        self._feas = {}         # prefix -> {point id: phase-one verdict}
        self._mid_of = {}       # point id of a bisection midpoint -> ids of the edge's ends
        self._feas_n = 0
        self._code_of, self._prefix_of = {(): 0}, {0: ()}      # (PrefixSearch._code, optima_at)

    def feasible_somewhere(self, prefixes, simplices):
        """Nothing is claimed: phase one over the simplex runs for every pair."""
        return np.zeros(len(prefixes), dtype=bool)

    # -- feasibility of (prefix, point) pairs, remembered --------------------------------------
    # The searches ask the same questions again and again: the children of a node share all but
    # one of its vertices, and their descents visit the same prefixes.  Phase-one verdicts are
    # kept per prefix and point (a dict of point ids per prefix); only the pairs nothing is known
    # about go to the solver.  On the 8-dimensional cell of DESIGN.md section 3.3e that is the
    # new midpoint of every node -- one vertex in nine.
    FEAS_MEMO_LIMIT = 3000000

    def point_ids(self, points):
        """Integer ids of parameter points (by value), the keys of the feasibility memo."""
        pid = self._pid
        out = []
        for pt in np.ascontiguousarray(points, dtype=np.float64).reshape(-1, self.mpc.n_x):
            key = pt.tobytes()
            i = pid.get(key)
            if i is None:
                i = pid[key] = len(pid)
            out.append(i)
        return out

    def register_midpoints(self, mids, ends_a, ends_b):
        """
        Tell the memo that ``mids[k]`` is the midpoint of ``ends_a[k]`` and ``ends_b[k]`` (the
        bisections of the partition): a relaxation feasible at both ends is feasible at the
        midpoint -- its feasible parameters form a convex set -- and needs no LP there.
        """
        mid_of = self._mid_of
        for m, a, b in zip(self.point_ids(mids), self.point_ids(ends_a), self.point_ids(ends_b)):
            mid_of[m] = (a, b)

    def feasible_sets(self, prefixes, point_sets, ids=None):
        """
        For every k: is the relaxation of ``prefixes[k]`` feasible at EVERY point of
        ``point_sets[k]`` ((npts, p) arrays; ``ids[k]`` their ``point_ids`` if the caller has
        them)?  One batched launch for the pairs the memo does not hold.
        """
        if self._feas_n > self.FEAS_MEMO_LIMIT:
            self.init_search()
            ids = None
        memo, mid_of = self._feas, self._mid_of
        flags = np.ones(len(prefixes), dtype=bool)
        uniq, where, idx, pts, ask = [], {}, [], [], []
        for k, q in enumerate(prefixes):
            known = memo.get(q)
            if known is None:
                known = memo[q] = {}
            vid = ids[k] if ids is not None else self.point_ids(point_sets[k])
            need = []
            for t, v in enumerate(vid):
                r = known.get(v)
                if r is None:
                    ends = mid_of.get(v)
                    if ends is not None and known.get(ends[0]) and known.get(ends[1]):
                        known[v] = True         # feasible at both ends of the bisected edge
                        continue
                    need.append(t)
                elif not r:
                    flags[k] = False
                    break
            if not flags[k] or not need:
                continue
            u = where.get(q)
            if u is None:
                u = where[q] = len(uniq)
                uniq.append(q)
            ps = np.asarray(point_sets[k], dtype=np.float64).reshape(-1, self.mpc.n_x)
            for t in need:
                idx.append(u)
                pts.append(ps[t])
                ask.append((k, known, vid[t]))
        if ask:
            J = self.solve_points_idx(uniq, np.array(idx, dtype=np.int64), np.array(pts),
                                      feasibility_only=True)[0]
            ok = np.isfinite(J)
            for (k, known, v), good in zip(ask, ok):
                known[v] = bool(good)
                if not good:
                    flags[k] = False
            self._feas_n += len(ask)
        return flags

    def feasible_at_all(self, prefixes, points):
        """For every prefix: is its relaxation feasible at every one of the points?"""
        if not len(prefixes):
            return np.zeros(0, dtype=bool)
        points = np.asarray(points, dtype=np.float64).reshape(-1, self.mpc.n_x)
        vid = self.point_ids(points)
        return self.feasible_sets(list(prefixes), [points] * len(prefixes),
                                  [vid] * len(prefixes))

    def first_feasible_many(self, point_sets, excludes=None):
        """
        For every point set: the first mode sequence, in enumeration order, that is feasible at
        every point of it (V_R's canonical answer, lib/oracle.py:175-218), None if there is
        none.  Depth-first in lexicographic order -- a prefix whose relaxation is infeasible at a
        point is not extended --, all descents in lockstep: one ``feasible_sets`` call per step.
        ``excludes[k]``: full sequences to skip (the reference's blacklist, lib/oracle.py:198).
        """
        n_modes, N = self.mpc.delta_size, self.mpc.N
        n = len(point_sets)
        point_sets = [np.asarray(ps, dtype=np.float64).reshape(-1, self.mpc.n_x)
                      for ps in point_sets]
        ids = [self.point_ids(ps) for ps in point_sets]
        excludes = excludes or [()] * n
        stacks = [[()] for _ in range(n)]
        out = [None] * n
        active = list(range(n))
        while active:
            kids, sets, vids = [], [], []
            for j in active:
                q = stacks[j].pop()
                for i in range(n_modes):
                    kids.append(q + (i,))
                    sets.append(point_sets[j])
                    vids.append(ids[j])
            ok = self.feasible_sets(kids, sets, vids)
            still = []
            for a, j in enumerate(active):
                mine = kids[a * n_modes:(a + 1) * n_modes]
                good = [k for k, g in zip(mine, ok[a * n_modes:(a + 1) * n_modes])
                        if g and k not in excludes[j]]
                if good and len(good[0]) == N:
                    out[j] = good[0]
                    continue
                stacks[j].extend(reversed(good))
                if stacks[j]:
                    still.append(j)
            active = still
        return out


# ---- the suboptimality-test searches of many nodes: the plain-Python statement of what
# bnb_frontier.bar_e_many does with the native queues (include/ehm_search.h: ehm_search_bare_*)
import heapq                                             # noqa: E402

from explicit_hybrid_mpc_amd import bnb_frontier        # noqa: E402


def bar_e_many_py(oracle, Rs, Vs, bound=None, learned=None, incumbents=None):
    """
    ``PrefixOracle.bar_E_delta_R`` for many nodes: (list of bool, list of margins).
    ``bound[j]``: upper bounds of t* per prefix inherited from node j's ancestors (a prefix they
    refute needs no problem); ``learned[j]`` (a dict, filled here): the optima solved on node j;
    ``incumbents[j]``: a full sequence tried first (the parent's best-slack sequence) -- where
    its slack is not negative the node is open and the search is not run.
    """
    table, mpc = oracle.table, oracle.mpc
    n_modes, N = mpc.delta_size, mpc.N
    n = len(Rs)
    heaps = [[(-np.inf, ())] for _ in range(n)]
    refuted = [np.inf] * n
    closed, margin = [None] * n, [np.inf] * n
    active = list(range(n))
    oracle.calls['bar_E'] += n
    guard = [bnb_frontier.INHERIT_GUARD * (1. + float(np.max(np.abs(V)))) for V in Vs]
    seeded = [j for j in range(n) if incumbents and incumbents[j] is not None]
    if seeded:
        Rw = np.array([Rs[j] for j in seeded])
        tw, aw = table.solve_slack([incumbents[j] for j in seeded], Rw,
                                   np.array([Vs[j] for j in seeded]),
                                   table.feasible_somewhere([incumbents[j] for j in seeded], Rw))
        for w, j in enumerate(seeded):
            if learned is not None:
                learned[j][incumbents[j]] = (float(tw[w]), aw[w].copy())
            if tw[w] >= 0.:
                closed[j], margin[j] = False, abs(float(tw[w]))
        active = [j for j in active if closed[j] is None]
    while active:
        pre, Rp, Vp, kid_of, val_of, ask_of = [], [], [], {}, {}, {}
        width = bnb_frontier._batch_size(len(active), n_modes)
        for j in active:
            batch = [heapq.heappop(heaps[j])[1] for _ in range(min(width, len(heaps[j])))]
            oracle.n_expanded += len(batch)
            kids = [k for q in batch for k in bnb_frontier._kids(q, n_modes)]
            kid_of[j] = kids
            val_of[j] = bnb_frontier._lookup(kids, learned[j] if learned else None,
                                bound[j] if bound else None, guard[j], N)
            ask_of[j] = [i for i, v in enumerate(val_of[j]) if v is None]
            pre.extend(kids[i] for i in ask_of[j])
            Rp.extend([Rs[j]] * len(ask_of[j]))
            Vp.extend([Vs[j]] * len(ask_of[j]))
        oracle.n_inherited += sum(len(kid_of[j]) - len(ask_of[j]) for j in active)
        if pre:
            Rp = np.array(Rp)
            t, _ = table.solve_slack(pre, Rp, np.array(Vp), table.feasible_somewhere(pre, Rp))
        pos, still = 0, []
        for j in active:
            kids, vals = kid_of[j], val_of[j]
            for i in ask_of[j]:
                vals[i] = (float(t[pos]), None)
                if learned is not None:
                    learned[j][kids[i]] = vals[i]
                pos += 1
            for q, (tq, _) in zip(kids, vals):
                if not tq >= 0.:
                    refuted[j] = min(refuted[j], abs(tq))
                elif len(q) == N:
                    closed[j], margin[j] = False, abs(tq)
                    break
                else:
                    heapq.heappush(heaps[j], (-tq, q))
            if closed[j] is None:
                if heaps[j]:
                    still.append(j)
                else:
                    closed[j], margin[j] = True, refuted[j]
        active = still
    return closed, margin


