"""
Search over mode PREFIXES on the device: which mode sequences can be feasible on a region of the
parameter space at all.

The reference hands its mixed-integer oracles to a branch-and-bound solver (lib/oracle.py:42-46,
89-102, lib/global_vars.py:25), which never looks at most of the delta_size^N commutations.  The
device engine (csrc/ehm_hybrid.h) enumerates a commutation TABLE of at most 256 entries; for an
instance like BASELINE.json's configs[4] (4 modes, N = 8: 65 536 sequences) the table of a
region is found here first:

    the rows of step k depend on the modes of the steps < k only, so the relaxation shared by
    every sequence with a given prefix (``PWAMPC.condense_prefix``) is one more block of the
    commutation table; a prefix whose relaxation is infeasible on the region kills all its
    completions.  Level by level, every surviving prefix is extended by every mode, the new
    relaxations are written into table slots (``ehm_problem_update_blocks``) and ONE batched
    phase-one launch over the region's simplices decides them (``ehm_simplex_idx_batch``).

``feasible_sequences`` prunes by feasibility only.  Where the inputs can steer the state into
every mode region (examples.pwa4_mpc) that prunes nothing, and ``relevant_sequences`` prunes by
COST as the branch-and-bound does: with U an upper bound of the optimal cost on the region (the
largest vertex cost of one sequence that is feasible at every vertex -- fixed-sequence costs are
convex), a prefix whose relaxation costs more than U everywhere on the region (its minimum over
the region, one LP) cannot be the start of

  * the minimiser of P_theta at any point of the region (its cost there exceeds U, the
    incumbent's does not),
  * the best-slack commutation of bar_E / bar_D on any simplex inside the region: where a pruned
    sequence beats the interpolated vertex costs by some margin the incumbent (feasible on the
    whole region, so a candidate of bar_D everywhere in it) beats them by more,

so P_theta, bar_E and bar_D restricted to the region, or to any simplex inside it, return on the
surviving table what they return on the full enumeration.  V_R is a FEASIBILITY problem (any
commutation feasible at the vertices; canonical rule: the first in enumeration order), its
answer is not a matter of cost: the table therefore also carries the first sequence, in
enumeration order, that is feasible at every vertex of the region (a lexicographic descent over
prefixes, 4 x (p+1) phase-one problems per level).  ecc on the region then adopts the commutation
it would adopt on the full enumeration, and everything below is lcss -- the partition of the
region on its table is the partition the full enumeration gives.  The surviving full sequences are the region's
commutation table (``PWAMPC.restrict``); the usual ``Oracle`` / partition engine then run on it.
Large regions keep more than the engine's 256 entries (the counts per level are reported); the
functions say so instead of truncating silently -- DESIGN.md section 7c states what the top of
a config-5 tree needs beyond this.
"""

import numpy as np

from .mpc_library import CanonicalLP
from . import engine

FEAS_TOL = 1e-8          # EHM_FEAS_TOL of csrc/ehm_capi.hip


class TableTooLarge(ValueError):
    """More prefixes survive on the region than the requested table holds."""


class NoIncumbent(ValueError):
    """No sequence feasible at every vertex of the region was found to bound the cost with."""


class PrefixSearch:
    """Searches over prefixes written on the three pair solvers of a table (``solve_points``,
    ``solve_min``, ``solve_slack``); the device table below provides them."""

    def init_search(self):
        """State of the feasibility memo (``feasible_sets``); every table calls it once."""
        self._pid = {}          # parameter point (bytes) -> id
        self._feas = {}         # prefix -> {point id: phase-one verdict}
        self._mid_of = {}       # point id of a bisection midpoint -> ids of the edge's ends
        self._feas_n = 0

    def solve_points_idx(self, uniq, idx, thetas, feasibility_only=False):
        """``solve_points`` with the prefixes given as indices into the list ``uniq`` (the device
        table turns them into slots without touching the pairs one by one)."""
        return self.solve_points([uniq[i] for i in idx], thetas, feasibility_only)

    def min_cost_on(self, prefixes, simplices):
        """
        For every prefix: the minimum of its relaxation's optimal cost over the union of the
        simplices (+inf where it is infeasible everywhere).
        """
        simplices = np.asarray(simplices, dtype=np.float64)
        ns = simplices.shape[0]
        if not len(prefixes):
            return np.zeros(0)
        pairs = [q for q in prefixes for _ in range(ns)]
        J = self.solve_min(pairs, np.tile(simplices, (len(prefixes), 1, 1)))
        return J.reshape(len(prefixes), ns).min(axis=1)

    def vertex_costs(self, sequence, points):
        """Optimal cost of one full sequence at every point (+inf where infeasible)."""
        points = np.asarray(points, dtype=np.float64)
        return self.solve_points([tuple(sequence)] * points.shape[0], points)[0]

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

    def first_feasible(self, points, exclude=()):
        """``first_feasible_many`` for one point set."""
        return self.first_feasible_many([points], [exclude])[0]

    def feasible_on(self, prefixes, simplices):
        """For every prefix: is its relaxation feasible on at least one of the simplices?"""
        return np.isfinite(self.min_cost_on(prefixes, simplices))


class PrefixTable(PrefixSearch):
    """
    A device problem whose commutation table holds prefix relaxations, written on demand:
    ``slots`` blocks in HBM, a map prefix -> slot of what is loaded, and a host cache of
    condensed blocks.  The three ``solve_*`` methods take PAIRS (prefix k with point / simplex
    k) and run them as batched launches; infeasible pairs come back as +-inf.
    """

    BLOCK_CACHE = 8192         # condensed (G, w, S) kept on the host (about 150 KB each at N = 8)

    def __init__(self, mpc, slots=1024, device=0, eps_a=1., eps_r=1.):
        self.mpc = mpc
        G0, w0, S0 = mpc.condense_prefix((0,) * mpc.N)
        nU = mpc.N * mpc.n_u
        c = np.concatenate([np.zeros(nU), np.ones(2 * mpc.N)])
        can = CanonicalLP(np.repeat(G0[None], slots, axis=0), np.repeat(w0[None], slots, axis=0),
                          np.repeat(S0[None], slots, axis=0), c,
                          np.zeros((slots, mpc.delta_size * mpc.N)), mpc.n_u, mpc.N,
                          mpc.delta_size)
        self.slots = slots
        self.device = int(device)
        self.gp = engine.GpuProblem(can, eps_a, eps_r, device=device)
        self.lp_solves = 0
        self.blocks_loaded = 0
        self._slot_of = {}
        self._blocks = {}
        self.init_search()

    def close(self):
        self.gp.close()

    def set_eps(self, eps_a, eps_r):
        self.gp.set_eps(eps_a, eps_r)

    # -- the table ----------------------------------------------------------------------------
    def _block(self, prefix):
        blk = self._blocks.get(prefix)
        if blk is None:
            if len(self._blocks) >= self.BLOCK_CACHE:
                self._blocks.clear()
            blk = self._blocks[prefix] = self.mpc.condense_prefix(prefix)
        return blk

    def _ensure(self, prefixes):
        """Slots of ``prefixes`` (at most ``slots`` distinct ones), loading what is missing."""
        missing = [q for q in dict.fromkeys(prefixes) if q not in self._slot_of]
        if missing:
            if len(self._slot_of) + len(missing) > self.slots:
                self._slot_of.clear()
                missing = list(dict.fromkeys(prefixes))
            first = len(self._slot_of)
            blocks = [self._block(q) for q in missing]
            self.gp.update_blocks(first, np.stack([b[0] for b in blocks]),
                                  np.stack([b[1] for b in blocks]),
                                  np.stack([b[2] for b in blocks]))
            for k, q in enumerate(missing):
                self._slot_of[q] = first + k
            self.blocks_loaded += len(missing)
        return np.array([self._slot_of[q] for q in prefixes], dtype=np.int32)

    def _chunks(self, prefixes):
        """Split pair indices so that each part names at most ``slots`` distinct prefixes."""
        prefixes = [tuple(q) for q in prefixes]
        order = {}
        for q in prefixes:
            order.setdefault(q, len(order))
        part = np.array([order[q] // self.slots for q in prefixes], dtype=np.int64)
        for c in range(int(part.max()) + 1 if len(prefixes) else 0):
            idx = np.flatnonzero(part == c)
            yield idx, self._ensure([prefixes[k] for k in idx])

    # -- pair solvers -------------------------------------------------------------------------
    def solve_points_idx(self, uniq, idx, thetas, feasibility_only=False):
        idx = np.asarray(idx, dtype=np.int64)
        thetas = np.asarray(thetas, dtype=np.float64).reshape(idx.size, -1)
        J = np.full(idx.size, np.inf)
        u0 = np.zeros((idx.size, self.mpc.n_u))
        for c0 in range(0, len(uniq), self.slots):
            part = uniq[c0:c0 + self.slots]
            sel = np.flatnonzero((idx >= c0) & (idx < c0 + len(part))) if len(uniq) > self.slots \
                else np.arange(idx.size)
            if not sel.size:
                continue
            slot = self._ensure(part)[idx[sel] - c0]
            tau = self.gp.point_idx(thetas[sel], slot, feas=True)[0]
            self.lp_solves += sel.size
            ok = tau <= FEAS_TOL
            if feasibility_only:
                J[sel[ok]] = 0.
            elif ok.any():
                Jk, uk, _ = self.gp.point_idx(thetas[sel[ok]], slot[ok])
                self.lp_solves += int(ok.sum())
                J[sel[ok]] = Jk
                u0[sel[ok]] = uk
        return J, u0

    def solve_points(self, prefixes, thetas, feasibility_only=False):
        """(J, u0): optimal cost (+inf: infeasible) and first input of prefix k at point k."""
        thetas = np.asarray(thetas, dtype=np.float64).reshape(len(prefixes), -1)
        J = np.full(len(prefixes), np.inf)
        u0 = np.zeros((len(prefixes), self.mpc.n_u))
        for idx, slot in self._chunks(prefixes):
            tau = self.gp.point_idx(thetas[idx], slot, feas=True)[0]
            self.lp_solves += idx.size
            ok = tau <= FEAS_TOL
            if feasibility_only:
                J[idx[ok]] = 0.
            elif ok.any():
                Jk, uk, _ = self.gp.point_idx(thetas[idx[ok]], slot[ok])
                self.lp_solves += int(ok.sum())
                J[idx[ok]] = Jk
                u0[idx[ok]] = uk
        return J, u0

    def solve_min(self, prefixes, simplices):
        """Minimum over simplex k of the optimal cost of prefix k (+inf: infeasible on it)."""
        simplices = np.asarray(simplices, dtype=np.float64)
        J = np.full(len(prefixes), np.inf)
        for idx, slot in self._chunks(prefixes):
            tau = self.gp.simplex_idx(simplices[idx], slot, mode=2)[0]
            self.lp_solves += idx.size
            ok = tau <= FEAS_TOL
            if ok.any():
                J[idx[ok]] = self.gp.simplex_idx(simplices[idx[ok]], slot[ok], mode=0)[0]
                self.lp_solves += int(ok.sum())
        return J

    def solve_slack(self, prefixes, simplices, vbars):
        """(t*, alpha) of the suboptimality test of prefix k on simplex k (-inf: infeasible)."""
        simplices = np.asarray(simplices, dtype=np.float64)
        vbars = np.asarray(vbars, dtype=np.float64)
        t = np.full(len(prefixes), -np.inf)
        alpha = np.zeros((len(prefixes), simplices.shape[1]))
        for idx, slot in self._chunks(prefixes):
            tau = self.gp.simplex_idx(simplices[idx], slot, mode=2)[0]
            self.lp_solves += idx.size
            ok = tau <= FEAS_TOL
            if ok.any():
                tk, ak, _ = self.gp.simplex_idx(simplices[idx[ok]], slot[ok], mode=1,
                                                Vbar=vbars[idx[ok]])
                self.lp_solves += int(ok.sum())
                t[idx[ok]] = tk
                alpha[idx[ok]] = ak
        return t, alpha


def feasible_sequences(mpc, simplices, max_sequences=256, slots=1024, device=0):
    """
    The mode sequences that can be feasible somewhere on the union of ``simplices``
    ((n, p+1, p) vertex arrays), by breadth-first search over prefixes on the device.
    Returns (sorted list of sequences, info); raises ValueError when more than
    ``max_sequences`` prefixes survive a level (the region's table would not fit the engine).
    info: prefixes alive per level, phase-one problems solved, sequences of the full enumeration.
    """
    table = PrefixTable(mpc, slots=slots, device=device)
    try:
        alive = [()]
        levels = []
        for _ in range(mpc.N):
            cand = [pre + (i,) for pre in alive for i in range(mpc.delta_size)]
            ok = table.feasible_on(cand, simplices)
            alive = [pre for pre, good in zip(cand, ok) if good]
            levels.append(len(alive))
            if len(alive) > max_sequences:
                raise TableTooLarge('%d prefixes of length %d are feasible on the region: its '
                                    'commutation table exceeds %d entries (a region around the '
                                    'origin keeps every sequence)' %
                                    (len(alive), len(levels), max_sequences))
        info = dict(alive_per_level=levels, phase_one_problems=table.lp_solves,
                    enumeration=mpc.delta_size ** mpc.N)
        return sorted(alive), info
    finally:
        table.close()


TIE_TOL = 1e-6           # the canonical tie tolerance: ties must survive the pruning


def relevant_sequences(mpc, simplices, max_sequences=256, slots=1024, device=0, table=None,
                       extra=()):
    """
    The mode sequences that can matter to an oracle on the union of ``simplices`` (module
    docstring): breadth-first search over prefixes on the device, pruned by the cost bound.
    Returns (sorted sequences, info); info carries the incumbent sequence and bound U, the
    prefixes alive per level, V_R's first feasible sequence (part of the table), the LPs solved
    and the size of the full enumeration.  Raises
    NoIncumbent when the greedy dive's sequence is infeasible at a vertex, TableTooLarge when
    more than ``max_sequences`` prefixes survive a level (both are ValueErrors).  ``extra``:
    full sequences to add to the table (a node's current commutation).
    """
    own = table is None
    if own:
        table = PrefixTable(mpc, slots=slots, device=device)
    try:
        simplices = np.asarray(simplices, dtype=np.float64)
        start = table.lp_solves
        # incumbent: follow the cheapest relaxation down to a full sequence
        dive = ()
        for _ in range(mpc.N):
            kids = [dive + (i,) for i in range(mpc.delta_size)]
            cost = table.min_cost_on(kids, simplices)
            if not np.isfinite(cost).any():
                raise NoIncumbent('the cheapest-relaxation descent ends at prefix %s' % (dive,))
            dive = kids[int(np.argmin(cost))]
        verts = np.unique(simplices.reshape(-1, simplices.shape[-1]), axis=0)
        U = float(table.vertex_costs(dive, verts).max())
        if not np.isfinite(U):
            raise NoIncumbent('the incumbent sequence %s is infeasible at a vertex of the '
                              'region; split the region first' % (dive,))
        bound = U + TIE_TOL * (1. + abs(U))
        alive, levels = [()], []
        for _ in range(mpc.N):
            cand = [pre + (i,) for pre in alive for i in range(mpc.delta_size)]
            cost = table.min_cost_on(cand, simplices)
            alive = [pre for pre, c in zip(cand, cost) if c <= bound]
            levels.append(len(alive))
            if len(alive) > max_sequences:
                raise TableTooLarge('%d prefixes of length %d survive on the region (levels so '
                                    'far %s): its commutation table exceeds %d entries' %
                                    (len(alive), len(levels), levels, max_sequences))
        first = table.first_feasible(verts)          # V_R's answer on the region (docstring)
        info = dict(incumbent=dive, upper_bound=U, alive_per_level=levels, first_feasible=first,
                    lp_solves=table.lp_solves - start, enumeration=mpc.delta_size ** mpc.N)
        return sorted(set(alive) | {first} | {tuple(q) for q in extra}), info
    finally:
        if own:
            table.close()
