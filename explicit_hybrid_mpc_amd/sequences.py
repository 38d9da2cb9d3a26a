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

import ctypes

from .mpc_library import CanonicalLP
from . import engine, _capi
from ._capi import ptr

FEAS_TOL = 1e-8          # EHM_FEAS_TOL of csrc/ehm_capi.hip
SLIVER_TOL = 1e-7        # EHM_SLIVER_TOL there


INACCURATE_MAX_DECADE = 6     # a stalled solve is usable up to merit 1e6 (value good to ~1e-4)


def decisive_inaccurate(status, values):
    """
    Which stalled solves still ANSWER a sign question.  The batched kernels report a solve that
    did not reach its tolerances as ``1 | (decade << 8)`` (csrc/ehm_dev.h, ``ehm_status_word``):
    it stalled at merit <= 10^decade, i.e. its optimum is good to about 10^(decade - 10),
    relative.  Where the value is a hundred such error bars away from zero its sign -- all the
    suboptimality test asks of a full sequence (lib/oracle.py:89-97, 285-309) -- is not in doubt,
    and the value is taken as the reference takes OPTIMAL_INACCURATE (lib/oracle.py:440-442).
    """
    status = np.asarray(status, dtype=np.int64)
    values = np.asarray(values, dtype=np.float64)
    dec = (status >> 8) & 0xff
    err = 100. * 10. ** (dec - 10.) * (1. + np.abs(values))
    return ((status & 1) == 1) & (dec <= INACCURATE_MAX_DECADE) & np.isfinite(values) & \
        (np.abs(values) > err)


class TableTooLarge(ValueError):
    """More prefixes survive on the region than the requested table holds."""


class NoIncumbent(ValueError):
    """No sequence feasible at every vertex of the region was found to bound the cost with."""


class PrefixSearch:
    """Searches over prefixes written on the three pair solvers of a table (``solve_points``,
    ``solve_min``, ``solve_slack``); the device table below provides them.  The bookkeeping
    between two launches -- point ids, the memo of phase-one verdicts, the lockstep descents --
    is native (include/ehm_search.h, csrc/ehm_search.cpp)."""

    def init_search(self):
        """Creates the native search state (``ehm_search``); every table calls it once."""
        self._lib = _capi.load()
        h = ctypes.c_void_p()
        _capi.check_search(self._lib.ehm_search_create(self.mpc.n_x, self.mpc.delta_size,
                                                       self.mpc.N, ctypes.byref(h)))
        self._search = h
        self._code_of = {(): 0}         # prefix -> its integer code (ehm_search.h), and back
        self._prefix_of = {0: ()}
        self._feas_n = 0                # pairs solved since the verdicts were last forgotten

    def forget(self):
        """Drops everything the searches remember -- point ids, phase-one verdicts, optima -- so
        that a second run on this table solves what the first one solved (the blocks loaded in
        the table are problem data and stay)."""
        self.close_search()
        self.init_search()
        self.__dict__.pop('_optima', None)
        self.optima_solved = self.optima_asked = 0

    def close_search(self):
        h, self._search = getattr(self, '_search', None), None
        if h:
            self._lib.ehm_search_destroy(h)

    def __del__(self):
        try:
            self.close_search()
        except Exception:
            pass

    def _code(self, q):
        c = self._code_of.get(q)
        if c is None:
            c = self._code(q[:-1]) + (int(q[-1]) + 1) * (self.mpc.delta_size + 1) ** (len(q) - 1)
            self._code_of[q] = c
            self._prefix_of.setdefault(c, tuple(int(i) for i in q))
        return c

    def _prefix(self, c):
        q = self._prefix_of.get(c)
        if q is None:
            base, x, digits = self.mpc.delta_size + 1, c, []
            while x:
                digits.append(x % base - 1)
                x //= base
            q = self._prefix_of[c] = tuple(digits)
            self._code_of[q] = c
        return q

    def search_counts(self):
        """(verdicts held, points, pairs handed out, pairs shared by searches of one launch)."""
        out = np.zeros(4, dtype=np.int64)
        _capi.check_search(self._lib.ehm_search_counts(self._search, ptr(out)))
        return tuple(int(v) for v in out)

    def solve_points_idx(self, uniq, idx, thetas, feasibility_only=False, known_feasible=False):
        """``solve_points`` with the prefixes given as indices into the list ``uniq`` (the device
        table turns them into slots without touching the pairs one by one)."""
        return self.solve_points([uniq[i] for i in idx], thetas, feasibility_only,
                                 known_feasible=known_feasible)

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
        tiled = np.tile(simplices, (len(prefixes), 1, 1))
        J = self.solve_min(pairs, tiled, self.feasible_somewhere(pairs, tiled))
        return J.reshape(len(prefixes), ns).min(axis=1)

    OPTIMA_MEMO_LIMIT = 2000000

    def optima_at(self, sequences, point_sets):
        """
        (J [n, nv], u0 [n, nv, n_u]): optimal cost and first input of full sequence k at every
        point of set k ((nv, p) arrays), for pairs the caller KNOWS to be feasible (see
        ``solve_points``).  An optimum is a function of the sequence and the point: it is
        computed once and remembered by (sequence, point id) -- the cells of a partition share
        their vertices, and neighbours that adopt the same sequence ask for the same optima.
        """
        n = len(sequences)
        sets = np.asarray(point_sets, dtype=np.float64).reshape(n, -1, self.mpc.n_x)
        nv = sets.shape[1]
        J = np.empty((n, nv))
        u0 = np.empty((n, nv, self.mpc.n_u))
        if not n:
            return J, u0
        memo = self.__dict__.setdefault('_optima', {})      # (sequence, point id) -> (J, u0)
        if len(memo) > self.OPTIMA_MEMO_LIMIT:
            memo.clear()
        pid = np.asarray(self.point_ids(sets.reshape(-1, self.mpc.n_x))).reshape(n, nv)
        keys = [[(self._code(tuple(q)) << 38) | int(v) for v in row]
                for q, row in zip(sequences, pid)]
        missing = {}
        for k, row in enumerate(keys):
            for t, key in enumerate(row):
                if key not in memo and key not in missing:
                    missing[key] = (k, t)
        if missing:
            rows = np.array(list(missing.values()), dtype=np.int64)
            Ja, ua = self.solve_points([tuple(sequences[k]) for k in rows[:, 0]],
                                       sets[rows[:, 0], rows[:, 1]], known_feasible=True)
            for a, key in enumerate(missing):
                memo[key] = (float(Ja[a]), ua[a].copy())
        for k, row in enumerate(keys):
            for t, key in enumerate(row):
                J[k, t], u0[k, t] = memo[key]
        self.optima_solved = getattr(self, 'optima_solved', 0) + len(missing)
        self.optima_asked = getattr(self, 'optima_asked', 0) + n * nv
        return J, u0

    def vertex_costs(self, sequence, points):
        """Optimal cost of one full sequence at every point (+inf where infeasible)."""
        points = np.asarray(points, dtype=np.float64)
        return self.solve_points([tuple(sequence)] * points.shape[0], points)[0]

    # -- feasibility of (prefix, point) pairs, remembered --------------------------------------
    # The searches ask the same questions again and again: the children of a node share all but
    # one of its vertices, and their descents visit the same prefixes.  Phase-one verdicts are
    # kept per prefix and point; only the pairs nothing is known about go to the solver, and a
    # pair several searches of one launch ask for goes once.  On the 8-dimensional cell of
    # DESIGN.md section 3.3e that is the new midpoint of every node -- one vertex in nine, asked
    # by both children of the node in the same launch.
    FEAS_MEMO_LIMIT = 3000000

    def point_ids(self, points):
        """Integer ids of parameter points (by value), the keys of the feasibility memo."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, self.mpc.n_x)
        ids = np.empty(pts.shape[0], dtype=np.int64)
        _capi.check_search(self._lib.ehm_search_point_ids(self._search, pts.shape[0], ptr(pts),
                                                          ptr(ids)))
        return ids

    def register_midpoints(self, mids, ends_a, ends_b):
        """
        Tell the memo that ``mids[k]`` is the midpoint of ``ends_a[k]`` and ``ends_b[k]`` (the
        bisections of the partition): a relaxation feasible at both ends is feasible at the
        midpoint -- its feasible parameters form a convex set -- and needs no LP there.
        """
        m, a, b = self.point_ids(mids), self.point_ids(ends_a), self.point_ids(ends_b)
        _capi.check_search(self._lib.ehm_search_register_midpoints(self._search, m.size, ptr(m),
                                                                   ptr(a), ptr(b)))

    def _forget_if_full(self):
        if self._feas_n > self.FEAS_MEMO_LIMIT:
            _capi.check_search(self._lib.ehm_search_forget(self._search))
            self._feas_n = 0

    def _solve_pending(self, n_ask, n_prefix):
        """One phase-one launch over the pairs the native state has pending; their verdicts."""
        codes = np.empty(n_prefix, dtype=np.uint64)
        idx = np.empty(n_ask, dtype=np.int64)
        thetas = np.empty((n_ask, self.mpc.n_x))
        _capi.check_search(self._lib.ehm_search_asks(self._search, ptr(codes), ptr(idx),
                                                     ptr(thetas)))
        try:
            J = self.solve_points_idx([self._prefix(int(c)) for c in codes], idx, thetas,
                                      feasibility_only=True)[0]
        except BaseException:
            # the launch failed: its pairs are unknown again, the native state stays usable
            self._lib.ehm_search_abandon(self._search)
            raise
        self._feas_n += n_ask
        return np.ascontiguousarray(np.isfinite(J), dtype=np.uint8)

    def feasible_sets(self, prefixes, point_sets, ids=None):
        """
        For every k: is the relaxation of ``prefixes[k]`` feasible at EVERY point of
        ``point_sets[k]`` ((npts, p) arrays; ``ids[k]`` their ``point_ids`` if the caller has
        them)?  One batched launch for the pairs the memo does not hold.
        """
        n = len(prefixes)
        flags = np.ones(n, dtype=np.uint8)
        if not n:
            return flags.astype(bool)
        self._forget_if_full()
        if ids is None:
            sets = [np.asarray(ps, dtype=np.float64).reshape(-1, self.mpc.n_x)
                    for ps in point_sets]
            sizes = [ps.shape[0] for ps in sets]
            pid = self.point_ids(np.vstack(sets))
        else:
            sizes = [len(v) for v in ids]
            pid = np.ascontiguousarray(np.concatenate(ids), dtype=np.int64)
        begin = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(sizes, out=begin[1:])
        codes = np.fromiter((self._code(q) for q in prefixes), dtype=np.uint64, count=n)
        n_ask, n_prefix = ctypes.c_int64(), ctypes.c_int64()
        _capi.check_search(self._lib.ehm_search_query(
            self._search, n, ptr(codes), ptr(begin), ptr(pid), ptr(flags), ctypes.byref(n_ask),
            ctypes.byref(n_prefix)))
        if n_ask.value:
            ok = self._solve_pending(n_ask.value, n_prefix.value)
            _capi.check_search(self._lib.ehm_search_answer(self._search, ptr(ok), ptr(flags)))
        return flags.astype(bool)

    def feasible_somewhere(self, prefixes, simplices):
        """
        For every k: is the relaxation of ``prefixes[k]`` KNOWN to be feasible at a vertex of
        ``simplices[k]`` -- then it is feasible on the simplex, and the phase-one problem over the
        simplex would only repeat that.  Where nothing is held about the pair, the first vertex
        without a verdict is asked (one point problem, remembered: the cells of a partition share
        their vertices, so the neighbours that ask about the same prefix get it for nothing).
        False = not known; the caller solves phase one over the simplex for those.
        """
        n = len(prefixes)
        if not n:
            return np.zeros(0, dtype=bool)
        R = np.asarray(simplices, dtype=np.float64).reshape(n, -1, self.mpc.n_x)
        nv = R.shape[1]
        pid = self.point_ids(R.reshape(-1, self.mpc.n_x))
        codes = np.fromiter((self._code(q) for q in prefixes), dtype=np.uint64, count=n)
        ver = np.empty(n * nv, dtype=np.int8)
        rep = np.repeat(codes, nv)
        _capi.check_search(self._lib.ehm_search_peek(self._search, n * nv, ptr(rep), ptr(pid),
                                                     ptr(ver)))
        ver, pid = ver.reshape(n, nv), pid.reshape(n, nv)
        known = (ver == 1).any(axis=1)
        ask = np.flatnonzero(~known & (ver == -1).any(axis=1))
        if ask.size:
            col = (ver[ask] == -1).argmax(axis=1)
            known[ask] = self.feasible_sets([prefixes[k] for k in ask], None,
                                            [pid[k, c:c + 1] for k, c in zip(ask, col)])
        return known

    def feasible_somewhere_codes(self, codes, owner, pids):
        """``feasible_somewhere`` for pairs given as (prefix code k, simplex ``owner[k]``) with the
        point ids of every simplex's vertices in ``pids`` ((n_simplices, nv)): no per-pair work in
        the interpreter except for the pairs nothing is known about."""
        n = len(codes)
        if not n:
            return np.zeros(0, dtype=bool)
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        pid = np.ascontiguousarray(pids[owner], dtype=np.int64)
        nv = pid.shape[1]
        ver = np.empty(n * nv, dtype=np.int8)
        rep = np.repeat(codes, nv)
        flat = np.ascontiguousarray(pid.reshape(-1))
        _capi.check_search(self._lib.ehm_search_peek(self._search, n * nv, ptr(rep), ptr(flat),
                                                     ptr(ver)))
        ver = ver.reshape(n, nv)
        known = (ver == 1).any(axis=1)
        ask = np.flatnonzero(~known & (ver == -1).any(axis=1))
        if ask.size:
            col = (ver[ask] == -1).argmax(axis=1)
            known[ask] = self.feasible_sets([self._prefix(int(c)) for c in codes[ask]], None,
                                            [pid[k, c:c + 1] for k, c in zip(ask, col)])
        return known

    def solve_slack_codes(self, codes, owner, Rs, Vs, pids):
        """
        Slacks t [n] of the suboptimality test of prefix ``codes[k]`` on simplex ``Rs[owner[k]]``
        with vertex costs ``Vs[owner[k]]`` (the pairs a step of the native best-first queues asks
        for, include/ehm_search.h).  This generic form goes through ``solve_slack``; the device
        table maps codes to slots without touching the pairs one by one.
        """
        prefixes = [self._prefix(int(c)) for c in codes]
        Rp = Rs[owner]
        return self.solve_slack(prefixes, Rp, Vs[owner], self.feasible_somewhere(prefixes, Rp))[0]

    def feasible_at_all(self, prefixes, points):
        """For every prefix: is its relaxation feasible at every one of the points?"""
        if not len(prefixes):
            return np.zeros(0, dtype=bool)
        points = np.asarray(points, dtype=np.float64).reshape(-1, self.mpc.n_x)
        vid = self.point_ids(points)
        return self.feasible_sets(list(prefixes), None, [vid] * len(prefixes))

    def first_feasible_many(self, point_sets, excludes=None):
        """
        For every point set: the first mode sequence, in enumeration order, that is feasible at
        every point of it (V_R's canonical answer, lib/oracle.py:175-218), None if there is
        none.  Depth-first in lexicographic order -- a prefix whose relaxation is infeasible at a
        point is not extended --, all descents in lockstep (``ehm_search_descent_*``): the native
        state walks every descent as far as remembered verdicts carry it, and one phase-one
        launch per step decides the pairs nothing is known about.
        ``excludes[k]``: full sequences to skip (the reference's blacklist, lib/oracle.py:198).
        """
        n, N = len(point_sets), self.mpc.N
        if not n:
            return []
        self._forget_if_full()
        sets = [np.asarray(ps, dtype=np.float64).reshape(-1, self.mpc.n_x) for ps in point_sets]
        begin = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([ps.shape[0] for ps in sets], out=begin[1:])
        pid = self.point_ids(np.vstack(sets))
        ex_begin = ex_codes = None
        if excludes is not None and any(len(e) for e in excludes):
            ex_begin = np.zeros(n + 1, dtype=np.int64)
            np.cumsum([len(e) for e in excludes], out=ex_begin[1:])
            ex_codes = np.array([self._code(tuple(q)) for e in excludes for q in e],
                                dtype=np.uint64)
        _capi.check_search(self._lib.ehm_search_descent_begin(
            self._search, n, ptr(begin), ptr(pid), ptr(ex_begin), ptr(ex_codes)))
        n_ask, n_prefix = ctypes.c_int64(), ctypes.c_int64()
        ok = None
        while True:
            _capi.check_search(self._lib.ehm_search_descent_step(
                self._search, ptr(ok), ctypes.byref(n_ask), ctypes.byref(n_prefix)))
            if not n_ask.value:
                break
            ok = self._solve_pending(n_ask.value, n_prefix.value)
        seq = np.empty((n, N), dtype=np.int32)
        _capi.check_search(self._lib.ehm_search_descent_result(self._search, ptr(seq), None))
        return [tuple(int(i) for i in row) if row[0] >= 0 else None for row in seq]

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
        # solves the device reported as stalled (status != 0: neither converged nor within the
        # acceptance band of csrc/ehm_ipm3.h): an optimum becomes +inf (the callers' blacklist /
        # retry paths, lib/oracle.py:214-218, 406-414, see it as a failed vertex solve), a slack
        # or a minimum over a simplex raises, a phase one whose last iterate is not feasible raises
        # a prefix of this many steps is a full sequence (its values are answers, not bounds);
        # None: every block of this table is a relaxation (the short table of SplitPrefixTable)
        self.full_length = mpc.N
        self.stalled = 0
        self.stalled_relaxations = 0    # of them: relaxations answered by "no information"
        self.slivers = 0                # of them: interior-free pairs, counted as infeasible
        # problems solved by (kind, prefix length): kind 0 point phase one, 1 point optimum,
        # 2 simplex phase one, 3 minimum over a simplex, 4 suboptimality test
        self.by_length = np.zeros((5, mpc.N + 1), dtype=np.int64)
        self._slot_len = np.zeros(slots, dtype=np.int64)
        self._slot_of = {}
        self._blocks = {}
        self.init_search()

    def close(self):
        self.close_search()
        self.gp.close()

    def set_eps(self, eps_a, eps_r):
        self.gp.set_eps(eps_a, eps_r)

    def device_stats(self):
        """ehm_stats of the table's problem handle (solves, iterations, kernel seconds of the
        batched launches)."""
        return self.gp.stats()

    def reset_counts(self):
        self.by_length[:] = 0

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
                self._slot_len[first + k] = len(q)
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
    def solve_points_idx(self, uniq, idx, thetas, feasibility_only=False, known_feasible=False):
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
            self._solve_chunk(sel, slot, thetas, J, u0, feasibility_only, known_feasible)
        return J, u0

    def _solve_chunk(self, sel, slot, thetas, J, u0, feasibility_only, known_feasible):
        """Phase one, then the optimum where it is feasible, of the pairs ``sel`` (their
        prefixes loaded in ``slot``); with ``known_feasible`` phase one is not run."""
        if known_feasible:
            ok = np.ones(sel.size, dtype=bool)
        else:
            tau, _, st = self.gp.point_idx(thetas[sel], slot, feas=True)
            self.lp_solves += sel.size
            self._tally(0, slot)
            ok = self._phase_one_verdict(tau, st)
        if feasibility_only:
            J[sel[ok]] = 0.
        elif ok.any():
            Jk, uk, st = self.gp.point_idx(thetas[sel[ok]], slot[ok])
            self.lp_solves += int(ok.sum())
            self._tally(1, slot[ok])
            bad = st != 0
            self.stalled += int(bad.sum())
            J[sel[ok]] = np.where(bad, np.inf, Jk)      # a failed solve, not an optimum
            u0[sel[ok]] = uk

    def _tally(self, kind, slot):
        self.by_length[kind] += np.bincount(self._slot_len[np.asarray(slot)],
                                            minlength=self.by_length.shape[1])

    def _phase_one_verdict(self, tau, status):
        """Feasible <=> tau* <= FEAS_TOL.  A stalled phase one whose last iterate already is
        feasible (tau <= FEAS_TOL at a primal-feasible iterate) proves feasibility; one that is not
        proves nothing -- it must not pass for "infeasible" and silently prune a prefix."""
        ok = tau <= FEAS_TOL
        unknown = (status != 0) & ~ok
        self.stalled += int((status != 0).sum())
        if unknown.any():
            from .oracle import SolverError
            raise SolverError('%d phase-one problem(s) stalled above the feasibility tolerance '
                              '(smallest tau %.3g): verdict unknown' %
                              (int(unknown.sum()), float(np.min(tau[unknown]))))
        return ok

    def _settle_stalled(self, values, status, simplices, slot, lens, what, infeasible_value,
                        unknown_value, full_is_error):
        """
        An optimisation over a simplex that stalled.  If the pair was taken as feasible without
        its phase one (``known_feasible``) the simplex may be an interior-free sliver of the
        relaxation's feasible set -- phase one decides (the rule of csrc/ehm_hybrid.h).  A stalled
        problem that IS feasible has no usable value.  Where the value only serves as a bound
        (every relaxation = proper prefix; minima in a region-table search) ``unknown_value`` --
        the bound that prunes nothing -- is always valid: the searches expand the prefix's
        children instead.  Where it is an answer (a full sequence's slack, the minimum of
        in_variability_ball): SolverError, never a silent verdict.
        """
        bad = np.flatnonzero(status != 0)
        if bad.size:
            self.stalled += int(bad.size)
            tau, _, st = self.gp.simplex_idx(simplices[bad], slot[bad], mode=2)
            self.lp_solves += int(bad.size)
            # phase one minimises the largest row violation tau (>= -1): an optimum within
            # SLIVER_TOL of zero means the relaxation's feasible set meets the simplex in a set
            # WITHOUT interior (a vertex, a face) -- the interior-point solve had nothing to
            # converge in.  Such a pair counts as infeasible on the simplex, the rule of the
            # enumerating engine (ehm_counters.slivers, csrc/ehm_capi.hip) and what a simplex
            # solver at its tolerances reports.
            sliver = (st == 0) & (tau >= -SLIVER_TOL)
            self.slivers += int(sliver.sum())
            feasible = ~sliver & ((tau <= FEAS_TOL) | (st != 0))
            full = (np.asarray(lens)[bad] == self.full_length) if self.full_length is not None \
                else np.zeros(bad.size, dtype=bool)
            keep = np.zeros(bad.size, dtype=bool)
            if full_is_error and what == 'suboptimality-test':
                # an answer that stalled close to its optimum and far from zero is an answer
                keep = feasible & full & decisive_inaccurate(np.asarray(status)[bad], values[bad])
                self.accepted_inaccurate = getattr(self, 'accepted_inaccurate', 0) + int(keep.sum())
            if full_is_error and (feasible & full & ~keep).any():
                from .oracle import SolverError
                raise SolverError('%d %s problem(s) of full mode sequences did not converge on '
                                  'the device' % (int((feasible & full & ~keep).sum()), what))
            values = values.copy()
            values[bad] = np.where(keep, values[bad],
                                   np.where(feasible, unknown_value, infeasible_value))
            self.stalled_relaxations += int((feasible & ~keep).sum())
        return values

    def solve_points(self, prefixes, thetas, feasibility_only=False, known_feasible=False):
        """
        (J, u0): optimal cost (+inf: infeasible) and first input of prefix k at point k.
        ``known_feasible``: the caller holds a proof that every pair is feasible -- a remembered
        phase-one verdict, or convexity (the point is a convex combination of points the prefix
        is feasible at) -- and phase one, which would only repeat it, is skipped.
        """
        thetas = np.asarray(thetas, dtype=np.float64).reshape(len(prefixes), -1)
        J = np.full(len(prefixes), np.inf)
        u0 = np.zeros((len(prefixes), self.mpc.n_u))
        for idx, slot in self._chunks(prefixes):
            self._solve_chunk(idx, slot, thetas, J, u0, feasibility_only, known_feasible)
        return J, u0

    def solve_min(self, prefixes, simplices, known_feasible=None, exact=False):
        """Minimum over simplex k of the optimal cost of prefix k (+inf: infeasible on it);
        ``known_feasible`` as in ``solve_slack``.  A stalled solve gives -inf (as a pruning
        bound: "keep it"); ``exact``: the minima of full sequences are ANSWERS
        (in_variability_ball) -- a stalled one raises."""
        simplices = np.asarray(simplices, dtype=np.float64)
        J = np.full(len(prefixes), np.inf)
        for idx, slot in self._chunks(prefixes):
            ok = np.ones(idx.size, dtype=bool)
            todo = np.arange(idx.size) if known_feasible is None else \
                np.flatnonzero(~np.asarray(known_feasible, dtype=bool)[idx])
            if todo.size:
                tau, _, st = self.gp.simplex_idx(simplices[idx[todo]], slot[todo], mode=2)
                self.lp_solves += todo.size
                self._tally(2, slot[todo])
                ok[todo] = self._phase_one_verdict(tau, st)
            if ok.any():
                Jk, _, st = self.gp.simplex_idx(simplices[idx[ok]], slot[ok], mode=0)
                self._tally(3, slot[ok])
                # (a minimum is only ever a pruning bound: -inf keeps the prefix / the sequence)
                J[idx[ok]] = self._settle_stalled(Jk, st, simplices[idx[ok]], slot[ok],
                                                  [len(prefixes[k]) for k in idx[ok]],
                                                  'minimum-over-a-simplex', np.inf, -np.inf,
                                                  exact)
                self.lp_solves += int(ok.sum())
        return J

    def solve_slack(self, prefixes, simplices, vbars, known_feasible=None):
        """(t*, alpha) of the suboptimality test of prefix k on simplex k (-inf: infeasible);
        ``known_feasible``: mask of the pairs whose relaxation is known to be feasible on the
        simplex (``feasible_somewhere``) -- phase one runs for the others only."""
        simplices = np.asarray(simplices, dtype=np.float64)
        vbars = np.asarray(vbars, dtype=np.float64)
        t = np.full(len(prefixes), -np.inf)
        alpha = np.zeros((len(prefixes), simplices.shape[1]))
        for idx, slot in self._chunks(prefixes):
            ok = np.ones(idx.size, dtype=bool)
            todo = np.arange(idx.size) if known_feasible is None else \
                np.flatnonzero(~np.asarray(known_feasible, dtype=bool)[idx])
            if todo.size:
                tau, _, st = self.gp.simplex_idx(simplices[idx[todo]], slot[todo], mode=2)
                self.lp_solves += todo.size
                self._tally(2, slot[todo])
                ok[todo] = self._phase_one_verdict(tau, st)
            if ok.any():
                tk, ak, st = self.gp.simplex_idx(simplices[idx[ok]], slot[ok], mode=1,
                                                 Vbar=vbars[idx[ok]])
                self._tally(4, slot[ok])
                tk = self._settle_stalled(tk, st, simplices[idx[ok]], slot[ok],
                                          [len(prefixes[k]) for k in idx[ok]],
                                          'suboptimality-test', -np.inf, np.inf, True)
                self.lp_solves += int(ok.sum())
                t[idx[ok]] = tk
                alpha[idx[ok]] = ak
        return t, alpha

    def slack_by_prefix_index(self, upre, inv, simplices, vbars, known):
        """Slacks of pairs (distinct prefix ``upre[inv[k]]``, ``simplices[k]``, ``vbars[k]``);
        ``known[k]``: the relaxation is known to be feasible on the simplex (no phase one).  The
        search state is not touched: this is the launch side of ``solve_slack_codes``."""
        n = len(inv)
        t = np.full(n, -np.inf)
        lens_u = np.array([len(q) for q in upre], dtype=np.int64)
        for c0 in range(0, len(upre), self.slots):
            part = upre[c0:c0 + self.slots]
            sel = np.flatnonzero((inv >= c0) & (inv < c0 + len(part))) if len(upre) > self.slots \
                else np.arange(n)
            if not sel.size:
                continue
            slot = self._ensure(part)[inv[sel] - c0]
            ok = np.asarray(known, dtype=bool)[sel].copy()
            todo = np.flatnonzero(~ok)
            if todo.size:
                tau, _, st = self.gp.simplex_idx(simplices[sel[todo]], slot[todo], mode=2)
                self.lp_solves += todo.size
                self._tally(2, slot[todo])
                ok[todo] = self._phase_one_verdict(tau, st)
            if ok.any():
                tk, _, st = self.gp.simplex_idx(simplices[sel[ok]], slot[ok], mode=1,
                                                Vbar=vbars[sel[ok]])
                self._tally(4, slot[ok])
                tk = self._settle_stalled(tk, st, simplices[sel[ok]], slot[ok],
                                          lens_u[inv[sel[ok]]], 'suboptimality-test', -np.inf,
                                          np.inf, True)
                self.lp_solves += int(ok.sum())
                t[sel[ok]] = tk
        return t

    def solve_slack_codes(self, codes, owner, Rs, Vs, pids):
        upre, inv = _unique_prefixes(self, codes)
        known = self.feasible_somewhere_codes(codes, owner, pids)
        return self.slack_by_prefix_index(upre, inv, Rs[owner], Vs[owner], known)


def _unique_prefixes(table, codes):
    """(list of distinct prefixes as tuples, index of every pair into it)."""
    uniq, inv = np.unique(np.asarray(codes, dtype=np.uint64), return_inverse=True)
    return [table._prefix(int(c)) for c in uniq], inv.astype(np.int64)


def short_horizon(mpc, max_cols=32, max_rows=256):
    """The longest horizon k < N whose relaxation blocks fit the shared-block kernels (columns
    n_k + p + 1 <= 32, rows m_k + p + 3 <= 256: csrc/ehm_capi.hip, ehm_problem_create); 0 if none
    or if the whole model fits them anyway."""
    if mpc.cost_type != 'inf':
        return 0
    # The relaxation of a prefix as a block of the FULL model keeps the input rows and the
    # ||R u|| epigraphs of the undecided steps; they vanish from the optimum -- and the block of
    # the shorter horizon is the same problem -- exactly when u = 0 is admissible there (it then
    # costs nothing).  Otherwise the short block's optimum is lower by a constant: still a bound,
    # but not the same search, so the split is not offered.
    if mpc.Gu.shape[0] and np.any(mpc.Gu @ np.zeros(mpc.n_u) > np.asarray(mpc.gu).ravel()):
        return 0
    p = mpc.n_x
    rows_per_step = mpc.Gx.shape[0] + mpc.Gu.shape[0] + 2 * (mpc.Q.shape[0] + mpc.R.shape[0]) + \
        max((0 if r is None else r[0].shape[0]) for r in mpc.regions)
    cols_per_step = mpc.n_u + 2

    def fits(k):
        return cols_per_step * k + p + 1 <= max_cols and rows_per_step * k + p + 3 <= max_rows
    if fits(mpc.N):
        return 0
    k = mpc.N - 1
    while k >= 1 and not fits(k):
        k -= 1
    return max(k, 0)


class SplitPrefixTable(PrefixSearch):
    """
    Two device tables behind ONE search state.  The relaxation of a mode prefix of k steps
    constrains and prices the first k steps only; as a block of the full model it is a 49 x 379
    LP (config 5) of which k/N is alive, and it runs on the wide kernels.  Here every prefix of
    at most ``short`` steps is solved as a block of the SAME law with horizon ``short``
    (``PWAMPC.with_horizon``: the same problem, the same optimum, the same first input, given
    that u = 0 is an admissible input -- ``short_horizon`` checks it) -- 29
    columns x 199 rows at short = 4, a size the shared-block kernels (one wavefront per LP, the
    constant block in LDS) solve an order of magnitude faster.  On config 5 at its stated
    tolerance 94 % of the suboptimality-test problems of the searches are such prefixes
    (DESIGN.md section 3.3e).  Longer prefixes and full sequences go to the table of the full
    model.  The memo of phase-one verdicts, the point ids and the remembered optima are those of
    this object (``PrefixSearch``): they are keyed by (prefix, point), whichever table solves.
    """

    def __init__(self, mpc, short, slots=1024, device=0, eps_a=1., eps_r=1.):
        assert 1 <= short < mpc.N
        self.mpc = mpc
        self.short_len = int(short)
        self.device = int(device)
        self.slots = slots
        self.long = PrefixTable(mpc, slots=slots, device=device, eps_a=eps_a, eps_r=eps_r)
        # every prefix of <= short steps has a slot of its own: nothing is ever evicted
        n_short = sum(mpc.delta_size ** k for k in range(self.short_len + 1))
        self.short = PrefixTable(mpc.with_horizon(self.short_len), slots=max(16, n_short),
                                 device=device, eps_a=eps_a, eps_r=eps_r)
        self.short.full_length = None
        self.gp = self.long.gp          # dimensions of the full model (bench.py, hand-offs)
        self.init_search()

    def close(self):
        self.close_search()
        self.short.close()
        self.long.close()

    def set_eps(self, eps_a, eps_r):
        self.short.set_eps(eps_a, eps_r)
        self.long.set_eps(eps_a, eps_r)

    # -- what the searches and the benchmark read -----------------------------------------------
    lp_solves = property(lambda self: self.short.lp_solves + self.long.lp_solves)
    blocks_loaded = property(lambda self: self.short.blocks_loaded + self.long.blocks_loaded)
    stalled = property(lambda self: self.short.stalled + self.long.stalled)
    stalled_relaxations = property(lambda self: self.short.stalled_relaxations +
                                   self.long.stalled_relaxations)
    slivers = property(lambda self: self.short.slivers + self.long.slivers)

    @property
    def by_length(self):
        out = self.long.by_length.copy()
        out[:, :self.short.by_length.shape[1]] += self.short.by_length
        return out

    def reset_counts(self):
        self.short.by_length[:] = 0
        self.long.by_length[:] = 0

    def device_stats(self):
        a, b = self.short.gp.stats(), self.long.gp.stats()
        out = {}
        for k in a:
            out[k] = [x + y for x, y in zip(a[k], b[k])] if isinstance(a[k], list) else a[k] + b[k]
        out['short_table'] = a
        out['long_table'] = b
        return out

    # -- pair solvers: split by prefix length, delegate, merge ----------------------------------
    def _split(self, prefixes):
        is_short = np.fromiter((len(q) <= self.short_len for q in prefixes), dtype=bool,
                               count=len(prefixes))
        return np.flatnonzero(is_short), np.flatnonzero(~is_short)

    def solve_points(self, prefixes, thetas, feasibility_only=False, known_feasible=False):
        prefixes = [tuple(q) for q in prefixes]
        thetas = np.asarray(thetas, dtype=np.float64).reshape(len(prefixes), -1)
        J = np.full(len(prefixes), np.inf)
        u0 = np.zeros((len(prefixes), self.mpc.n_u))
        for table, sel in zip((self.short, self.long), self._split(prefixes)):
            if sel.size:
                J[sel], u0[sel] = table.solve_points([prefixes[k] for k in sel], thetas[sel],
                                                     feasibility_only, known_feasible)
        return J, u0

    def solve_points_idx(self, uniq, idx, thetas, feasibility_only=False, known_feasible=False):
        idx = np.asarray(idx, dtype=np.int64)
        return self.solve_points([uniq[i] for i in idx], thetas, feasibility_only, known_feasible)

    def solve_min(self, prefixes, simplices, known_feasible=None, exact=False):
        prefixes = [tuple(q) for q in prefixes]
        simplices = np.asarray(simplices, dtype=np.float64)
        kf = None if known_feasible is None else np.asarray(known_feasible, dtype=bool)
        J = np.full(len(prefixes), np.inf)
        for table, sel in zip((self.short, self.long), self._split(prefixes)):
            if sel.size:
                J[sel] = table.solve_min([prefixes[k] for k in sel], simplices[sel],
                                         None if kf is None else kf[sel], exact)
        return J

    def solve_slack(self, prefixes, simplices, vbars, known_feasible=None):
        prefixes = [tuple(q) for q in prefixes]
        simplices = np.asarray(simplices, dtype=np.float64)
        vbars = np.asarray(vbars, dtype=np.float64)
        kf = None if known_feasible is None else np.asarray(known_feasible, dtype=bool)
        t = np.full(len(prefixes), -np.inf)
        alpha = np.zeros((len(prefixes), simplices.shape[1]))
        for table, sel in zip((self.short, self.long), self._split(prefixes)):
            if sel.size:
                t[sel], alpha[sel] = table.solve_slack([prefixes[k] for k in sel], simplices[sel],
                                                       vbars[sel],
                                                       None if kf is None else kf[sel])
        return t, alpha


def _split_solve_slack_codes(self, codes, owner, Rs, Vs, pids):
    upre, inv = _unique_prefixes(self, codes)
    known = self.feasible_somewhere_codes(codes, owner, pids)       # this object's memo
    short_u = np.array([len(q) <= self.short_len for q in upre], dtype=bool)
    is_short = short_u[inv]
    t = np.full(len(inv), -np.inf)
    for table, sel, keep in ((self.short, np.flatnonzero(is_short), short_u),
                             (self.long, np.flatnonzero(~is_short), ~short_u)):
        if sel.size:
            renum = np.cumsum(keep) - 1                             # index among the kept prefixes
            t[sel] = table.slack_by_prefix_index([q for q, k in zip(upre, keep) if k],
                                                 renum[inv[sel]], Rs[owner[sel]], Vs[owner[sel]],
                                                 known[sel])
    return t


SplitPrefixTable.solve_slack_codes = _split_solve_slack_codes


def make_table(mpc, slots=1024, device=0, eps_a=1., eps_r=1., split=None):
    """The device table of a law: two tables behind one search state (``SplitPrefixTable``) when
    the full model needs the wide kernels and a shorter horizon fits the shared-block ones
    (``split`` = that horizon, None = decide here, 0 = never), else one ``PrefixTable``."""
    k = short_horizon(mpc) if split is None else int(split)
    if k >= 1:
        return SplitPrefixTable(mpc, k, slots=slots, device=device, eps_a=eps_a, eps_r=eps_r)
    return PrefixTable(mpc, slots=slots, device=device, eps_a=eps_a, eps_r=eps_r)


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
