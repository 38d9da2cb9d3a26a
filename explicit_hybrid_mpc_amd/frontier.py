"""
The partition driver for laws whose mode sequences cannot be enumerated, native: the host side of
include/ehm_frontier.h (csrc/ehm_frontier.cpp).

``bnb_frontier.grow_frontier`` keeps the interpreter in every launch of a configs[4] cell (pair
lists, slot maps, numpy condensation of the relaxation blocks: 55-60 % of a cell's wall time,
profiles/r4/config5_host_profile*.txt).  Here ONE call grows the cells: the round loop, the
searches' bookkeeping, the condensation and the launches are C++ behind ``ehm_frontier_run``;
Python states the law once (``ehm_pwa_law``), adds the roots and reads the flat tree back.

What stays with ``bnb_frontier``: a cell whose vertex solves fail (the blacklist-and-retry paths
of lib/oracle.py:214-218, 406-414) comes back flagged and is finished by ``grow_frontier`` on a
``bnb.PrefixOracle`` -- none on configs[4] so far.  The hand-off of small regions to the
enumerating engine (``bnb_frontier._hand_off``) is not taken by the native driver.

Per-cell semantics: lib/worker.py:241-417; canonical answers of the oracles: bnb.py.
"""

import ctypes

import numpy as np

from . import _capi
from ._capi import ptr
from .tree import NodeData, Tree

FR_CLOSED, FR_HAS_RECORD, FR_OPEN, FR_PENDING, FR_NEEDS_ECC, FR_DEPTH = 1, 2, 4, 8, 16, 32


def _law_struct(mpc):
    """(ehm_pwa_law, the arrays it points into)."""
    if getattr(mpc, 'cost_type', 'inf') != 'inf':
        raise ValueError('the native driver states infinity-norm laws')
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    keep = dict(A=f(np.stack(mpc.A)), B=f(np.stack(mpc.B)), w=f(np.stack(mpc.w)),
                Gx=f(mpc.Gx), gx=f(mpc.gx), Gu=f(mpc.Gu), gu=f(mpc.gu), Q=f(mpc.Q), R=f(mpc.R))
    rows = np.array([0 if r is None else r[0].shape[0] for r in mpc.regions], dtype=np.int32)
    keep['rows'] = rows
    regs = [r for r in mpc.regions if r is not None]
    keep['Hx'] = f(np.vstack([r[0] for r in regs])) if regs else np.zeros((0, mpc.n_x))
    keep['hx'] = f(np.concatenate([np.asarray(r[1]).ravel() for r in regs])) if regs else np.zeros(0)
    dp = lambda a: a.ctypes.data_as(_capi.c_double_p)
    law = _capi.PwaLaw(n_x=mpc.n_x, n_u=mpc.n_u, n_modes=mpc.delta_size, N=mpc.N,
                       A=dp(keep['A']), B=dp(keep['B']), w=dp(keep['w']),
                       region_rows=rows.ctypes.data_as(_capi.c_int32_p),
                       Hx=dp(keep['Hx']), hx=dp(keep['hx']),
                       n_gx=mpc.Gx.shape[0], Gx=dp(keep['Gx']), gx=dp(keep['gx']),
                       n_gu=mpc.Gu.shape[0], Gu=dp(keep['Gu']), gu=dp(keep['gu']),
                       n_q=mpc.Q.shape[0], Q=dp(keep['Q']), n_r=mpc.R.shape[0], R=dp(keep['R']))
    return law, keep


def condense_native(mpc, prefix, horizon=None):
    """(G, w, S) of a prefix relaxation as the native driver writes it into its tables
    (``ehm_frontier_condense``; tests compare it with ``PWAMPC.condense_prefix``)."""
    lib = _capi.load()
    law, keep = _law_struct(mpc)
    H = mpc.N if horizon is None else int(horizon)
    pre = np.ascontiguousarray(prefix, dtype=np.int32)
    dims = np.zeros(2, dtype=np.int32)
    _capi.check_frontier(lib.ehm_frontier_condense(ctypes.byref(law), H, pre.size, ptr(pre),
                                                   ptr(dims), None, None, None))
    n, m = int(dims[0]), int(dims[1])
    G, w, S = np.empty((m, n)), np.empty(m), np.empty((m, mpc.n_x))
    _capi.check_frontier(lib.ehm_frontier_condense(ctypes.byref(law), H, pre.size, ptr(pre),
                                                   ptr(dims), ptr(G), ptr(w), ptr(S)))
    del keep
    return G, w, S


def _as(addr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not addr:
        return np.zeros(shape, dtype=dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class TableSolvers:
    """``ehm_pair_solvers`` on any object with the pair solvers of ``sequences.PrefixSearch``
    (``solve_points``, ``solve_slack``, ``_prefix``) and a bisection ``split_batch(R)`` -- how the
    CPU tests run the native driver on the CPU statement of the table."""

    def __init__(self, table, split_batch):
        self.table, self.split_batch = table, split_batch
        self.error = None
        mpc = table.mpc
        p, nv, n_u = mpc.n_x, mpc.n_x + 1, mpc.n_u

        def guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except BaseException as e:          # nothing may propagate through C frames
                    self.error = e
                    return _capi.EHM_E_NUMERIC
            return run

        def points(user, n, code, theta, fo, kf, J, u0):
            pre = [table._prefix(int(c)) for c in _as(code, (n,), np.uint64)]
            Jv, uv = table.solve_points(pre, _as(theta, (n, p), np.float64).copy(),
                                        feasibility_only=bool(fo), known_feasible=bool(kf))
            _as(J, (n,), np.float64)[:] = Jv
            if u0:
                _as(u0, (n, n_u), np.float64)[:] = uv

        def slack(user, n, code, R, V, known, t, alpha):
            pre = [table._prefix(int(c)) for c in _as(code, (n,), np.uint64)]
            kn = _as(known, (n,), np.uint8).astype(bool) if known else None
            tv, av = table.solve_slack(pre, _as(R, (n, nv, p), np.float64).copy(),
                                       _as(V, (n, nv), np.float64).copy(), kn)
            _as(t, (n,), np.float64)[:] = tv
            if alpha:
                _as(alpha, (n, nv), np.float64)[:] = av

        def minimum(user, n, code, R, known, J):
            pre = [table._prefix(int(c)) for c in _as(code, (n,), np.uint64)]
            kn = _as(known, (n,), np.uint8).astype(bool) if known else None
            _as(J, (n,), np.float64)[:] = table.solve_min(
                pre, _as(R, (n, nv, p), np.float64).copy(), kn)

        def split(user, n, R, S1, S2, ij):
            a, b, e = self.split_batch(_as(R, (n, nv, p), np.float64).copy())
            _as(S1, (n, nv, p), np.float64)[:] = a
            _as(S2, (n, nv, p), np.float64)[:] = b
            _as(ij, (n, 2), np.int32)[:] = e
        self._fns = (_capi.POINTS_FN(guard(points)), _capi.SLACK_FN(guard(slack)),
                     _capi.MIN_FN(guard(minimum)), _capi.SPLIT_FN(guard(split)))
        self.struct = _capi.PairSolvers(user=None, points=self._fns[0], slack=self._fns[1],
                                        min=self._fns[2], split=self._fns[3])


def default_horizons(mpc):
    """Horizons of the device tables of a law: the short horizon whose blocks fit the shared-block
    kernels with every column kept (``sequences.short_horizon``), then one table per longer horizon
    up to N -- the relaxation of a prefix of k steps runs at the size of the law with horizon k,
    not of the full model (include/ehm_frontier.h).  ``[N]`` where the split is not admissible."""
    from . import sequences
    k = sequences.short_horizon(mpc)
    if k < 1:
        return [mpc.N]
    return [k] + list(range(k + 1, mpc.N + 1))


class NativeFrontier:
    """One native driver handle: its device tables (or the caller's solvers), the searches'
    memory, the flat tree.  ``horizons``: ascending horizons of the tables (None =
    ``default_horizons``); ``slots``: blocks resident in the tables that do not hold every
    prefix of their lengths at once."""

    def __init__(self, mpc, eps_a, eps_r, slots=16384, device=0, horizons=None, solvers=None,
                 short_len=None):
        self.mpc = mpc
        self._lib = _capi.load()
        self._solvers = solvers
        h = ctypes.c_void_p()
        if solvers is not None:
            _capi.check_frontier(self._lib.ehm_frontier_create_custom(
                mpc.n_x, mpc.n_u, mpc.delta_size, mpc.N, ctypes.byref(solvers.struct),
                float(eps_a), float(eps_r), ctypes.byref(h)))
            self.horizons = []
        else:
            if horizons is None and short_len is not None:      # two tables: short_len and N
                horizons = [int(short_len), mpc.N] if 0 < int(short_len) < mpc.N else [mpc.N]
            self.horizons = [int(x) for x in (default_horizons(mpc) if horizons is None
                                              else horizons)]
            hz = np.array(self.horizons, dtype=np.int32)
            # a table whose prefixes all fit gets a slot each (0); the others share `slots`
            count, lo, sl = 0, 0, []
            for hh in self.horizons:
                count = sum(mpc.delta_size ** k for k in range(lo, hh + 1))
                sl.append(0 if count <= int(slots) else int(slots))
                lo = hh + 1
            sl = np.array(sl, dtype=np.int32)
            law, keep = _law_struct(mpc)
            _capi.check_frontier(self._lib.ehm_frontier_create(
                ctypes.byref(law), hz.size, ptr(hz), ptr(sl), int(device), float(eps_a),
                float(eps_r), ctypes.byref(h)))
            del keep
        self.short_len = self.horizons[0] if len(self.horizons) > 1 else 0
        self._h = h
        self.eps_a, self.eps_r = float(eps_a), float(eps_r)
        self.last_stats = None

    def close(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._lib.ehm_frontier_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc and self._solvers is not None and self._solvers.error is not None:
            err, self._solvers.error = self._solvers.error, None
            raise err
        _capi.check_frontier(rc)

    def set_eps(self, eps_a, eps_r):
        self.eps_a, self.eps_r = float(eps_a), float(eps_r)
        self._check(self._lib.ehm_frontier_set_eps(self._h, self.eps_a, self.eps_r))

    def reset(self):
        """Drops the tree and what the searches remember; the loaded blocks stay."""
        self._check(self._lib.ehm_frontier_reset(self._h))

    def add_roots(self, roots):
        for R in np.asarray(roots, dtype=np.float64).reshape(-1, self.mpc.n_x + 1, self.mpc.n_x):
            R = np.ascontiguousarray(R)
            self._check(self._lib.ehm_frontier_add_root(self._h, ptr(R)))

    def run(self, round_cap=4096, launch_target=65536, max_visits=0, min_regions=0, speculate=0,
            max_depth=0):
        opts = _capi.FrontierOpts(round_cap=int(round_cap), launch_target=int(launch_target),
                                  max_visits=int(max_visits or 0),
                                  min_regions=int(min_regions or 0), speculate=int(speculate),
                                  max_depth=int(max_depth or 0))
        st = _capi.FrontierStats()
        self._check(self._lib.ehm_frontier_run(self._h, ctypes.byref(opts), ctypes.byref(st)))
        self.last_stats = {k: getattr(st, k) for k, _ in st._fields_}
        return self.last_stats

    def p_theta(self, thetas):
        """``Oracle.P_theta`` for many parameters (``bnb_frontier.p_theta_many``): list of
        (u0, delta, J), (None, None, None) where no mode sequence is feasible."""
        th = np.ascontiguousarray(thetas, dtype=np.float64).reshape(-1, self.mpc.n_x)
        n = th.shape[0]
        J = np.empty(n)
        u0 = np.empty((n, self.mpc.n_u))
        seq = np.empty((n, self.mpc.N), dtype=np.int32)
        self._check(self._lib.ehm_frontier_p_theta(self._h, n, ptr(th), ptr(J), ptr(u0), ptr(seq)))
        return [(u0[k].copy(), self.mpc.sequence_to_delta(tuple(int(i) for i in seq[k])),
                 float(J[k])) if np.isfinite(J[k]) else (None, None, None) for k in range(n)]

    def table_stats(self):
        """ehm_stats of the device tables: list of dicts (horizon, slots, evicted, counters)."""
        out = []
        k = 0
        while True:
            h, hz, sl, ev = ctypes.c_void_p(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
            self._check(self._lib.ehm_frontier_table(self._h, k, ctypes.byref(h), ctypes.byref(hz),
                                                     ctypes.byref(sl), ctypes.byref(ev)))
            if not h:
                return out
            c = _capi.Counters()
            _capi.check(self._lib.ehm_stats(h, ctypes.byref(c)))
            out.append(dict(horizon=hz.value, slots=sl.value, evicted=ev.value,
                            lp_solves=c.lp_solves, ipm_iters=c.ipm_iters,
                            kernel_launches=c.kernel_launches, stalled=c.stalled,
                            batch_seconds=list(c.batch_seconds),
                            batch_launches=list(c.batch_launches)))
            k += 1

    def lp_counts(self):
        """Problems solved by (table, kind, prefix length): (n_tables, 5, N+1)."""
        out = np.zeros((max(len(self.horizons), 1), 5, self.mpc.N + 1), dtype=np.int64)
        self._check(self._lib.ehm_frontier_lp_counts(self._h, ptr(out)))
        return out

    def export(self):
        """The flat tree: dict of arrays (roots first, children after their parents)."""
        n, nr = ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.ehm_frontier_sizes(self._h, ctypes.byref(n), ctypes.byref(nr)))
        n, p, nv, nu, N = n.value, self.mpc.n_x, self.mpc.n_x + 1, self.mpc.n_u, self.mpc.N
        out = dict(n_nodes=n, n_roots=nr.value, vertices=np.empty((n, nv, p)),
                   left=np.empty(n, dtype=np.int32), right=np.empty(n, dtype=np.int32),
                   sequence=np.empty((n, N), dtype=np.int32), vertex_costs=np.empty((n, nv)),
                   vertex_inputs=np.empty((n, nv, nu)), flags=np.empty(n, dtype=np.uint8))
        self._check(self._lib.ehm_frontier_export(
            self._h, ptr(out['vertices']), ptr(out['left']), ptr(out['right']),
            ptr(out['sequence']), ptr(out['vertex_costs']), ptr(out['vertex_inputs']),
            ptr(out['flags'])))
        return out


def graft(flat, mpc, targets, depth_limited=None):
    """Writes the flat tree of ``NativeFrontier.export`` into ``targets`` (one ``Tree`` per root,
    grown in place).  Returns the list of (Tree node, flags) of the cells handed back open;
    ``depth_limited``: a list that receives the leaves a depth limit left unbisected."""
    import gc
    was_on = gc.isenabled()
    gc.disable()            # (millions of small objects, nothing cyclic among them)
    try:
        return _graft(flat, mpc, targets, depth_limited if depth_limited is not None else [])
    finally:
        if was_on:
            gc.enable()


def _graft(flat, mpc, targets, depth_limited):
    n = flat['n_nodes']
    nodes = [None] * n
    for r, t in enumerate(targets):
        nodes[r] = t
    left, right, flags, seq = flat['left'], flat['right'], flat['flags'], flat['sequence']
    V, C, U = flat['vertices'], flat['vertex_costs'], flat['vertex_inputs']
    # the 0/1 vector of a sequence (lib/mpc_library.py:160), once per distinct sequence
    has = (flags & FR_HAS_RECORD) != 0
    deltas = {}
    handed_back = []
    for k in range(n):
        node = nodes[k]
        if has[k]:
            key = seq[k].tobytes()
            d = deltas.get(key)
            if d is None:
                d = deltas[key] = mpc.sequence_to_delta(tuple(int(i) for i in seq[k]))
            node.data = NodeData(vertices=V[k].copy(), commutation=d.copy(),
                                 vertex_costs=C[k].copy(), vertex_inputs=U[k].copy())
        else:
            node.data = NodeData(vertices=V[k].copy())
        node.data.is_epsilon_suboptimal = bool(flags[k] & FR_CLOSED)
        if left[k] >= 0:
            node.grow(None, None)
            nodes[left[k]] = node.left
            nodes[right[k]] = node.right
        elif flags[k] & FR_OPEN:
            handed_back.append((node, int(flags[k])))
        elif flags[k] & FR_DEPTH:
            # an open leaf the run's depth limit left unbisected: NOT epsilon-suboptimal, and --
            # where the limit met it before a commutation was found -- without a record.  Marked,
            # so that consumers written against lib/tree.py can tell it from a closed region.
            node.data.depth_limited = True
            depth_limited.append(node)
    return handed_back


def grow_cells(native, branches, slow_oracle=None, round_cap=4096, launch_target=65536,
               max_visits=0, min_regions=0, speculate=0, slow_opts=None, deadline=None,
               slice_visits=100000, max_depth=0):
    """
    ``bnb_frontier.grow_frontier(oracle, branches, 'ecc')`` on the native driver: ``branches`` (a
    ``Tree`` or a list of them, data = the root simplices) are grown in place.  ``slow_oracle``: a
    callable returning the ``bnb.PrefixOracle`` that finishes the cells handed back open (created
    on first need -- most cells of configs[4] never need it).  ``deadline``: a
    ``time.perf_counter()`` value; the run is then made in slices of ``slice_visits`` cell visits
    and stops (truncated: pending cells stay open leaves) at the first slice that ends after it.
    Returns a dict of counts.
    """
    import time
    from . import bnb_frontier
    branches = list(branches) if isinstance(branches, (list, tuple)) else [branches]
    native.reset()
    native.add_roots([np.asarray(b.data.vertices, dtype=np.float64) for b in branches])
    if deadline is None:
        st = dict(native.run(round_cap=round_cap, launch_target=launch_target,
                             max_visits=max_visits, min_regions=min_regions, speculate=speculate,
                             max_depth=max_depth))
    else:
        budget = 0
        while True:
            budget += int(slice_visits)
            cap = min(budget, int(max_visits)) if max_visits else budget
            st = dict(native.run(round_cap=round_cap, launch_target=launch_target, max_visits=cap,
                                 min_regions=min_regions, speculate=speculate,
                                 max_depth=max_depth))
            if not st['truncated'] or time.perf_counter() >= deadline or \
                    (max_visits and st['visits'] >= max_visits) or \
                    (min_regions and st['regions'] >= min_regions):
                break
    limited = []
    back = graft(native.export(), native.mpc, branches, limited)
    # leaves the depth limit left open (flag FR_DEPTH): neither regions nor handed back -- reported,
    # and marked in the tree (NodeData.depth_limited); those without a commutation have no vertex
    # inputs either, so a tree that holds any must not be given to ExplicitMPC as it is
    st['depth_limited_leaves'] = len(limited)
    st['depth_limited_without_commutation'] = sum(1 for nd in limited
                                                  if not hasattr(nd.data, 'commutation'))
    st['slow_path_cells'] = len(back)
    st['slow_path_regions'] = st['slow_path_visits'] = 0
    if back and not st['truncated']:
        if slow_oracle is None:
            raise RuntimeError('%d cell(s) were handed back open and no slow-path oracle was '
                               'given' % len(back))
        orc = slow_oracle()
        for action, want in (('lcss', 0), ('ecc', FR_NEEDS_ECC)):
            part = [nd for nd, fl in back if (fl & FR_NEEDS_ECC) == want]
            if part:
                s2 = bnb_frontier.grow_frontier(orc, part, action, **(slow_opts or {}))
                st['slow_path_regions'] += int(s2.get('regions', 0))
                st['slow_path_visits'] += int(s2['host_visits'])
        st['regions'] += st['slow_path_regions']
    return st
