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

FR_CLOSED, FR_HAS_RECORD, FR_OPEN, FR_PENDING, FR_NEEDS_ECC, FR_DEPTH, FR_REMOTE = 1, 2, 4, 8, 16, 32, 64


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

    def pending(self):
        """Cells in the handle's work lists (a truncated run's, or the roots before the first)."""
        n = ctypes.c_int64()
        self._check(self._lib.ehm_frontier_pending(self._h, ctypes.byref(n)))
        return n.value

    def take(self, max_cells):
        """
        ``ehm_frontier_take``: up to ``max_cells`` pending cells leave this handle, shallowest
        first; here they stay leaves flagged ``FR_REMOTE``.  Returns their records as a dict of
        arrays (``node`` = index in this handle's tree) -- what ``give`` of another handle takes,
        picklable, so it can travel between ranks (distributed.CellExchange).
        """
        m = int(min(max_cells, self.pending()))
        p, nv, nu, N = self.mpc.n_x, self.mpc.n_x + 1, self.mpc.n_u, self.mpc.N
        out = dict(node=np.empty(m, dtype=np.int32), vertices=np.empty((m, nv, p)),
                   sequence=np.empty((m, N), dtype=np.int32), vertex_costs=np.empty((m, nv)),
                   vertex_inputs=np.empty((m, nv, nu)), depth=np.empty(m, dtype=np.int32))
        n = ctypes.c_int64()
        self._check(self._lib.ehm_frontier_take(
            self._h, m, ctypes.byref(n), ptr(out['node']), ptr(out['vertices']),
            ptr(out['sequence']), ptr(out['vertex_costs']), ptr(out['vertex_inputs']),
            ptr(out['depth'])))
        assert n.value == m
        return out

    def give(self, cells):
        """``ehm_frontier_give``: the cells of another handle's ``take`` become roots of this
        handle's forest (only before it has grown: after ``reset``)."""
        m = int(len(cells['node']))
        V = np.ascontiguousarray(cells['vertices'], dtype=np.float64)
        S = np.ascontiguousarray(cells['sequence'], dtype=np.int32)
        C = np.ascontiguousarray(cells['vertex_costs'], dtype=np.float64)
        U = np.ascontiguousarray(cells['vertex_inputs'], dtype=np.float64)
        D = np.ascontiguousarray(cells['depth'], dtype=np.int32)
        self._check(self._lib.ehm_frontier_give(self._h, m, ptr(V), ptr(S), ptr(C), ptr(U), ptr(D)))

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


def merge_taken(giver, cells, taker):
    """
    One flat tree out of two: ``giver`` = export of the handle that gave ``cells`` away (they are
    its ``FR_REMOTE`` leaves ``cells['node']``), ``taker`` = export of the handle that grew them
    (its roots, in the same order).  The taker's sub-trees are appended behind the giver's nodes
    (children still follow their parents) and each taken leaf gets its root's final record and
    children.  The result is what ``graft`` and ``NativeFrontier.export`` consumers expect.
    """
    m = len(cells['node'])
    if taker['n_roots'] != m:
        raise ValueError('the taker holds %d roots for %d taken cells' % (taker['n_roots'], m))
    ng, nt = giver['n_nodes'], taker['n_nodes']
    where = np.empty(nt, dtype=np.int64)
    where[:m] = cells['node']
    where[m:] = ng + np.arange(nt - m)
    out = dict(n_nodes=ng + nt - m, n_roots=giver['n_roots'])
    for key in ('vertices', 'sequence', 'vertex_costs', 'vertex_inputs', 'flags', 'left', 'right'):
        out[key] = np.concatenate([giver[key], taker[key][m:]])
    for side in ('left', 'right'):
        kids = taker[side].astype(np.int64)
        mapped = np.where(kids >= 0, where[np.maximum(kids, 0)], -1).astype(np.int32)
        out[side][ng:] = mapped[m:]
        out[side][cells['node']] = mapped[:m]
    for k, g in enumerate(cells['node']):
        if not giver['flags'][g] & FR_REMOTE:
            raise ValueError('node %d of the giver is not a taken leaf' % g)
        if not np.array_equal(giver['vertices'][g], taker['vertices'][k]):
            raise ValueError('taken cell %d: the two handles disagree on its vertices' % k)
    for key in ('sequence', 'vertex_costs', 'vertex_inputs', 'flags'):
        out[key][cells['node']] = taker[key][:m]
    return out


def graft(flat, mpc, targets, depth_limited=None, remote=None):
    """Writes the flat tree of ``NativeFrontier.export`` into ``targets`` (one ``Tree`` per root,
    grown in place).  Returns the list of (Tree node, flags) of the cells handed back open;
    ``depth_limited``: a list that receives the leaves a depth limit left unbisected;
    ``remote``: a dict that receives {node index: Tree node} of the leaves ``take`` gave away
    (another handle grows them; ``NodeData.remote`` marks them until their sub-tree is attached)."""
    import gc
    was_on = gc.isenabled()
    gc.disable()            # (millions of small objects, nothing cyclic among them)
    try:
        return _graft(flat, mpc, targets, depth_limited if depth_limited is not None else [],
                      remote if remote is not None else {})
    finally:
        if was_on:
            gc.enable()


def attach(leaf, grown):
    """The sub-tree another handle grew from a cell ``take`` gave away (``grown``: its root) takes
    the place of the ``remote`` leaf it was taken as."""
    if not np.array_equal(leaf.data.vertices, grown.data.vertices):
        raise ValueError('the sub-tree does not belong to this leaf')
    leaf.data = grown.data              # (the taker's final record: it may have swapped in place)
    if not grown.is_leaf():
        leaf.left, leaf.right = grown.left, grown.right
    return leaf


def attach_adopted(given, adopted_by_id, never_grown=None):
    """Attaches the sub-trees other handles grew to the leaves they were taken as: ``given`` =
    [(parcel id, [leaf per cell])] of the runs that gave cells away, ``adopted_by_id`` = {parcel
    id: dict(trees, given)} of everybody who took some (a taker may have given cells of its
    parcel away in turn: those are attached first).  Returns the number of sub-trees attached.
    ``never_grown``: a list that receives the leaves of parcels nobody adopted (a run that a time
    limit ended while a parcel was waiting; they stay open leaves marked ``remote``) -- without
    it such a parcel is an error."""
    n = 0
    for pid, leaves in given:
        entry = adopted_by_id.get(pid)
        if entry is None:
            if never_grown is None:
                raise KeyError('parcel %r was given away and never grown' % (pid,))
            never_grown += leaves
            continue
        n += attach_adopted(entry['given'], adopted_by_id, never_grown)
        for leaf, grown in zip(leaves, entry['trees']):
            attach(leaf, grown)
            n += 1
    return n


class LocalExchange:
    """
    Take / give between the driver handles of ONE process (one interpreter thread each; the native
    calls release the GIL): a worker that finds no root left waits here, a worker in the middle
    of a run answers between two slices (``grow_cells(between_slices=exchange.serve)``) with the
    shallowest half of its pending cells.  The rank-to-rank form of the same protocol is
    ``distributed.CellExchange``.
    """

    def __init__(self, n_workers, poll=0.001):
        import collections
        import threading
        self.n = int(n_workers)
        self.poll = float(poll)
        self.lock = threading.Lock()
        self.hungry = 0         # waiting workers nobody has taken cells for yet
        self.idle = 0           # workers inside wait_for_work
        self.parcels = collections.deque()
        self.given = 0

    def serve(self, native, stats=None):
        """Every waiting worker gets an equal share: with k of them, k / (k + 1) of the pending
        cells leave, dealt in stripes of the shallowest-first order (a stripe each)."""
        pending = native.pending()
        with self.lock:
            k = min(self.hungry, pending - 1)
            if k <= 0:
                return None
            self.hungry -= k
            first = self.given
            self.given += k
        taken = native.take(max(k, pending * k // (k + 1)))
        parcels = []
        for i in range(k):
            part = {key: np.ascontiguousarray(v[i::k]) for key, v in taken.items()}
            part['id'] = first + 1 + i
            parcels.append(part)
        with self.lock:
            self.parcels.extend(parcels)
        return parcels

    def wait_for_work(self, stop=None):
        """A parcel, or None once every worker is idle (or ``stop()`` says the run is over)."""
        import time
        with self.lock:
            self.hungry += 1
            self.idle += 1
        while True:
            with self.lock:
                if self.parcels:
                    self.idle -= 1
                    return self.parcels.popleft()
                # (a parcel being taken for somebody keeps its giver out of `idle`)
                if self.idle >= self.n or (stop is not None and stop()):
                    self.hungry = max(0, self.hungry - 1)
                    return None
            time.sleep(self.poll)


def _graft(flat, mpc, targets, depth_limited, remote):
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
        elif flags[k] & FR_REMOTE:
            node.data.remote = True
            remote[k] = node
    return handed_back


def grow_cells(native, branches, slow_oracle=None, round_cap=4096, launch_target=65536,
               max_visits=0, min_regions=0, speculate=0, slow_opts=None, deadline=None,
               slice_visits=100000, max_depth=0, cells=None, between_slices=None):
    """
    ``bnb_frontier.grow_frontier(oracle, branches, 'ecc')`` on the native driver: ``branches`` (a
    ``Tree`` or a list of them, data = the root simplices) are grown in place.  ``slow_oracle``: a
    callable returning the ``bnb.PrefixOracle`` that finishes the cells handed back open (created
    on first need -- most cells of configs[4] never need it).  ``deadline``: a
    ``time.perf_counter()`` value; the run is then made in slices of ``slice_visits`` cell visits
    and stops (truncated: pending cells stay open leaves) at the first slice that ends after it.

    Take / give (include/ehm_frontier.h): ``cells`` = a parcel another handle's ``take`` produced
    -- the handle is given those instead of bare roots and ``branches`` are the trees they grow
    into (one per cell).  ``between_slices(native, stats)``: called after every slice that left
    work pending (the run is sliced by ``slice_visits`` then, deadline or not); it may ``take``
    cells away and returns the parcels it took (or None).  The leaves given away are reported as
    ``given_away`` = [(parcel, [Tree leaf per cell])], for ``attach`` when their sub-trees return.
    Returns a dict of counts.
    """
    import time
    from . import bnb_frontier
    branches = list(branches) if isinstance(branches, (list, tuple)) else [branches]
    native.reset()
    if cells is not None:
        if len(cells['node']) != len(branches):
            raise ValueError('one tree per given cell')
        native.give(cells)
    else:
        native.add_roots([np.asarray(b.data.vertices, dtype=np.float64) for b in branches])
    parcels = []
    if deadline is None and between_slices is None:
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
            if not st['truncated'] or (deadline is not None and time.perf_counter() >= deadline) \
                    or (max_visits and st['visits'] >= max_visits) or \
                    (min_regions and st['regions'] >= min_regions):
                break
            if between_slices is not None:
                parcels += [pc for pc in (between_slices(native, st) or []) if len(pc['node'])]
    limited, remote = [], {}
    back = graft(native.export(), native.mpc, branches, limited, remote)
    st['given_away'] = [(pc, [remote[int(k)] for k in pc['node']]) for pc in parcels]
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
