"""
Batched device oracles and the partition engine: numpy-in / numpy-out wrappers over the
C-ABI (include/ehmpc.h).  This is the layer ``oracle.Oracle`` (reference signatures) and
``partition`` sit on.

Reference counterparts: the oracle problems of lib/oracle.py:23-443 evaluated for whole
batches of parameters / simplices at a time, and the per-node work of
lib/worker.py:241-417 executed as frontier sweeps on the GPU.
"""

import ctypes
import weakref
import numpy as np

from . import _capi
from ._capi import f64, u8, ptr, check


# -- page-locked arrays for the flat export ------------------------------------------------------
# ehm_tree_export copies straight into the caller's arrays; into page-locked memory
# (ehm_host_alloc) those copies run at the speed of the host link.  Blocks are recycled by size: a
# FlatTree that is garbage-collected hands its blocks to the next export of the same shape
# (pinning half a gigabyte costs more than copying it).
_PINNED_FREE = {}           # nbytes -> [address, ...]
_PINNED_CACHE_LIMIT = 4 << 30
_pinned_cached = [0]


def _pinned_release(addr, nbytes):
    if _pinned_cached[0] + nbytes <= _PINNED_CACHE_LIMIT:
        _PINNED_FREE.setdefault(nbytes, []).append(addr)
        _pinned_cached[0] += nbytes
    else:
        try:
            _capi.load().ehm_host_free(ctypes.c_void_p(addr))
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64):
    """An uninitialised numpy array in page-locked host memory (freed / recycled when the last
    view of it is dropped)."""
    import weakref
    dtype = np.dtype(dtype)
    nbytes = max(int(np.prod(shape)) * dtype.itemsize, 1)
    free = _PINNED_FREE.get(nbytes)
    if free:
        addr = free.pop()
        _pinned_cached[0] -= nbytes
    else:
        out = ctypes.c_void_p()
        check(_capi.load().ehm_host_alloc(nbytes, ctypes.byref(out)))
        addr = out.value
    buf = (ctypes.c_char * nbytes).from_address(addr)
    weakref.finalize(buf, _pinned_release, addr, nbytes)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def _info_dict(info):
    """ehm_tree_info as a dict (array members as lists)."""
    out = {}
    for name, _ in _capi.TreeInfo._fields_:
        v = getattr(info, name)
        out[name] = list(v) if hasattr(v, '__len__') else v
    return out


class FlatTree:
    """
    Flat export of a grown partition (struct of arrays, node k):
    vertices (K,p+1,p), left/right (K,) child index or -1, delta_idx (K,),
    vertex_costs (K,p+1), vertex_inputs (K,p+1,n_u), flags (K,) bit0 = epsilon-suboptimal,
    bit1 = has commutation data, tstar (K,) slack of the node's last decision.
    Nodes 0..n_roots-1 are the roots in input order.
    """

    def __init__(self, vertices, left, right, delta_idx, vertex_costs, vertex_inputs, flags,
                 tstar, info, deltas):
        self.vertices = vertices
        self.left = left
        self.right = right
        self.delta_idx = delta_idx
        self.vertex_costs = vertex_costs
        self.vertex_inputs = vertex_inputs
        self.flags = flags
        self.tstar = tstar
        self.info = info
        self.deltas = deltas

    @property
    def n_nodes(self):
        return self.left.shape[0]

    def is_leaf(self, k):
        return self.left[k] < 0

    def locations(self, root_locations=None, received=None):
        """
        Location string of every node ('0' left / '1' right, lib/worker.py:254-258).
        received: {node id: location} for the nodes another rank handed over (flag bit5);
        nodes below a received root whose location is not known yet stay None.
        """
        n_roots = self.info['n_roots']
        loc = [None] * self.n_nodes
        for r in range(n_roots):
            loc[r] = '' if root_locations is None else root_locations[r]
        for k, name in (received or {}).items():
            loc[k] = name
        for k in range(self.n_nodes):         # children always have larger indices
            if self.left[k] >= 0 and loc[k] is not None:
                loc[self.left[k]] = loc[k] + '0'
                loc[self.right[k]] = loc[k] + '1'
        return loc


class GpuProblem:
    """Device-resident canonical MPC instance (one GPU, one HIP stream)."""

    def __init__(self, canonical, eps_a, eps_r, device=0):
        self.can = canonical
        self.device = int(device)
        self._lib = _capi.load()
        self._handle = ctypes.c_void_p()
        can = canonical
        self._keep = (f64(can.G), f64(can.w), f64(can.S), f64(can.c),
                      u8(can.deltas.astype(int)))
        G, w, S, c, deltas = self._keep
        desc = _capi.ProblemDesc(
            n=can.n, m=can.m, p=can.p, n_u=can.n_u, n_delta=can.n_delta,
            delta_len=deltas.shape[1],
            G=G.ctypes.data_as(_capi.c_double_p), w=w.ctypes.data_as(_capi.c_double_p),
            S=S.ctypes.data_as(_capi.c_double_p), c=c.ctypes.data_as(_capi.c_double_p),
            deltas=deltas.ctypes.data_as(_capi.c_uint8_p),
            eps_a=float(eps_a), eps_r=float(eps_r))
        check(self._lib.ehm_problem_create(ctypes.byref(desc), self.device,
                                           ctypes.byref(self._handle)))
        self.eps_a = float(eps_a)
        self.eps_r = float(eps_r)
        if getattr(can, 'quadratic', False):
            # quadratic cost (the reference's cvx.quad_form laws): convex QP / QCQP oracles
            q = tuple(f64(getattr(can, k)) for k in ('H', 'F', 'f0', 'C', 'c1', 'c0'))
            self._keep = self._keep + q
            check(self._lib.ehm_problem_set_quadratic(self._handle, *[ptr(a) for a in q]))

    def close(self):
        if getattr(self, '_handle', None) is not None and self._handle:
            # an unfinished run goes first: its tree swaps its node pool into this handle's cache
            # (the library also detaches every tree still alive, ehm_problem_destroy)
            run = getattr(self, '_live_run', None)
            run = run() if run is not None else None
            if run is not None:
                run.abort()
            self._lib.ehm_problem_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_eps(self, eps_a, eps_r):
        check(self._lib.ehm_problem_set_eps(self._handle, float(eps_a), float(eps_r)))
        self.eps_a, self.eps_r = float(eps_a), float(eps_r)

    def set_solver(self, generation):
        """Kernel generation: 2 = shared constant block, several wavefronts per workgroup
        (default); 1 = one wavefront per workgroup (kept as an on-device cross-check)."""
        check(self._lib.ehm_problem_set_solver(self._handle, int(generation)))

    def set_option(self, name, value):
        """Named options of the handle (include/ehmpc.h: "solver", "decide_full")."""
        check(self._lib.ehm_problem_set_option(self._handle, name.encode(), float(value)))

    # -- helpers ------------------------------------------------------------------------
    def _delta_arg(self, delta, n_inst):
        if delta is None:
            return None
        d = u8(np.asarray(delta).astype(int))
        if d.ndim == 1:
            d = np.ascontiguousarray(np.broadcast_to(d, (n_inst, d.shape[0])))
        return d

    # -- batched oracles ------------------------------------------------------------------
    def solve_ptd(self, theta, delta=None):
        """P_theta_delta for a batch: returns (J, u0, status, iters)."""
        theta = f64(np.atleast_2d(theta))
        n = theta.shape[0]
        d = self._delta_arg(delta, n)
        J = np.empty(n)
        u0 = np.empty((n, self.can.n_u))
        status = np.empty(n, dtype=np.int32)
        iters = np.empty(n, dtype=np.int32)
        check(self._lib.ehm_solve_ptd_batch(self._handle, n, ptr(theta), ptr(d), ptr(J),
                                            ptr(u0), ptr(status), ptr(iters)))
        return J, u0, status, iters

    def feasible_ptd(self, theta, delta=None):
        """check_feasibility form of P_theta_delta: returns (feasible bool array, tau)."""
        theta = f64(np.atleast_2d(theta))
        n = theta.shape[0]
        d = self._delta_arg(delta, n)
        feas = np.empty(n, dtype=np.uint8)
        tau = np.empty(n)
        check(self._lib.ehm_feas_ptd_batch(self._handle, n, ptr(theta), ptr(d), ptr(feas),
                                           ptr(tau)))
        return feas.astype(bool), tau

    def solve_pt(self, theta):
        """P_theta: returns (J, u0, delta_idx)."""
        theta = f64(np.atleast_2d(theta))
        n = theta.shape[0]
        J = np.empty(n)
        u0 = np.empty((n, self.can.n_u))
        didx = np.empty(n, dtype=np.int32)
        check(self._lib.ehm_solve_pt_batch(self._handle, n, ptr(theta), ptr(J), ptr(u0),
                                           ptr(didx)))
        return J, u0, didx

    def v_r(self, R):
        """V_R: returns (delta_idx, vertex J, vertex u0)."""
        R = f64(R).reshape(-1, self.can.p + 1, self.can.p)
        n = R.shape[0]
        didx = np.empty(n, dtype=np.int32)
        vJ = np.empty((n, self.can.p + 1))
        vu = np.empty((n, self.can.p + 1, self.can.n_u))
        check(self._lib.ehm_vr_batch(self._handle, n, ptr(R), ptr(didx), ptr(vJ), ptr(vu)))
        return didx, vJ, vu

    def slack(self, R, Vbar, delta=None):
        """t*(delta) of the suboptimality test: returns (tstar, alpha, status)."""
        R = f64(R).reshape(-1, self.can.p + 1, self.can.p)
        n = R.shape[0]
        Vbar = f64(Vbar).reshape(n, self.can.p + 1)
        d = self._delta_arg(delta, n)
        t = np.empty(n)
        alpha = np.empty((n, self.can.p + 1))
        status = np.empty(n, dtype=np.int32)
        check(self._lib.ehm_slack_batch(self._handle, n, ptr(R), ptr(Vbar), ptr(d), ptr(t),
                                        ptr(alpha), ptr(status)))
        return t, alpha, status

    def bar_e(self, R, Vbar):
        """bar_E_delta_R: returns (closed bool array, best slack)."""
        R = f64(R).reshape(-1, self.can.p + 1, self.can.p)
        n = R.shape[0]
        Vbar = f64(Vbar).reshape(n, self.can.p + 1)
        closed = np.empty(n, dtype=np.uint8)
        tb = np.empty(n)
        check(self._lib.ehm_bar_e_batch(self._handle, n, ptr(R), ptr(Vbar), ptr(closed),
                                        ptr(tb)))
        return closed.astype(bool), tb

    def min_simplex(self, R, delta=None):
        R = f64(R).reshape(-1, self.can.p + 1, self.can.p)
        n = R.shape[0]
        d = self._delta_arg(delta, n)
        J = np.empty(n)
        status = np.empty(n, dtype=np.int32)
        check(self._lib.ehm_min_simplex_batch(self._handle, n, ptr(R), ptr(d), ptr(J),
                                              ptr(status)))
        return J, status

    def bar_d(self, R, Vbar, delta_ref):
        """bar_D_delta_R: returns (delta_idx, theta_star, vJ, vu0, var_small)."""
        R = f64(R).reshape(-1, self.can.p + 1, self.can.p)
        n = R.shape[0]
        Vbar = f64(Vbar).reshape(n, self.can.p + 1)
        d = self._delta_arg(delta_ref, n)
        didx = np.empty(n, dtype=np.int32)
        ths = np.empty((n, self.can.p))
        vJ = np.empty((n, self.can.p + 1))
        vu = np.empty((n, self.can.p + 1, self.can.n_u))
        vs = np.empty(n, dtype=np.uint8)
        check(self._lib.ehm_bar_d_batch(self._handle, n, ptr(R), ptr(Vbar), ptr(d), ptr(didx),
                                        ptr(ths), ptr(vJ), ptr(vu), ptr(vs)))
        return didx, ths, vJ, vu, vs.astype(bool)

    def feas_all(self, theta):
        """Feasibility of every commutation at every parameter: (n, n_delta) bool."""
        theta = f64(np.atleast_2d(theta))
        n = theta.shape[0]
        out = np.empty((n, self.can.n_delta), dtype=np.uint8)
        check(self._lib.ehm_feas_all_batch(self._handle, n, ptr(theta), ptr(out)))
        return out.astype(bool)

    def lcss(self, R, Vbar, delta_ref, vfeas, cand=None):
        """
        One ``Worker.lcss`` visit per node (lib/worker.py:367-401): bar_E and, for the open
        nodes, bar_D, sharing slacks and feasibility knowledge (include/ehmpc.h,
        ehm_lcss_batch).  vfeas (n, p+1, n_delta) bool, cand (n, n_delta) bool or None.
        Returns (closed, tbest, cand_out, delta_idx, theta_star, vJ, vu0, var_small).
        """
        R = f64(R).reshape(-1, self.can.p + 1, self.can.p)
        n = R.shape[0]
        p, n_u, nd = self.can.p, self.can.n_u, self.can.n_delta
        Vbar = f64(Vbar).reshape(n, p + 1)
        d = self._delta_arg(delta_ref, n)
        vf = u8(np.asarray(vfeas).reshape(n, p + 1, nd))
        cd = None if cand is None else u8(np.asarray(cand).reshape(n, nd))
        closed = np.empty(n, dtype=np.uint8)
        tb = np.empty(n)
        cand_out = np.empty((n, nd), dtype=np.uint8)
        didx = np.empty(n, dtype=np.int32)
        ths = np.zeros((n, p))
        vJ = np.zeros((n, p + 1))
        vu = np.zeros((n, p + 1, n_u))
        vs = np.empty(n, dtype=np.uint8)
        check(self._lib.ehm_lcss_batch(self._handle, n, ptr(R), ptr(Vbar), ptr(d), ptr(vf),
                                       ptr(cd), ptr(closed), ptr(tb), ptr(cand_out), ptr(didx),
                                       ptr(ths), ptr(vJ), ptr(vu), ptr(vs)))
        return (closed.astype(bool), tb, cand_out.astype(bool), didx, ths, vJ, vu,
                vs.astype(bool))

    # -- the commutation table as a cache of generated problems -------------------------------
    def update_blocks(self, first, G, w, S):
        """Replace the blocks of slots first .. first+len(G)-1 (ehm_problem_update_blocks)."""
        G, w, S = f64(G), f64(w), f64(S)
        check(self._lib.ehm_problem_update_blocks(self._handle, int(first), int(G.shape[0]),
                                                  ptr(G), ptr(w), ptr(S)))

    def simplex_idx(self, R, slot, mode, Vbar=None):
        """Problems over simplices with the commutation as a slot index: (obj, alpha, status).
        mode 0 = min, 1 = suboptimality-test slack, 2 = phase one."""
        R = f64(R).reshape(-1, self.can.p + 1, self.can.p)
        n = R.shape[0]
        slot = np.ascontiguousarray(slot, dtype=np.int32).reshape(n)
        V = None if Vbar is None else f64(Vbar).reshape(n, self.can.p + 1)
        obj = np.empty(n)
        alpha = np.empty((n, self.can.p + 1))
        status = np.empty(n, dtype=np.int32)
        check(self._lib.ehm_simplex_idx_batch(self._handle, n, ptr(R), ptr(V), ptr(slot),
                                              int(mode), ptr(obj), ptr(alpha), ptr(status)))
        return obj, alpha, status

    def point_idx(self, theta, slot, feas=False):
        """P_theta_delta (or its phase-one form) with slot indices: (J or tau, u0, status)."""
        theta = f64(np.atleast_2d(theta))
        n = theta.shape[0]
        slot = np.ascontiguousarray(slot, dtype=np.int32).reshape(n)
        J = np.empty(n)
        u0 = np.empty((n, self.can.n_u))
        status = np.empty(n, dtype=np.int32)
        check(self._lib.ehm_point_idx_batch(self._handle, n, ptr(theta), ptr(slot),
                                            1 if feas else 0, ptr(J), ptr(u0), ptr(status)))
        return J, u0, status

    # -- partition ------------------------------------------------------------------------------
    def partition(self, roots, action='ecc', init=None, max_nodes=0, max_depth=0, engine=1,
                  export=True, shard=None, with_volume=True, status=None, status_sweeps=1,
                  deal_depth=0):
        """
        Grow every root simplex until all leaves are epsilon-suboptimal.
        engine: 1 (default) = one launch of the persistent frontier kernel where it applies (a
        single rank running to completion on the shared-block kernels; the export relabels the
        nodes to the breadth-first numbering), 0 = level-synchronous sweeps.  Sharded runs,
        runs advanced sweep by sweep (``status``, ``begin``/``step``), the one-wavefront and the
        wide kernels always sweep.
        roots: (n_roots, p+1, p).  init: optional dict(delta, vertex_costs, vertex_inputs)
        for action 'lcss'.  shard = (rank, world, min_frontier) keeps only this rank's share
        of the frontier once it is min_frontier wide (multi-GPU); with deal_depth > 0 the
        whole run is instead ONE persistent launch per rank from the roots, dealt at that tree
        depth by the nodes' path codes (include/ehmpc.h, ehm_run_opts.deal_depth).  Returns a
        FlatTree, or just the info dict when export is False.
        status: optional ``status.MainStatusPublisher``; the run is then advanced
        ``status_sweeps`` frontier sweeps at a time and the publisher is fed the device's
        progress counters in between (status.txt / statistics.pkl of the reference).
        """
        if status is not None:
            from . import status as _status
            run = self.begin(roots, action=action, init=init, max_nodes=max_nodes,
                             max_depth=max_depth, shard=shard, with_volume=with_volume)
            worker = _status.WorkerStatus(algorithm=action)
            worker.update(active=True)
            while True:
                live = run.step(max(1, int(status_sweeps)))
                worker.absorb(run.progress())
                status.update([worker.data], num_tasks_in_queue=live)
                if live == 0:
                    break
            worker.update(active=False)
            status.update([worker.data], num_tasks_in_queue=0, force=True)
            return run.finish(export=export)
        roots = f64(roots).reshape(-1, self.can.p + 1, self.can.p)
        n_roots = roots.shape[0]
        rank, world, min_frontier = shard if shard is not None else (0, 1, 0)
        opts = _capi.RunOpts(max_nodes=int(max_nodes), max_depth=int(max_depth),
                             action=0 if action == 'ecc' else 1, engine=int(engine),
                             shard_rank=int(rank), shard_world=int(world),
                             shard_min_frontier=int(min_frontier),
                             skip_volume=0 if with_volume else 1, deal_depth=int(deal_depth))
        init_struct = None
        keep = None
        if init is not None:
            dl = u8(np.asarray(init['delta']).astype(int)).reshape(n_roots, -1)
            vc = f64(init['vertex_costs']).reshape(n_roots, self.can.p + 1)
            vi = f64(init['vertex_inputs']).reshape(n_roots, self.can.p + 1, self.can.n_u)
            keep = (dl, vc, vi)
            init_struct = ctypes.pointer(_capi.NodeInit(
                delta=dl.ctypes.data_as(_capi.c_uint8_p),
                vcost=vc.ctypes.data_as(_capi.c_double_p),
                vinput=vi.ctypes.data_as(_capi.c_double_p)))
        tree = ctypes.c_void_p()
        check(self._lib.ehm_partition_run(self._handle, n_roots, ptr(roots), init_struct,
                                          ctypes.byref(opts), ctypes.byref(tree)))
        del keep
        try:
            info = _capi.TreeInfo()
            check(self._lib.ehm_tree_info_get(tree, ctypes.byref(info)))
            info_d = _info_dict(info)
            if not export:
                return info_d
            K = info.n_nodes
            p, n_u = self.can.p, self.can.n_u
            vertices = pinned_empty((K, p + 1, p))
            left = pinned_empty(K, np.int32)
            right = pinned_empty(K, np.int32)
            didx = pinned_empty(K, np.int32)
            vcost = pinned_empty((K, p + 1))
            vinput = pinned_empty((K, p + 1, n_u))
            flags = pinned_empty(K, np.uint8)
            tstar = pinned_empty(K)
            check(self._lib.ehm_tree_export(tree, ptr(vertices), ptr(left), ptr(right),
                                            ptr(didx), ptr(vcost), ptr(vinput), ptr(flags),
                                            ptr(tstar)))
        finally:
            self._lib.ehm_tree_destroy(tree)
        return FlatTree(vertices, left, right, didx, vcost, vinput, flags, tstar, info_d,
                        self.can.deltas)

    def begin(self, roots, action='ecc', init=None, max_nodes=0, max_depth=0, shard=None,
              with_volume=True):
        """Resumable partition run (see PartitionRun); same arguments as partition()."""
        return PartitionRun(self, roots, action, init, max_nodes, max_depth, shard, with_volume)

    def layout(self):
        """ehm_problem_layout: dict(nd0, nE, LE, lda) -- what the solver eliminates (reporting)."""
        out = (ctypes.c_int32 * 4)()
        check(self._lib.ehm_problem_layout(self._handle, ctypes.addressof(out)))
        return dict(nd0=int(out[0]), nE=int(out[1]), LE=int(out[2]), lda=int(out[3]))

    def stats(self):
        c = _capi.Counters()
        check(self._lib.ehm_stats(self._handle, ctypes.byref(c)))
        return {name: (list(getattr(c, name)) if hasattr(getattr(c, name), '__len__')
                       else getattr(c, name)) for name, _ in _capi.Counters._fields_}


class PartitionRun:
    """
    The partition run in resumable pieces (ehm_partition_begin/step/take/give/finish): what the
    multi-GPU driver in ``distributed.py`` needs to move frontier nodes between ranks.
    """

    def __init__(self, gp, roots, action, init, max_nodes, max_depth, shard, with_volume):
        self.gp = gp
        self._lib = gp._lib
        can = gp.can
        roots = f64(roots).reshape(-1, can.p + 1, can.p)
        n_roots = roots.shape[0]
        rank, world, min_frontier = shard if shard is not None else (0, 1, 0)
        opts = _capi.RunOpts(max_nodes=int(max_nodes), max_depth=int(max_depth),
                             action=0 if action == 'ecc' else 1, engine=0,
                             shard_rank=int(rank), shard_world=int(world),
                             shard_min_frontier=int(min_frontier),
                             skip_volume=0 if with_volume else 1)
        init_struct, keep = None, None
        if init is not None:
            dl = u8(np.asarray(init['delta']).astype(int)).reshape(n_roots, -1)
            vc = f64(init['vertex_costs']).reshape(n_roots, can.p + 1)
            vi = f64(init['vertex_inputs']).reshape(n_roots, can.p + 1, can.n_u)
            keep = (dl, vc, vi)
            init_struct = ctypes.pointer(_capi.NodeInit(
                delta=dl.ctypes.data_as(_capi.c_uint8_p),
                vcost=vc.ctypes.data_as(_capi.c_double_p),
                vinput=vi.ctypes.data_as(_capi.c_double_p)))
        self._tree = ctypes.c_void_p()
        check(self._lib.ehm_partition_begin(gp._handle, n_roots, ptr(roots), init_struct,
                                            ctypes.byref(opts), ctypes.byref(self._tree)))
        del keep
        w = ctypes.c_int32(0)
        check(self._lib.ehm_partition_movable(self._tree, None, ctypes.addressof(w)))
        self.nrec = int(w.value)       # doubles per travelling record (hybrid: + bit rows)
        self.frontier = n_roots
        gp._live_run = weakref.ref(self)

    def abort(self):
        """Release the device tree of a run that will not be finished (the handle is free again)."""
        if getattr(self, '_tree', None):
            self._lib.ehm_tree_destroy(self._tree)
            self._tree = ctypes.c_void_p()

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass

    def _checked(self, rc):
        """A failed engine call ends the run: one run at a time per handle, so let go of it."""
        if rc != _capi.EHM_OK:
            msg = _capi.load().ehm_last_error().decode('utf-8', 'replace')
            self.abort()
            raise _capi.EhmError(rc, msg)

    def step(self, max_sweeps=0):
        """Up to max_sweeps frontier sweeps (0 = until done); returns the live frontier size."""
        n = ctypes.c_int64(0)
        self._checked(self._lib.ehm_partition_step(self._tree, int(max_sweeps),
                                                   ctypes.addressof(n)))
        self.frontier = int(n.value)
        return self.frontier

    def advance(self, max_pops=0):
        """
        Up to max_pops node visits of the persistent frontier kernel (0 = to completion); what
        is left of its queue is the live frontier again (ehm_partition_advance).
        """
        n = ctypes.c_int64(0)
        self._checked(self._lib.ehm_partition_advance(self._tree, int(max_pops),
                                                      ctypes.addressof(n)))
        self.frontier = int(n.value)
        return self.frontier

    def progress(self):
        """Volume filled / simplex count / LP solves so far (ehm_partition_progress)."""
        pr = _capi.Progress()
        check(self._lib.ehm_partition_progress(self._tree, ctypes.byref(pr)))
        return {name: getattr(pr, name) for name, _ in _capi.Progress._fields_
                if name != 'reserved'}

    def movable(self):
        """Frontier nodes ``take`` can hand over right now (hybrid runs: lcss nodes only)."""
        n = ctypes.c_int64(0)
        check(self._lib.ehm_partition_movable(self._tree, ctypes.addressof(n), None))
        return int(n.value)

    def free_nodes(self):
        """Node records this run may still allocate (ehm_partition_counts)."""
        n, cap = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.ehm_partition_counts(self._tree, ctypes.addressof(n),
                                             ctypes.addressof(cap)))
        return int(cap.value - n.value)

    def take(self, count):
        """Hand over the last `count` frontier nodes: (node ids, records, meta)."""
        count = int(count)
        ids = np.empty(count, dtype=np.int32)
        rec = np.empty((count, self.nrec))
        meta = np.empty((count, 2), dtype=np.int32)
        check(self._lib.ehm_partition_take(self._tree, count, ptr(ids), ptr(rec), ptr(meta)))
        self.frontier -= count
        return ids, rec, meta

    def take_device(self, count, device):
        """``take`` into torch tensors on ``device`` (the block stays in device memory: the
        multi-GPU driver sends it peer to peer).  Returns (node ids [numpy], records, meta)."""
        import torch
        count = int(count)
        ids = np.empty(count, dtype=np.int32)
        rec = torch.empty((count, self.nrec), dtype=torch.float64, device=device)
        meta = torch.empty((count, 2), dtype=torch.int32, device=device)
        check(self._lib.ehm_partition_take(self._tree, count, ptr(ids),
                                           ctypes.c_void_p(rec.data_ptr()),
                                           ctypes.c_void_p(meta.data_ptr())))
        self.frontier -= count
        return ids, rec, meta

    def give_device(self, records, meta):
        """``give`` from torch tensors in device memory (float64 (n, nrec), int32 (n, 2))."""
        import torch
        assert records.dtype == torch.float64 and meta.dtype == torch.int32
        records, meta = records.contiguous(), meta.contiguous()
        if records.is_cuda:
            torch.cuda.current_stream(records.device).synchronize()    # e.g. a pending recv
        first = ctypes.c_int32(0)
        check(self._lib.ehm_partition_give(self._tree, records.shape[0],
                                           ctypes.c_void_p(records.data_ptr()),
                                           ctypes.c_void_p(meta.data_ptr()),
                                           ctypes.addressof(first)))
        self.frontier += records.shape[0]
        return int(first.value)

    def give(self, records, meta):
        """Adopt nodes another rank took from its frontier; returns the id of the first one."""
        rec = f64(records).reshape(-1, self.nrec)
        meta = np.ascontiguousarray(meta, dtype=np.int32).reshape(-1, 2)
        first = ctypes.c_int32(0)
        check(self._lib.ehm_partition_give(self._tree, rec.shape[0], ptr(rec), ptr(meta),
                                           ctypes.addressof(first)))
        self.frontier += rec.shape[0]
        return int(first.value)

    def finish(self, export=True):
        """Totals (and the flat export); releases the device tree."""
        gp = self.gp
        try:
            check(self._lib.ehm_partition_finish(self._tree))
            info = _capi.TreeInfo()
            check(self._lib.ehm_tree_info_get(self._tree, ctypes.byref(info)))
            info_d = _info_dict(info)
            if not export:
                return info_d
            K = info.n_nodes
            p, n_u = gp.can.p, gp.can.n_u
            vertices = np.empty((K, p + 1, p))
            left = np.empty(K, dtype=np.int32)
            right = np.empty(K, dtype=np.int32)
            didx = np.empty(K, dtype=np.int32)
            vcost = np.empty((K, p + 1))
            vinput = np.empty((K, p + 1, n_u))
            flags = np.empty(K, dtype=np.uint8)
            tstar = np.empty(K)
            check(self._lib.ehm_tree_export(self._tree, ptr(vertices), ptr(left), ptr(right),
                                            ptr(didx), ptr(vcost), ptr(vinput), ptr(flags),
                                            ptr(tstar)))
        finally:
            self._lib.ehm_tree_destroy(self._tree)
            self._tree = ctypes.c_void_p()
        return FlatTree(vertices, left, right, didx, vcost, vinput, flags, tstar, info_d,
                        gp.can.deltas)


def selftest(device=0, max_instances=64):
    """Wave-primitive self test of every compiled kernel instance: (n, 5) array, each row
    expected to be (1072, 99, 25, 1/3, -1)."""
    out = np.zeros((max_instances, 5))
    n = ctypes.c_int32(0)
    lib = _capi.load()
    check(lib.ehm_selftest(int(device), ptr(out), int(max_instances), ctypes.addressof(n)))
    return out[:n.value]


# -- geometry (no problem handle needed) -----------------------------------------------------
def split_batch(R, device=0):
    """tools.split_along_longest_edge for a batch: returns (S1, S2, ij)."""
    R = f64(R)
    p = R.shape[-1]
    R = R.reshape(-1, p + 1, p)
    n = R.shape[0]
    S1 = np.empty_like(R)
    S2 = np.empty_like(R)
    ij = np.empty((n, 2), dtype=np.int32)
    lib = _capi.load()
    check(lib.ehm_split_batch(int(device), n, p, ptr(R), ptr(S1), ptr(S2), ptr(ij)))
    return S1, S2, ij


def volume_batch(R, device=0):
    R = f64(R)
    p = R.shape[-1]
    R = R.reshape(-1, p + 1, p)
    vol = np.empty(R.shape[0])
    lib = _capi.load()
    check(lib.ehm_volume_batch(int(device), R.shape[0], p, ptr(R), ptr(vol)))
    return vol
