"""
ctypes binding of libehmpc.so (include/ehmpc.h).  Thin: argument marshalling and error
translation only.  There is no CPU fallback -- if the library is missing, or there is no
HIP device, the calls raise.
"""

import ctypes
import os
import numpy as np

from . import build as _build

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)

EHM_OK = 0
EHM_E_INVALID = -1
EHM_E_NO_DEVICE = -2
EHM_E_HIP = -3
EHM_E_CAPACITY = -4
EHM_E_INFEASIBLE = -5
EHM_E_NUMERIC = -6
EHM_MAX_P = 8           # include/ehmpc.h

# every symbol include/ehmpc.h declares
EXPORTED = [
    'ehm_problem_create', 'ehm_problem_destroy', 'ehm_problem_set_eps', 'ehm_sync',
    'ehm_stream', 'ehm_solve_ptd_batch', 'ehm_feas_ptd_batch', 'ehm_solve_pt_batch',
    'ehm_vr_batch', 'ehm_slack_batch', 'ehm_bar_e_batch', 'ehm_min_simplex_batch',
    'ehm_bar_d_batch', 'ehm_split_batch', 'ehm_volume_batch', 'ehm_partition_run',
    'ehm_tree_info_get', 'ehm_tree_export', 'ehm_tree_destroy', 'ehm_stats',
    'ehm_last_error', 'ehm_version', 'ehm_problem_set_solver', 'ehm_selftest',
    'ehm_problem_set_option', 'ehm_partition_begin', 'ehm_partition_step',
    'ehm_partition_take', 'ehm_partition_give', 'ehm_partition_finish',
    'ehm_explicit_create', 'ehm_explicit_eval_batch', 'ehm_explicit_destroy',
    'ehm_explicit_last_error', 'ehm_partition_progress', 'ehm_partition_counts', 'ehm_partition_advance', 'ehm_problem_set_quadratic',
    'ehm_feas_all_batch', 'ehm_lcss_batch', 'ehm_partition_movable',
    'ehm_problem_update_blocks', 'ehm_simplex_idx_batch', 'ehm_point_idx_batch',
    'ehm_abi_sizes', 'ehm_solver_phase_ticks', 'ehm_problem_layout',
    'ehm_host_alloc', 'ehm_host_free',
]


# every symbol include/ehm_search.h declares (host bookkeeping of the prefix searches)
EXPORTED_SEARCH = [
    'ehm_search_create', 'ehm_search_destroy', 'ehm_search_last_error', 'ehm_search_point_ids',
    'ehm_search_register_midpoints', 'ehm_search_forget', 'ehm_search_counts',
    'ehm_search_query', 'ehm_search_asks', 'ehm_search_answer', 'ehm_search_descent_begin',
    'ehm_search_descent_step', 'ehm_search_descent_result', 'ehm_search_peek',
    'ehm_search_abandon', 'ehm_search_peek_any',
    'ehm_search_bare_create', 'ehm_search_bare_destroy', 'ehm_search_bare_seed',
    'ehm_search_bare_bounds', 'ehm_search_bare_step', 'ehm_search_bare_asks',
    'ehm_search_bare_answer', 'ehm_search_bare_result', 'ehm_search_bare_learned',
]


# every symbol include/ehm_frontier.h declares (the native partition driver of configs[4])
EXPORTED_FRONTIER = [
    'ehm_frontier_create', 'ehm_frontier_create_custom', 'ehm_frontier_destroy',
    'ehm_frontier_last_error', 'ehm_frontier_set_eps', 'ehm_frontier_table',
    'ehm_frontier_reset', 'ehm_frontier_add_root', 'ehm_frontier_run', 'ehm_frontier_sizes',
    'ehm_frontier_export', 'ehm_frontier_lp_counts', 'ehm_frontier_condense',
    'ehm_frontier_p_theta', 'ehm_frontier_pending', 'ehm_frontier_take', 'ehm_frontier_give',
]


class EhmError(RuntimeError):
    def __init__(self, code, message):
        super().__init__('libehmpc error %d: %s' % (code, message))
        self.code = code


class ProblemDesc(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int32), ('m', ctypes.c_int32), ('p', ctypes.c_int32),
                ('n_u', ctypes.c_int32), ('n_delta', ctypes.c_int32),
                ('delta_len', ctypes.c_int32),
                ('G', c_double_p), ('w', c_double_p), ('S', c_double_p), ('c', c_double_p),
                ('deltas', c_uint8_p),
                ('eps_a', ctypes.c_double), ('eps_r', ctypes.c_double)]


class RunOpts(ctypes.Structure):
    _fields_ = [('max_nodes', ctypes.c_int64), ('max_depth', ctypes.c_int32),
                ('action', ctypes.c_int32), ('engine', ctypes.c_int32),
                ('shard_rank', ctypes.c_int32), ('shard_world', ctypes.c_int32),
                ('skip_volume', ctypes.c_int32), ('shard_min_frontier', ctypes.c_int64),
                ('deal_depth', ctypes.c_int32), ('reserved0', ctypes.c_int32)]


class NodeInit(ctypes.Structure):
    _fields_ = [('delta', c_uint8_p), ('vcost', c_double_p), ('vinput', c_double_p)]


class TreeInfo(ctypes.Structure):
    _fields_ = [('n_nodes', ctypes.c_int64), ('n_leaves', ctypes.c_int64),
                ('n_roots', ctypes.c_int64), ('n_closed', ctypes.c_int64),
                ('lp_solves', ctypes.c_int64), ('ref_solves', ctypes.c_int64),
                ('ipm_iters', ctypes.c_int64), ('sweeps', ctypes.c_int64),
                ('max_depth', ctypes.c_int32), ('truncated', ctypes.c_int32),
                ('volume_closed', ctypes.c_double), ('min_margin', ctypes.c_double),
                ('device_seconds', ctypes.c_double), ('decide_seconds', ctypes.c_double),
                ('expand_seconds', ctypes.c_double), ('decide_launches', ctypes.c_int64),
                ('expand_launches', ctypes.c_int64), ('decide_solves', ctypes.c_int64),
                ('decide_iters', ctypes.c_int64), ('replicated_closed', ctypes.c_int64),
                ('replicated_nodes', ctypes.c_int64), ('replicated_solves', ctypes.c_int64),
                ('cert_closed', ctypes.c_int64), ('witness_open', ctypes.c_int64),
                ('swaps', ctypes.c_int64), ('blacklisted', ctypes.c_int64),
                ('kind_solves', ctypes.c_int64 * 5), ('kind_iters', ctypes.c_int64 * 5),
                ('near_threshold', ctypes.c_int64), ('witness_inherited', ctypes.c_int64),
                ('midpoints_shared', ctypes.c_int64),
                ('persist_ticks', ctypes.c_int64 * 10), ('witness_table', ctypes.c_int64)]


class Progress(ctypes.Structure):
    _fields_ = [('n_nodes', ctypes.c_int64), ('n_closed', ctypes.c_int64),
                ('frontier', ctypes.c_int64), ('sweeps', ctypes.c_int64),
                ('lp_solves', ctypes.c_int64), ('ipm_iters', ctypes.c_int64),
                ('depth', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('volume_closed', ctypes.c_double), ('n_splits', ctypes.c_int64)]


class PwaLaw(ctypes.Structure):
    _fields_ = [('n_x', ctypes.c_int32), ('n_u', ctypes.c_int32), ('n_modes', ctypes.c_int32),
                ('N', ctypes.c_int32), ('A', c_double_p), ('B', c_double_p), ('w', c_double_p),
                ('region_rows', c_int32_p), ('Hx', c_double_p), ('hx', c_double_p),
                ('n_gx', ctypes.c_int32), ('Gx', c_double_p), ('gx', c_double_p),
                ('n_gu', ctypes.c_int32), ('Gu', c_double_p), ('gu', c_double_p),
                ('n_q', ctypes.c_int32), ('Q', c_double_p),
                ('n_r', ctypes.c_int32), ('R', c_double_p)]


POINTS_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                             ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                             ctypes.c_void_p)
SLACK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.c_void_p)
MIN_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
SPLIT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)


class PairSolvers(ctypes.Structure):
    _fields_ = [('user', ctypes.c_void_p), ('points', POINTS_FN), ('slack', SLACK_FN),
                ('min', MIN_FN), ('split', SPLIT_FN)]


class FrontierOpts(ctypes.Structure):
    _fields_ = [('round_cap', ctypes.c_int32), ('launch_target', ctypes.c_int32),
                ('max_visits', ctypes.c_int64), ('min_regions', ctypes.c_int64),
                ('speculate', ctypes.c_int32), ('max_depth', ctypes.c_int32)]


class FrontierStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int64) for k in (
        'rounds', 'visits', 'ecc_visits', 'lcss_visits', 'regions', 'open_cells', 'n_nodes',
        'calls_v_r', 'calls_p_theta', 'calls_bar_e', 'calls_bar_d', 'depth_limited', 'swaps',
        'witness_hits', 'prefixes_expanded',
        'answered_without_a_problem', 'optima_asked', 'optima_solved', 'lp_solves', 'launches',
        'blocks_loaded', 'stalled', 'slivers')] + [
        ('truncated', ctypes.c_int32), ('depth', ctypes.c_int32),
        ('seconds_solvers', ctypes.c_double), ('seconds_total', ctypes.c_double)]


class Counters(ctypes.Structure):
    _fields_ = [('lp_solves', ctypes.c_int64), ('ipm_iters', ctypes.c_int64),
                ('kernel_launches', ctypes.c_int64), ('stalled', ctypes.c_int64),
                ('fallbacks', ctypes.c_int64), ('slivers', ctypes.c_int64),
                ('batch_seconds', ctypes.c_double * 2), ('batch_launches', ctypes.c_int64 * 2)]


_lib = None


def library_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load libehmpc.so (building it first if the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('EHM_LIB') or _build.LIB      # EHM_LIB: an experimental build
    if build_if_missing and path == _build.LIB and _build.is_stale():
        _build.build()
    if not os.path.exists(path):
        raise EhmError(EHM_E_NO_DEVICE,
                       'libehmpc.so is not built (run explicit_hybrid_mpc_amd/build.py); '
                       'there is no CPU fallback')
    lib = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    i64 = ctypes.c_int64
    i32 = ctypes.c_int32
    # the struct mirrors below must be the library's structs: a mirror that is too small would be
    # overrun by the library's writes (ehm_abi_sizes, include/ehmpc.h)
    mirrors = (ProblemDesc, RunOpts, NodeInit, Progress, TreeInfo, Counters)
    sizes = (i64 * len(mirrors))()
    lib.ehm_abi_sizes.argtypes = [vp, i32]
    if lib.ehm_abi_sizes(ctypes.addressof(sizes), len(mirrors)) != len(mirrors) or any(
            int(sizes[k]) != ctypes.sizeof(m) for k, m in enumerate(mirrors)):
        raise EhmError(EHM_E_INVALID, 'libehmpc.so and its ctypes binding disagree about the public '
                       'structs: %s against %s (rebuild the library)' % (
                           [int(v) for v in sizes], [ctypes.sizeof(m) for m in mirrors]))
    lib.ehm_last_error.restype = ctypes.c_char_p
    lib.ehm_version.restype = ctypes.c_char_p
    lib.ehm_stream.restype = vp
    lib.ehm_stream.argtypes = [vp]
    lib.ehm_problem_create.argtypes = [ctypes.POINTER(ProblemDesc), i32, ctypes.POINTER(vp)]
    lib.ehm_problem_destroy.argtypes = [vp]
    lib.ehm_problem_set_eps.argtypes = [vp, ctypes.c_double, ctypes.c_double]
    lib.ehm_sync.argtypes = [vp]
    lib.ehm_problem_set_solver.argtypes = [vp, i32]
    lib.ehm_problem_set_quadratic.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.ehm_selftest.argtypes = [i32, vp, i32, vp]
    lib.ehm_problem_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_double]
    lib.ehm_solve_ptd_batch.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    lib.ehm_feas_ptd_batch.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.ehm_solve_pt_batch.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.ehm_vr_batch.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.ehm_slack_batch.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    lib.ehm_bar_e_batch.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.ehm_min_simplex_batch.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.ehm_bar_d_batch.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ehm_feas_all_batch.argtypes = [vp, i64, vp, vp]
    lib.ehm_lcss_batch.argtypes = [vp, i64] + [vp] * 13
    lib.ehm_split_batch.argtypes = [i32, i64, i32, vp, vp, vp, vp]
    lib.ehm_volume_batch.argtypes = [i32, i64, i32, vp, vp]
    lib.ehm_partition_run.argtypes = [vp, i64, vp, ctypes.POINTER(NodeInit),
                                      ctypes.POINTER(RunOpts), ctypes.POINTER(vp)]
    lib.ehm_partition_begin.argtypes = lib.ehm_partition_run.argtypes
    lib.ehm_partition_step.argtypes = [vp, i32, vp]
    lib.ehm_partition_take.argtypes = [vp, i64, vp, vp, vp]
    lib.ehm_partition_give.argtypes = [vp, i64, vp, vp, vp]
    lib.ehm_partition_finish.argtypes = [vp]
    lib.ehm_partition_progress.argtypes = [vp, ctypes.POINTER(Progress)]
    lib.ehm_partition_counts.argtypes = [vp, vp, vp]
    lib.ehm_partition_advance.argtypes = [vp, i64, vp]
    lib.ehm_partition_movable.argtypes = [vp, vp, vp]
    lib.ehm_problem_update_blocks.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.ehm_simplex_idx_batch.argtypes = [vp, i64, vp, vp, vp, i32, vp, vp, vp]
    lib.ehm_point_idx_batch.argtypes = [vp, i64, vp, vp, i32, vp, vp, vp]
    lib.ehm_explicit_create.argtypes = [i32, i64, i32, i32, i32, vp, vp, vp, vp,
                                        ctypes.POINTER(vp)]
    lib.ehm_explicit_eval_batch.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.ehm_explicit_destroy.argtypes = [vp]
    lib.ehm_explicit_last_error.restype = ctypes.c_char_p
    lib.ehm_tree_info_get.argtypes = [vp, ctypes.POINTER(TreeInfo)]
    lib.ehm_tree_export.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ehm_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.ehm_host_free.argtypes = [vp]
    lib.ehm_tree_destroy.argtypes = [vp]
    lib.ehm_stats.argtypes = [vp, ctypes.POINTER(Counters)]
    lib.ehm_solver_phase_ticks.argtypes = [vp, vp]
    lib.ehm_problem_layout.argtypes = [vp, vp]
    for name in EXPORTED:
        fn = getattr(lib, name)
        if name not in ('ehm_last_error', 'ehm_version', 'ehm_stream',
                        'ehm_explicit_last_error'):
            fn.restype = i32
    lib.ehm_search_last_error.restype = ctypes.c_char_p
    lib.ehm_search_create.argtypes = [i32, i32, i32, ctypes.POINTER(vp)]
    lib.ehm_search_destroy.argtypes = [vp]
    lib.ehm_search_point_ids.argtypes = [vp, i64, vp, vp]
    lib.ehm_search_register_midpoints.argtypes = [vp, i64, vp, vp, vp]
    lib.ehm_search_forget.argtypes = [vp]
    lib.ehm_search_abandon.argtypes = [vp]
    lib.ehm_search_counts.argtypes = [vp, vp]
    lib.ehm_search_query.argtypes = [vp, i64, vp, vp, vp, vp, ctypes.POINTER(i64),
                                     ctypes.POINTER(i64)]
    lib.ehm_search_asks.argtypes = [vp, vp, vp, vp]
    lib.ehm_search_peek.argtypes = [vp, i64, vp, vp, vp]
    lib.ehm_search_answer.argtypes = [vp, vp, vp]
    lib.ehm_search_descent_begin.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.ehm_search_descent_step.argtypes = [vp, vp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.ehm_search_descent_result.argtypes = [vp, vp, ctypes.POINTER(i64)]
    lib.ehm_search_peek_any.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.ehm_search_bare_create.argtypes = [i32, i32, i32, vp, ctypes.POINTER(vp)]
    lib.ehm_search_bare_destroy.argtypes = [vp]
    lib.ehm_search_bare_seed.argtypes = [vp, i32, ctypes.c_uint64, ctypes.c_double, i32]
    lib.ehm_search_bare_bounds.argtypes = [vp, i32, i64, vp, vp]
    lib.ehm_search_bare_step.argtypes = [vp, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.ehm_search_bare_asks.argtypes = [vp, vp, vp]
    lib.ehm_search_bare_answer.argtypes = [vp, vp, ctypes.POINTER(i64)]
    lib.ehm_search_bare_result.argtypes = [vp, vp, vp, vp]
    lib.ehm_search_bare_learned.argtypes = [vp, i32, ctypes.POINTER(i64), vp, vp]
    for name in EXPORTED_SEARCH:
        if name != 'ehm_search_last_error':
            getattr(lib, name).restype = i32
    # include/ehm_frontier.h
    lib.ehm_frontier_last_error.restype = ctypes.c_char_p
    lib.ehm_frontier_create.argtypes = [ctypes.POINTER(PwaLaw), i32, vp, vp, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_double, ctypes.POINTER(vp)]
    lib.ehm_frontier_create_custom.argtypes = [i32, i32, i32, i32, ctypes.POINTER(PairSolvers),
                                               ctypes.c_double, ctypes.c_double,
                                               ctypes.POINTER(vp)]
    lib.ehm_frontier_destroy.argtypes = [vp]
    lib.ehm_frontier_set_eps.argtypes = [vp, ctypes.c_double, ctypes.c_double]
    lib.ehm_frontier_table.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i32),
                                       ctypes.POINTER(i32), ctypes.POINTER(i64)]
    lib.ehm_frontier_reset.argtypes = [vp]
    lib.ehm_frontier_add_root.argtypes = [vp, vp]
    lib.ehm_frontier_run.argtypes = [vp, ctypes.POINTER(FrontierOpts),
                                     ctypes.POINTER(FrontierStats)]
    lib.ehm_frontier_sizes.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.ehm_frontier_export.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ehm_frontier_lp_counts.argtypes = [vp, vp]
    lib.ehm_frontier_p_theta.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.ehm_frontier_pending.argtypes = [vp, ctypes.POINTER(i64)]
    lib.ehm_frontier_take.argtypes = [vp, i64, ctypes.POINTER(i64), vp, vp, vp, vp, vp, vp]
    lib.ehm_frontier_give.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.ehm_frontier_condense.argtypes = [ctypes.POINTER(PwaLaw), i32, i32, vp, vp, vp, vp, vp]
    for name in EXPORTED_FRONTIER:
        if name != 'ehm_frontier_last_error':
            getattr(lib, name).restype = i32
    _lib = lib
    return lib


def check(rc):
    if rc != EHM_OK:
        raise EhmError(rc, load().ehm_last_error().decode('utf-8', 'replace'))


def check_search(rc):
    if rc != EHM_OK:
        raise EhmError(rc, load().ehm_search_last_error().decode('utf-8', 'replace'))


def check_frontier(rc):
    if rc != EHM_OK:
        raise EhmError(rc, load().ehm_frontier_last_error().decode('utf-8', 'replace'))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def ptr(a):
    return None if a is None else a.ctypes.data
