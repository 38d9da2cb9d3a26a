#!/usr/bin/env python
"""
Benchmark of the partitioning hot path on MI355X.

A "step" is one complete partition of the synthetic config-2 instance
(n_x=4, n_u=2, N=5, p=4 linear MPC, infinity-norm LP cost, eps_r=1e-2; BASELINE.json
configs[1]): feasible-commutation pass over the 22 Delaunay root simplices, then
epsilon-suboptimal refinement until every leaf is closed.  Problem constants and the
root simplices are resident in HBM before the timed region; the timed region contains
everything else (the persistent frontier kernel: all suboptimality tests, midpoint solves, child
construction).

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line on rank 0.  With N = 1 and the default workload the line also carries
"secondary": a few steps each of the other BASELINE.json configurations that fit one GPU
(config3, config4, config2q, config5), every entry with its own roofline and cpu_baseline.

For N > 1 it is launched by torch.distributed.run with one rank per GPU; every rank runs ONE
launch of the persistent frontier kernel from the roots: the tree above a deal depth is grown
identically everywhere, a node created at that depth is pursued only by the rank a hash of its
path names -- no collective in the data path (`--balance static`, the default; measured by running
the shards one after the other on one GPU: 1.68x / 2.55x / 3.50x at 2 / 4 / 8 on the 1.6 M-node
bench tree with the round-3 kernel; `--balance dynamic`: budgeted rounds of
the persistent kernel, an all-gather of the frontier sizes and point-to-point node transfers
between them, explicit_hybrid_mpc_amd/distributed.py).  The total work is fixed ("strong"
scaling), `value` is the whole-job LP-solve rate, and rank 0 checks after the timed region that
the regions of all ranks together are the regions of one unsharded partition
(`config.tree_identity`).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIG3 = (0.1, 1e-2, 18)    # abs_frac, eps_r, max_depth of --workload config3 (eps as config 2)
FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = FP32 vector / 2 (157.3 TF, MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def flops_per_iteration(n, m):
    """dense primal-dual IPM iteration on  min c^T z, G z <= h."""
    return 2. * m * n * n + n ** 3 / 3. + 8. * m * n + 4. * n * n


def node_bytes(p, n_u, delta_len):
    """node payload + 16 B topology/flags."""
    return 8 * ((p + 1) * p + (p + 1) + (p + 1) * n_u) + delta_len + 16


def flops_executed_per_iteration(n_lp, n_mpc, m, nE=0, LE=0):
    """
    What the shared-block solver (csrc/ehm_ipm2.h, round 4) really executes per iteration, as
    opposed to the SURVEY formula above.  nE z-columns are eliminated from the Newton system
    (ehm_problem_layout), nr = n_lp - nE are factorised at the compiled capacity NP:
      * eliminated block G, Delta: (16 or 32) x nE tasks x LE rows x 2 FMAs;
      * normal matrix of the MPC rows: ONE 16 x 16 matrix-core tile when the factorised columns
        with MPC entries are <= 16 (256 FMAs per row, both triangles and the padding included,
        + the Schur update as ceil(nE / 4) more K-steps), else vector FMAs on the lower-triangular
        4 x 4 block pairs (nb (nb + 1) / 2 pairs x 16 entries per row, the Schur rows included);
      * elimination unsymmetric at the capacity NP, both triangles: 2 NP^3 / 3;
      * four matrix-vector products on the factorised columns + one gathered entry per row for the
        eliminated ones; two pairs of triangular solves + the eliminated block's products.
    """
    nr = n_lp - nE
    nm = n_mpc - nE
    np_cap = np_capacity(nr)
    f = 0.
    if nE:
        f += 2. * 2. * (16 if np_cap <= 16 else 32) * nE * LE
    if nm <= 16:
        f += 2. * 256. * (m + 4. * ((nE + 3) // 4))
    else:
        nb = (nm + 3) // 4
        f += nb * (nb + 1) / 2. * 32. * (m + nE)
    f += 2. * np_cap ** 3 / 3.
    f += 8. * m * (nr + (1 if nE else 0))
    f += 4. * np_cap * np_cap + 8. * nE * nr
    return f


def np_capacity(n_lp):
    """Column capacity of the compiled instance that holds n_lp columns (build.K2_NPS)."""
    return next(c for c in (8, 12, 16, 20, 24, 28, 32, 64) if c >= n_lp)


def kernel_source_hash():
    """sha256 over the KERNEL sources: a profile is only quoted for the code it was taken from.
    Every .hip / .h under csrc/ except ehm_capi.hip -- the C-ABI and its host orchestration; the
    kernels bench.py prices live in ehm_kp.hip / ehm_k2.hip / ehm_k3.hip and the headers they
    include, and a host-side edit must not make their committed profiles look stale."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'explicit_hybrid_mpc_amd', 'csrc')
    for name in sorted(os.listdir(src)):
        if name.endswith(('.hip', '.h')) and name != 'ehm_capi.hip':
            h.update(open(os.path.join(src, name), 'rb').read())
    return h.hexdigest()[:16]


def wide_kernel(what):
    """Name of the wide-LP kernel the library launches: the LDS-resident family (ehm_k4.hip) unless
    EHM_K4=0 selects the streaming one (ehm_k3.hip)."""
    return ('k3_' if os.environ.get('EHM_K4') == '0' else 'k4_') + what


def pmc_traffic(kernel, summary):
    """
    HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of THIS workload
    (profiles/<round>/pmc_summary_*.json, tools/profile.sh): FETCH_SIZE / WRITE_SIZE are in KiB
    and come from separate passes; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64
    bytes, so it is doubled (MI355X_MICROARCH.md, HBM section).  The summary carries the hash of
    the kernel sources it was measured on (tools/pmc_summary.py); a profile of OTHER code is not
    quoted: returns (None, reason).
    """
    for rnd in ('r6', 'r5', 'r4', 'r3', 'r2', 'r1'):
        path = os.path.join(ROOT, 'profiles', rnd, summary)
        if os.path.exists(path):
            break
    else:
        return None, 'no PMC profile committed for this workload'
    try:
        doc = json.load(open(path))
        c = doc['counters'][kernel]
        n_f, n_w = c['_dispatches_pmc3'], c['_dispatches_pmc4']
        traffic = (2. * c['FETCH_SIZE'] / n_f + c['WRITE_SIZE'] / n_w) * 1024.
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None, 'profile %s holds no counters for %s' % (os.path.relpath(path, ROOT), kernel)
    if doc.get('kernel_source_sha') != kernel_source_hash():
        return None, ('stale: %s was measured on kernel sources %s, this tree is %s' %
                      (os.path.relpath(path, ROOT), doc.get('kernel_source_sha'),
                       kernel_source_hash()))
    return traffic, os.path.relpath(path, ROOT)


def make_mpc(workload, seed):
    from explicit_hybrid_mpc_amd import examples
    if workload == 'config4':
        return examples.integrator_chain_mpc()
    if workload == 'config3':
        return examples.pwa_mpc(seed=seed)
    return examples.linear_mpc(seed=seed, cost='quadratic' if workload == 'config2q' else 'inf')


def _cpu_worker(job):
    """One host core: its share of the oracle's work list, grown for a bounded wall time."""
    workload, seed, eps_a, eps_r, nodes, work, seconds = job
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    orc = OracleCPU(make_mpc(workload, seed), eps_a, eps_r)
    part = PartitionCPU(orc, max_nodes=0)
    part.nodes = nodes
    part._work = work
    t0 = time.perf_counter()
    while part._work and time.perf_counter() - t0 < seconds:
        part.max_nodes = part.visits + 5
        part.resume()
    closed = sum(1 for nd in part.nodes.values()
                 if nd.get('leaf') and nd.get('is_epsilon_suboptimal'))
    return orc.n_solves, part.visits, time.perf_counter() - t0, closed


def usable_cores():
    """Cores this process may really use: the affinity mask and the cgroup CPU quota cap
    os.cpu_count() (a container on a 256-thread host is typically given far fewer)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]       # cgroup v2
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:                                                                    # cgroup v1
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                n = min(n, max(1, int(quota / float(period) + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(workload, seed, eps_a, eps_r, seconds):
    """
    The oracle (CPU restatement of lib/worker.py + lib/oracle.py, HiGHS) on ALL host cores, laid
    out like the reference's run (lib/prepare.py:161, lib/scheduler.py:620-642): one process per
    core; the main process grows the top of the SAME partition until its work list (the
    reference's task queue: one task per open leaf) holds a task per core, deals it round-robin,
    and every process grows its tasks for `seconds`.
    """
    import multiprocessing as mp
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    from oracle import geometry
    from explicit_hybrid_mpc_amd import examples
    host_cores = os.cpu_count() or 1
    cores = usable_cores()
    mpc = make_mpc(workload, seed)
    V = examples.box_vertices(examples.theta_box(mpc))
    roots, locs = geometry.delaunay_simplices(V)
    orc = OracleCPU(mpc, eps_a, eps_r)
    top = PartitionCPU(orc, max_nodes=0)
    t0 = time.perf_counter()
    top.run(roots, locs, 'ecc')
    # breadth first (oldest task first) so the queue widens instead of diving
    while top._work and len(top._work) < cores and time.perf_counter() - t0 < 0.5 * seconds:
        top._work.insert(0, top._work.pop())      # rotate: next visit takes the OLDEST entry
        top.max_nodes = top.visits + 1
        top.resume()
    t_top = time.perf_counter() - t0
    work = list(top._work)
    n_proc = max(1, min(cores, len(work)))
    jobs = []
    for r in range(n_proc):
        mine = work[r::n_proc]
        jobs.append((workload, seed, eps_a, eps_r, {loc: top.nodes[loc] for loc, _ in mine},
                     mine, seconds))
    t1 = time.perf_counter()
    # one process per core means ONE thread per process: without this every numpy / SciPy import
    # starts a BLAS pool as wide as the machine (256 x 256 threads on the GPU box)
    saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS',
                                            'MKL_NUM_THREADS')}
    os.environ.update({k: '1' for k in saved})
    try:
        with mp.get_context('spawn').Pool(n_proc) as pool:
            # bounded: a worker that dies at start-up must not hang the bench
            res = pool.map_async(_cpu_worker, jobs).get(timeout=3 * seconds + 120)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.perf_counter() - t1
    solves = sum(r[0] for r in res)
    visits = sum(r[1] for r in res)
    busy = max(r[2] for r in res)
    closed = sum(r[3] for r in res)
    # the same work in the units of the device line next to it: every node visit is one
    # epsilon-suboptimality decision (the oracle call the device answers with or without an LP),
    # a closed leaf is a region
    return dict(value=solves / busy, unit='LP solves/s', cores=n_proc, host_cores=host_cores,
                usable_cores=cores,
                regions_per_s=closed / busy, oracle_calls_answered_per_s=visits / busy,
                kind='port',
                sample='%d processes (one per usable core: affinity / cgroup quota; the top of the partition, %d node visits / '
                       '%d LP solves in %.1f s on one core, produced %d tasks, dealt round-robin), '
                       '%.1f s each on the same partition: %d node visits, %d HiGHS LP solves '
                       '(oracle/partition_cpu.py; %.1f s wall incl. process start)' %
                       (n_proc, top.visits, orc.n_solves, t_top, len(work), busy, visits, solves,
                        wall))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--workload', choices=['config2', 'config3', 'config4', 'config2q', 'config5',
                                           'explicit'],
                    default='config2',
                    help='config2 = the BASELINE.json metric (default); config3 = the 2-mode PWA '
                         'hybrid instance (32 commutations, mixed-integer oracles on the device '
                         'engine of csrc/ehm_hybrid.h; BASELINE.json configs[2]); config4 = n_x=6 '
                         'n_u=3 N=10 box-constrained instance on the wide kernels (a parity-test '
                         'configuration, timed for the record); config2q = config 2 with the '
                         'quadratic cost of the same weights (convex QP / QCQP oracles, the '
                         "reference's cvx.quad_form cost class; not the headline metric)")
    ap.add_argument('--abs-frac', type=float, default=None,
                    help='eps_a rule of lib/examples.py:42-46 (default 0.02; config4: 0.4)')
    ap.add_argument('--eps-r', type=float, default=None, help='default 1e-2; config4: 0.25')
    ap.add_argument('--max-nodes', type=int, default=1 << 22)
    ap.add_argument('--max-depth', type=int, default=None,
                    help='leave nodes open at this tree depth (config3 default: 24 -- its optimal '
                         'cost jumps where a mode stops being admissible, and no simplicial '
                         'partition is epsilon-suboptimal ACROSS a jump: the refinement along that '
                         'surface never ends, in the reference as here)')
    ap.add_argument('--shard-min-frontier', type=int, default=0,
                    help='frontier size at which it is dealt over the ranks (0 = 64 per rank)')
    ap.add_argument('--balance', choices=['static', 'dynamic'], default='static',
                    help='N > 1: static = ONE launch of the persistent frontier kernel per rank '
                         'from the roots, dealt at a tree depth by a hash of the node path, no '
                         'collective in the data path (measured load imbalance 1.024 at 8 '
                         'shards); dynamic = level-synchronous sweeps '
                         'with an all-gather of the frontier sizes and point-to-point node '
                         'transfers every --sweeps-per-round sweeps')
    ap.add_argument('--sweeps-per-round', type=int, default=2,
                    help='frontier sweeps between two rebalancing rounds (N > 1)')
    ap.add_argument('--status-dir', default=None,
                    help='write status.txt / statistics.pkl (reference formats) there; adds one '
                         'progress read-back (and, N > 1, one all-gather) per round -- off by '
                         'default, the headline number is measured without it')
    ap.add_argument('--cpu-seconds', type=float, default=15.)
    ap.add_argument('--steal-slice', type=int, default=2048,
                    help='config5 with several --host-streams: cell visits between two looks at '
                         'the requests of idle handles (also the largest round)')
    ap.add_argument('--steal', action='store_true',
                    help='config5 with several --host-streams: a handle that finds no root left '
                         'takes pending cells from the busy ones (frontier.LocalExchange on '
                         'ehm_frontier_take / _give).  Off by default: the handles of ONE process '
                         'share one GPU, and a single root shared by them costs 17-35 %% more LPs '
                         '(no shared memo) for no gain in wall time '
                         '(profiles/r6/take_give_one_root.txt); between RANKS it is what evens out '
                         'an expensive root (distributed.grow_roots_sharded(steal=True))')
    ap.add_argument('--host-streams', type=int, default=1,
                    help='config5, native driver: driver handles that grow different roots at the '
                         'same time from as many interpreter threads (the scale entry uses 6)')
    ap.add_argument('--scale-seconds', type=float, default=600.,
                    help='soft time limit of the config5_scale entry of "secondary" (0 = skip it)')
    ap.add_argument('--queries', type=int, default=1 << 21,
                    help='explicit: states evaluated per step')
    ap.add_argument('--secondary-cpu-seconds', type=float, default=6.,
                    help='CPU-baseline time of each entry of the "secondary" list')
    ap.add_argument('--no-secondary', action='store_true',
                    help='headline only: skip the few steps of config3 / config4 / config2q / '
                         'config5 that the default invocation appends as "secondary"')
    ap.add_argument('--regions', type=int, default=0,
                    help='config5: regions to close per rank before a step stops (default: none, the '
                         'cells are grown to completion)')
    ap.add_argument('--cells', type=int, default=0,
                    help='config5: root cells of the box that are grown (default: one per rank)')
    ap.add_argument('--roots', choices=['delaunay', 'kuhn'], default='delaunay',
                    help='config5: the root cells.  delaunay (default) = the reference\'s roots, '
                         'tools.delaunay_roots of the box (lib/tools.py:152-189: Qhull, 34 573 '
                         'simplices at p = 8) in Qhull order; kuhn = the 8 Kuhn cells of the cyclic '
                         'coordinate orders (rounds 3-4)')
    ap.add_argument('--driver', choices=['native', 'python'], default='native',
                    help='config5: native = the round loop, the searches\' bookkeeping, the block '
                         'condensation and the launches in C++ behind ehm_frontier_run '
                         '(include/ehm_frontier.h); python = bnb_frontier.grow_frontier (rounds 3-4)')
    ap.add_argument('--seconds', type=float, default=0.,
                    help='config5 with --regions: soft limit of a step; the cell in progress is cut '
                         'short (its pending leaves stay open and are reported) and no further cell '
                         'is started')
    ap.add_argument('--progress-file', default=None,
                    help='config5: append one JSON line per finished group of cells (a run cut '
                         'short by a time limit leaves what it measured)')
    ap.add_argument('--cells-at-once', type=int, default=0,
                    help='config5: grow a rank\'s cells in groups of this many (0 = all of them '
                         'together, sharing the launches); 1 = one cell after the other, each to '
                         'completion -- what several host processes per GPU run (EHM_BENCH_BACKEND='
                         'gloo + torch.distributed.run: the ranks share the GPU, the searches of '
                         'one overlap the launches of the others)')
    ap.add_argument('--order', choices=['lcss-first', 'fifo', 'deepest'], default='lcss-first',
                    help='config5: visiting order of the search driver (bnb_frontier.grow_frontier; '
                         'the finished tree does not depend on it).  lcss-first (default) '
                         'completes subtrees; fifo goes level by level -- with --regions it closes '
                         'the cells that need no refinement first')
    ap.add_argument('--round-cap', type=int, default=4096,
                    help='config5: nodes visited together in one round of the search driver (their '
                         'problems share the launches)')
    ap.add_argument('--max-visits', type=int, default=None,
                    help='config5: cap on the node visits of a step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--solver', type=int, default=2, help='kernel generation (1 or 2)')
    ap.add_argument('--no-mid-first', action='store_true',
                    help='persistent kernel WITHOUT the midpoint solve before the suboptimality '
                         'test (the round-1 flow; A/B measurements)')
    ap.add_argument('--no-inherit-witness', action='store_true',
                    help='open nodes do NOT hand the point that proved them open to their '
                         'children (A/B measurements)')
    ap.add_argument('--engine', type=int, default=1,
                    help='1 = persistent frontier kernel (one launch per partition; single rank, '
                         'shared-block kernels), 0 = level-synchronous sweeps')
    ap.add_argument('--decide-full', action='store_true',
                    help='solve the suboptimality-test LPs to full accuracy (no sign-only stop)')
    return ap.parse_args(argv)



def measure(args, ctx):
    """One workload on the process group of ``ctx``; rank 0 returns the result line (a dict)."""
    import torch
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import distributed
    backend, rank, world, device_index = (ctx['backend'], ctx['rank'], ctx['world'],
                                          ctx['device_index'])
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools

    wide = args.workload == 'config4'
    hybrid = args.workload == 'config3'
    if args.abs_frac is None:
        args.abs_frac = {'config4': 0.4, 'config3': CONFIG3[0],
                         'config2q': 0.1}.get(args.workload, 0.02)
    if args.eps_r is None:
        args.eps_r = {'config4': 0.25, 'config3': CONFIG3[1],
                      'config2q': 0.1}.get(args.workload, 1e-2)
    if args.max_depth is None:
        args.max_depth = CONFIG3[2] if hybrid else 0
    quad = args.workload == 'config2q'
    mpc = make_mpc(args.workload, args.seed)
    can = mpc.compile()
    gp = engine.GpuProblem(can, 1., 1., device=device_index)
    if not wide:
        gp.set_solver(args.solver)
    gp.set_option('decide_full', 1 if args.decide_full else 0)
    gp.set_option('timing', 1)           # kernel seconds / solves by kind of the hybrid engine
    if args.no_mid_first:
        gp.set_option('mid_first', 0)
    if args.no_inherit_witness:
        gp.set_option('inherit_witness', 0)
    hybrid = args.workload == 'config3'
    static = args.balance == 'static' and not args.status_dir
    # both balancing modes run the persistent frontier kernel: static = one launch per rank,
    # dynamic = budgeted rounds of it
    # (wide LPs: the LDS-resident family has a persistent kernel of its own, single rank only --
    # csrc/ehm_k4.hip, k4_persist; EHM_K4=0 / EHM_K4_PERSIST=0 select the streaming kernels / the
    # level-synchronous sweeps)
    wide_persist = (wide and world == 1 and args.engine == 1 and not args.status_dir and
                    os.environ.get('EHM_K4') != '0' and os.environ.get('EHM_K4_PERSIST') != '0')
    persistent = (args.engine == 1 and args.solver == 2 and not wide and not hybrid and
                  not args.status_dir) or wide_persist
    # (the persistent kernel exists at one solver width, k2_persist, and -- where a pair of
    # instances is compiled, as for this workload -- at two, kp_persist; the library picks)
    kname = (wide_kernel('persist') if wide_persist else wide_kernel('lcss_decide')) if wide else \
        'k2_simplex_batch' if hybrid else (
        ('kp_persist' if not quad else 'k2_persist') if persistent else
        'k2_lcss_decide' if args.solver == 2 else 'k_lcss_decide')
    pmc_file = {'config4': 'pmc_summary_wide.json', 'config3': 'pmc_summary_config3.json',
                'config2q': 'pmc_summary_quad.json'}.get(args.workload, 'pmc_summary_bench.json')
    half = examples.theta_box(mpc)
    V = examples.box_vertices(half)
    # eps_a by the reference's rule (lib/examples.py:42-46), evaluated on the GPU oracle
    J_abs, _, _ = gp.solve_pt(args.abs_frac * V)
    eps_a = float(np.max(J_abs))
    gp.set_eps(eps_a, args.eps_r)
    roots, _ = ehm_tools.delaunay_roots(V)
    if args.shard_min_frontier <= 0:
        # static: deal late enough for the shares to even out (tools/shard_balance.py: max/mean
        # 1.31 at 64 nodes per rank, 1.04 at 1024); dynamic: the top of the tree is latency-bound
        # on any number of GPUs, deal early and rebalance often
        args.shard_min_frontier = (1024 if static else 64) * world
    shard = distributed.shard_spec(rank, world, args.shard_min_frontier)
    # static: ONE persistent launch per rank from the roots, dealt at a tree depth by path code
    # (128 nodes per rank at the deal depth: with the round-3 kernel the replicated top of the tree
    # costs more than the imbalance of an early deal -- 10.6 ms per shard at depth 6 against 11.1 at
    # depth 10 for 8 shards, profiles/r3/shard_balance_deal_1p6M_nodes.txt)
    deal_depth = distributed.deal_depth_for(len(roots), world, 64 if hybrid else 128) \
        if (static and world > 1) else 0

    xdev = ('cuda:%d' % device_index) if backend == 'nccl' else None
    publisher = None
    if args.status_dir and rank == 0:
        from explicit_hybrid_mpc_amd import status as ehm_status
        os.makedirs(args.status_dir, exist_ok=True)
        publisher = ehm_status.MainStatusPublisher(
            float(np.prod(2. * half)), os.path.join(args.status_dir, 'status.txt'),
            os.path.join(args.status_dir, 'statistics.pkl'))

    def step():
        if (world == 1 or static) and not args.status_dir:
            return gp.partition(roots, action='ecc', max_nodes=args.max_nodes, export=False,
                                shard=shard, with_volume=False,
                                engine=args.engine, max_depth=args.max_depth,
                                deal_depth=deal_depth)
        info, log, rounds = distributed.run_balanced(
            gp, roots, action='ecc', max_nodes=args.max_nodes,
            min_frontier=args.shard_min_frontier, sweeps_per_round=args.sweeps_per_round,
            device=xdev, export=False, status=publisher, max_depth=args.max_depth,
            publish_status=bool(args.status_dir),
            engine='persistent' if (args.engine == 1 and args.solver == 2 and not wide and
                                    not hybrid) else 'sweeps')
        info['rounds'] = rounds
        info['moved'] = sum(len(e['ids']) for e in log if e['kind'] == 'give')
        return info

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # a rank whose engine raises (pool exhausted, numeric error) still reaches every barrier; the
    # ranks then agree on the failure and leave together instead of hanging in a collective
    failure = None
    try:
        for _ in range(args.warmup):
            step()
    except Exception as e:
        failure = e
    barrier()
    t0 = time.perf_counter()
    infos = []
    try:
        if failure is None:
            infos = [step() for _ in range(args.steps)]
    except Exception as e:
        failure = e
    barrier()
    elapsed = time.perf_counter() - t0
    _, any_failed = distributed.allreduce_counters(
        [0. if failure is None else 1.],
        device=('cuda:%d' % device_index) if backend == 'nccl' else 'cpu')
    if any_failed[0] > 0:
        sys.stderr.write('bench.py rank %d: %s\n' % (
            rank, failure if failure is not None else 'another rank failed'))
        gp.close()
        raise SystemExit(1)
    # what the caller gets back: one more partition WITH the flat export (device -> host copy of
    # every record + breadth-first relabelling), outside the timed region, reported next to it
    # (the branch is the worker's output, lib/worker.py:456-458).  One untimed pass first: it pins
    # the host arrays of the export and sizes the device staging, like the warm-up of the steps.
    ms_export = ms_with_export = None
    if world == 1:
        def with_export():
            flat = gp.partition(roots, action='ecc', max_nodes=args.max_nodes, export=True,
                                with_volume=False, engine=args.engine, max_depth=args.max_depth)
            n = flat.n_nodes
            del flat
            return n
        with_export()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = max(2, min(args.steps, 5))
        for _ in range(reps):
            with_export()
        ms_with_export = 1e3 * (time.perf_counter() - t1) / reps
        ms_export = ms_with_export - 1e3 * elapsed / args.steps
    # N > 1: the regions of all ranks together must be the regions of ONE unsharded partition (a
    # node's fate depends on its own record only, so the tree does not depend on who grows which
    # part of it): rank 0 grows the whole tree once more, outside the timed region, and the line
    # carries both counts
    whole = None
    if world > 1 and rank == 0:
        try:
            whole = gp.partition(roots, action='ecc', max_nodes=args.max_nodes, export=False,
                                 with_volume=False, engine=args.engine, max_depth=args.max_depth)
        except Exception as e:          # reported, not fatal: the timed numbers stand
            whole = dict(error=str(e))
    # totals over ranks (max time, summed work)
    keys = ['lp_solves', 'ipm_iters', 'n_nodes', 'n_closed', 'ref_solves', 'decide_solves',
            'decide_iters', 'cert_closed', 'witness_open', 'witness_inherited',
            'midpoints_shared', 'witness_table']
    if rank > 0:
        # the top of the tree is grown identically on every rank: count it once (rank 0)
        for i in infos:
            i['lp_solves'] -= i['replicated_solves']
            i['n_closed'] -= i['replicated_closed']
            i['n_nodes'] -= i['replicated_nodes']
    kinds = [float(sum(i['kind_solves'][q] for i in infos)) for q in range(5)] + \
        [float(sum(i['kind_iters'][q] for i in infos)) for q in range(5)]
    local = [float(sum(i[k] for i in infos)) for k in keys] + \
        [elapsed, sum(i['decide_seconds'] for i in infos),
         sum(i['expand_seconds'] for i in infos),
         float(sum(i.get('moved', 0) for i in infos)),
         float(sum(i.get('rounds', 0) for i in infos))] + kinds
    red_dev = ('cuda:%d' % device_index) if backend == 'nccl' else 'cpu'
    tot, mx = distributed.allreduce_counters(local, device=red_dev)
    per_rank = distributed.allgather_counts([int(local[0])], device=red_dev)[:, 0]
    elapsed_max = float(mx[len(keys)])
    agg = dict(zip(keys, tot[:len(keys)]))
    decide_s = float(mx[len(keys) + 1])
    expand_s = float(mx[len(keys) + 2])
    kind_solves = [float(v) for v in tot[len(keys) + 5:len(keys) + 10]]
    kind_iters = [float(v) for v in tot[len(keys) + 10:len(keys) + 15]]
    out = None
    if rank == 0:
        K = args.steps
        info0 = infos[-1]
        n, m, p = can.n, can.m, can.p
        # (columns, columns with MPC entries, rows) of the five problem kinds (ehm_dev.h: LP_*)
        dims = [(n, n, m), (n + 1, n + 1, m + 1), (n + p, n + p, m + p + 1),
                (n + p + 1, n + p, m + p + 3), (n + p + 1, n + p + 1, m + p + 2)]
        if not hybrid and not persistent:
            # sweeps of a single-commutation handle: the decide kernel alone is timed
            kind_iters = [agg['ipm_iters'] - agg['decide_iters'], 0., 0., agg['decide_iters'], 0.]
        f_survey = [kind_iters[q] * flops_per_iteration(dims[q][0], dims[q][2]) for q in range(5)]
        lay = gp.layout()         # what the solver eliminates (ehm_problem_layout)
        f_exec = [kind_iters[q] * flops_executed_per_iteration(
            dims[q][0], dims[q][1], dims[q][2], lay['nE'], lay['LE']) for q in range(5)]
        simplex_kinds, point_kinds = (2, 3, 4), (0, 1)
        B_node = node_bytes(p, can.n_u, can.deltas.shape[1])      # 301 B at config 2
        grad_bytes = 8 * (p + 1) * p if agg['cert_closed'] > 0 else 0
        closed, nodes = agg['n_closed'], agg['n_nodes']
        splits = (nodes - K * len(roots)) / 2.
        visits = (agg['decide_solves'] + agg['cert_closed'] + agg['witness_open'] +
                  agg['witness_inherited'] + agg['witness_table'])
        if persistent:
            # ONE kernel per partition: suboptimality tests AND splits / midpoint solves
            flops, flops_x = sum(f_survey), sum(f_exec)
            hbm_alg = visits * (B_node + 8) + splits * (3 * B_node + 16)
            hbm_grad = (visits + 3 * splits) * grad_bytes
        elif hybrid:
            # dominant kernel = the batched problems over a simplex (slack / phase one / min);
            # per instance: a work-list entry, the record's vertices + vertex costs in, the
            # optimum, the maximiser's weights and the status out
            flops = sum(f_survey[q] for q in simplex_kinds)
            flops_x = sum(f_exec[q] for q in simplex_kinds)
            n_sx = sum(kind_solves[q] for q in simplex_kinds)
            hbm_alg = n_sx * (12 + 8 * ((p + 1) * p + (p + 1)) + 8 + 8 * (p + 1) + 4)
            hbm_grad = 0.
        else:
            flops, flops_x = f_survey[3], f_exec[3]
            hbm_alg = agg['decide_solves'] * (B_node + 8)
            hbm_grad = agg['decide_solves'] * grad_bytes
        # per GPU: the ranks' launches run side by side, each against its own peak
        achieved = flops / world / decide_s / 1e12
        flops_x /= world
        hbm_alg /= world
        hbm_grad /= world
        launches = max(info0['decide_launches'] * K, 1)
        traffic, traffic_src = pmc_traffic(kname, pmc_file)
        n_slack, m_slack = dims[3][0], dims[3][2]
        n_pt, m_pt = dims[0][0], dims[0][2]
        names = {'config2': 'configs[1]: n_x=4 n_u=2 N=5 p=4 linear MPC',
                 'config2q': 'configs[1]: n_x=4 n_u=2 N=5 p=4 linear MPC',
                 'config3': 'configs[2] (NOT the headline configuration): 2-mode PWA hybrid '
                            'system, n_x=4 n_u=2 N=5 p=4, %d commutations, mixed-integer oracles'
                            % can.n_delta,
                 'config4': 'configs[3] (NOT the headline configuration): n_x=6 n_u=3 N=10 p=6 '
                            'box-constrained chain'}
        out = {
            'metric': 'oracle LP solves/sec + final regions/sec, 4-state 2-input N=5 hybrid MPC',
            'value': agg['lp_solves'] / elapsed_max,
            'unit': 'LP solves/s',
            'regions_per_s': closed / elapsed_max,
            'oracle_calls_answered_per_s': agg['ref_solves'] / elapsed_max,
            'value_note': 'value counts the LP solves EXECUTED; the same partition issues '
                          'reference_equivalent_solves_per_step oracle calls whatever the engine '
                          'does, and every call a bound or a witness answers without an LP lowers '
                          'value while ms_per_step falls -- compare ms_per_step, regions_per_s and '
                          'oracle_calls_answered_per_s between builds',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed_max / K,
            'ms_export': ms_export,
            # the second headline figure: a step INCLUDING the flat export of the whole tree
            # (breadth-first numbering on the device, one gather pass, copies into page-locked
            # host arrays) -- what a caller of alg_call pays before it holds the branch
            'ms_per_step_with_export': ms_with_export,
            'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': names[args.workload] +
                            ', %s (n=%d m=%d), seed %d, eps_r=%g, eps_a=%.6g '
                            '(abs_frac=%g), %d Delaunay roots%s' % (
                                'quadratic-cost QP / QCQP oracle (NOT the headline metric)'
                                if quad else 'inf-norm LP oracle',
                                can.n, can.m, args.seed, args.eps_r, eps_a, args.abs_frac,
                                len(roots),
                                ', nodes left open at depth %d' % args.max_depth
                                if args.max_depth else ''),
                'regions_per_step': closed / K,
                'nodes_per_step': nodes / K,
                'open_leaves_at_max_depth_per_step': (nodes - splits - closed) / K,
                'lp_solves_per_step': agg['lp_solves'] / K,
                'lp_solves_by_kind_per_step': dict(zip(
                    ('point', 'point_phase_one', 'min_over_simplex', 'slack', 'simplex_phase_one'),
                    [v / K for v in kind_solves])),
                'reference_equivalent_solves_per_step': agg['ref_solves'] / K,
                'reference_equivalent_solves_per_s': agg['ref_solves'] / elapsed_max,
                'leaves_closed_without_lp_per_step': agg['cert_closed'] / K,
                'nodes_proved_open_by_midpoint_per_step': agg['witness_open'] / K,
                'nodes_proved_open_by_inherited_witness_per_step': agg['witness_inherited'] / K,
                'midpoint_optima_taken_from_the_table_per_step': agg['midpoints_shared'] / K,
                'nodes_proved_open_by_another_edges_midpoint_in_the_table_per_step':
                    agg['witness_table'] / K,
                'mean_ipm_iterations': agg['ipm_iters'] / max(agg['lp_solves'], 1),
                'sweeps': info0['sweeps'], 'tree_depth': info0['max_depth'],
                'min_decision_margin': info0['min_margin'],
                'near_threshold_decisions_per_step': info0['near_threshold'],
                'kernels': 'wide (one workgroup per LP, MFMA normal matrix)' if wide else
                           'generation %d%s' % (args.solver, ' with the quadratic block' if quad
                                                else ''),
                'engine': 'persistent frontier kernel (one launch per partition)' if persistent
                          else 'multi-commutation device engine (frontier sweeps, work lists and '
                               'decisions on the device)' if hybrid
                          else 'level-synchronous sweeps',
                'suboptimality_test': 'full accuracy' if (args.decide_full or args.solver == 1
                                                          or hybrid) else
                                      'sign-only stop (lower bound of |t*| recorded)',
                'parallelism': 'frontier dealt round-robin over %d GPU(s)' % world +
                               ('' if world == 1 else
                                ': one persistent launch per rank from the roots, the tree above '
                                'depth %d replicated, the nodes of that depth dealt by a hash of '
                                'their path (static; no data-path collective)' % deal_depth
                                if static else
                                ': rank 0 owns the roots; budgeted rounds of the persistent '
                                'kernel (4096 node visits, doubling), after each an all-gather of '
                                'the frontier sizes and point-to-point node records (every s-th '
                                'entry of a donor frontier); once a round ends balanced with >= '
                                '4096 nodes on every rank, one unbudgeted launch per rank'
                                if (args.engine == 1 and args.solver == 2 and not wide) else
                                ', rebalanced every %d sweeps (all-gather of frontier sizes + '
                                'point-to-point node records)' % args.sweeps_per_round),
                'rebalance_rounds_per_step': float(mx[len(keys) + 4]) / K,
                'nodes_moved_per_step': float(tot[len(keys) + 3]) / K,
                'lp_solves_per_rank': [int(v) for v in per_rank],
                'tree_identity': None if whole is None else (
                    dict(ok=False, error=whole['error']) if 'error' in whole else
                    dict(regions_of_one_unsharded_partition=int(whole['n_closed']),
                         regions_summed_over_the_ranks_per_step=closed / K,
                         ok=bool(abs(closed / K - whole['n_closed']) < 0.5))),
                'load_imbalance_max_over_mean': distributed.imbalance(per_rank),
            },
            # where the wavefronts of the persistent kernel spent their time (fractions of the
            # summed residency; in-kernel 100 MHz clock, ehm_tree_info.persist_ticks)
            'persist_ticks': (lambda t: None if not t or not t[0] else {
                'starved_waiting_for_a_queue_slot': t[1] / t[0],
                'waiting_for_a_midpoint_being_solved': t[2] / t[0],
                'midpoint_solves': t[3] / t[0], 'suboptimality_test_solves': t[4] / t[0],
                'from_pop_to_midpoint_claim': t[6] / t[0], 'child_records_and_pushes': t[7] / t[0],
                'everything_else': 1. - (t[1] + t[2] + t[3] + t[4] + t[6] + t[7]) / t[0],
                'midpoint_waits_per_step': t[5] / K, 'nodes_put_back_per_step': t[8] / K,
                'mean_resident_ms_per_wavefront': t[0] / 1e5 / K / (256 * 12),
            })([float(sum(i['persist_ticks'][q] for i in infos)) for q in range(10)]
               if persistent else None),
            'roofline': {
                # the path is compute-bound on FP64; the contract's two values are
                # "hbm" | "mfma": the wide kernels form their normal matrix on the matrix cores,
                # the shared-block kernels execute no MFMA at all -- FP64 vector FMA issue
                'bound': 'mfma' if wide else 'valu-fp64', 'kernel': kname,
                'note': ('LDS-resident reduced block (57 -> 37 factorised columns), normal matrix '
                         'and the trailing updates of the blocked factorisation on '
                         'v_mfma_f64_16x16x4_f64 (csrc/ehm_ipm4.h); peak = FP64 matrix = vector '
                         'peak of MI355X' if wide else
                         'one launch holds the slack LPs (n=%d m=%d) and the midpoint LPs (n=%d '
                         'm=%d); the solver eliminates %d epigraph columns from the Newton systems '
                         'and factorises %d / %d; FP64 issue bound (vector FMA + one matrix-core '
                         'tile for the normal matrix, which holds the same double-precision pipe); '
                         'peak = FP64 vector (= matrix) peak of MI355X'
                         % (n_slack, m_slack, n_pt, m_pt, lay['nE'], n_slack - lay['nE'],
                            n_pt - lay['nE']) if persistent else
                         'FP64 vector FMA bound (no f64 contraction >= 32 wide at n=25); peak = '
                         'FP64 vector (= matrix) peak of MI355X'),
                'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / FP64_PEAK_TFLOPS,
                # the same launches priced with what the solver really executes
                # (flops_executed_per_iteration: matrix-core tile incl. its padding, unsymmetric
                # elimination at the compiled width of the FACTORISED columns)
                'achieved_executed': flops_x / decide_s / 1e12,
                'frac_executed': flops_x / decide_s / 1e12 / FP64_PEAK_TFLOPS,
                'flop_per_ipm_iteration': flops_per_iteration(n_slack, m_slack),
                'flop_executed_per_ipm_iteration': flops_executed_per_iteration(
                    n_slack, dims[3][1], m_slack, lay['nE'], lay['LE']),
                'eliminated_columns': lay['nE'],
                'factorised_columns': {'slack': n_slack - lay['nE'], 'point': n_pt - lay['nE']},
                'traffic': traffic,
                'traffic_unit': 'HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)',
                'traffic_source': traffic_src,
                'algorithmic_bytes_per_launch': hbm_alg / launches,
                'algorithmic_bytes_note': 'SURVEY 8(d): B_node = %d B; node visit = B_node + 8, '
                                          'split = 3 B_node + 16' % B_node if not hybrid else
                                          'per instance of the batch: list entry + vertices + '
                                          'vertex costs in, optimum + weights + status out',
                'gradient_bytes_per_launch': hbm_grad / launches,
                'kernel_seconds': decide_s, 'launches': info0['decide_launches'] * K,
                'expand_kernel': None if persistent else
                                 {'kernel': 'k2_point_batch' if hybrid else 'k2_lcss_expand',
                                  'achieved': sum(f_survey[q] for q in point_kinds) /
                                  max(expand_s, 1e-12) / 1e12,
                                  'kernel_seconds': expand_s},
                'hbm': {'achieved': hbm_alg / decide_s / 1e9, 'peak': HBM_PEAK_GBS,
                        'unit': 'GB/s', 'frac': hbm_alg / decide_s / 1e9 / HBM_PEAK_GBS,
                        'bytes_per_node': B_node + 8,
                        'note': 'not the binding resource: a node is ~0.3-0.9 KB of traffic against '
                                '~2 Mflop of FP64 work, so north_star\'s ">= 40 % of the HBM '
                                'roofline" is structurally out of reach (SURVEY 8(d))'},
            },
        }
        ident = out['config']['tree_identity']
        if ident is not None and not ident.get('ok'):
            sys.stderr.write('bench.py: the ranks together did NOT grow the tree of one '
                             'unsharded partition: %s\n' % (ident,))
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.workload, args.seed, eps_a, args.eps_r,
                                               args.cpu_seconds)
        else:
            out['cpu_baseline'] = None
    gp.close()
    return out


# ---- config 5: mode sequences that cannot be enumerated -----------------------------------------
CONFIG5 = dict(abs_frac=0.2, eps_r=1e-3, regions=1 << 30, N=8)    # regions: the cell(s) to completion


def kuhn_cell(half, k):
    """The k-th Kuhn simplex of the box [-half, half] (one of p! that tile it): the corner path
    of the k-th cyclic rotation of the coordinate order."""
    p = len(half)
    order = np.roll(np.arange(p), -(k % p))
    R = np.tile(-np.asarray(half, dtype=np.float64), (p + 1, 1))
    for step in range(1, p + 1):
        R[step:, order[step - 1]] = half[order[step - 1]]
    return R


def _cpu5_worker(job):
    """One host core: the search driver on the CPU statement of the table (HiGHS on the
    uncondensed relaxations, oracle/prefix_bb.py) on its own cell, for a bounded wall time."""
    seed, eps_a, eps_r, cell, seconds = job        # cell: (p+1, p) vertices of the root cell
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier, examples
    from explicit_hybrid_mpc_amd.tree import NodeData, Tree
    from oracle import geometry, prefix_bb
    mpc = examples.pwa4_mpc(N=CONFIG5['N'], seed=seed)
    orc = bnb.PrefixOracle(mpc, eps_a, eps_r, table=prefix_bb.CpuPrefixTable(mpc))

    def split_batch(R):
        S1, S2, ij = [], [], []
        for r in R:
            a, b, e = geometry.split_along_longest_edge(r)
            S1.append(a), S2.append(b), ij.append(e)
        return np.array(S1), np.array(S2), np.array(ij)
    t0 = time.perf_counter()
    tree = Tree(NodeData(vertices=np.array(cell, dtype=np.float64)))
    stats = bnb_frontier.grow_frontier(orc, tree, 'ecc', handoff=False, split_batch=split_batch,
                                       round_cap=4, order='lcss-first', deadline=t0 + seconds)
    return orc.table.lp_solves, stats['host_visits'], stats['regions'], time.perf_counter() - t0


def cpu_baseline_config5(seed, eps_a, eps_r, seconds, cell_vertices=None):
    """The same search driver (bnb_frontier.grow_frontier on bnb.PrefixOracle) with the CPU
    statement of the table, one process per usable core, each on its own cell of the box."""
    import multiprocessing as mp
    cores = usable_cores()
    saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS',
                                            'MKL_NUM_THREADS')}
    os.environ.update({k: '1' for k in saved})
    t1 = time.perf_counter()
    try:
        with mp.get_context('spawn').Pool(cores) as pool:
            if cell_vertices is None:
                from explicit_hybrid_mpc_amd import examples
                half = examples.theta_box(examples.pwa4_mpc(N=CONFIG5['N'], seed=seed))
                cell_vertices = lambda c: kuhn_cell(half, c)
            res = pool.map_async(_cpu5_worker, [(seed, eps_a, eps_r, cell_vertices(c), seconds)
                                                for c in range(cores)]).get(timeout=6 * seconds + 180)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.perf_counter() - t1
    busy = max(r[3] for r in res)
    return dict(value=sum(r[0] for r in res) / busy, unit='LP solves/s', cores=cores,
                host_cores=os.cpu_count() or 1, usable_cores=cores, kind='port',
                regions_per_s=sum(r[2] for r in res) / busy,
                node_visits_per_s=sum(r[1] for r in res) / busy,
                sample='%d processes (one per usable core), each the search driver of the device '
                       'path on the CPU statement of the table (oracle/prefix_bb.py: HiGHS on the '
                       'uncondensed prefix relaxations) on its own root cell of the box, %.1f s '
                       'each (rounds of 4 nodes): %d node visits, %d LP solves, %d regions closed '
                       '(%.1f s wall incl. process start)' %
                       (cores, busy, sum(r[1] for r in res), sum(r[0] for r in res),
                        sum(r[2] for r in res), wall))


def measure_config5(args, ctx):
    """
    BASELINE.json configs[4]: n_x = 8, n_u = 3, four modes, N = 8 (65 536 mode sequences),
    eps_r = 1e-3.  A step grows ``--cells`` Kuhn cells of the box (rank r: cells r, r + world,
    ...: independent roots, no data-path collective -- "weak" scaling over the cells) with the
    search oracles of bnb.PrefixOracle until ``--regions`` regions are closed per rank: every
    mixed-integer oracle call is a branch-and-bound over mode prefixes whose relaxations are LPs
    solved on the device (sequences.PrefixTable: blocks of the commutation table written on
    demand).  Everything the searches remember is dropped before every step; the table's blocks
    (problem data) stay resident.
    """
    import torch
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier, distributed, examples
    from explicit_hybrid_mpc_amd.tree import NodeData, Tree
    backend, rank, world, device_index = (ctx['backend'], ctx['rank'], ctx['world'],
                                          ctx['device_index'])
    abs_frac = CONFIG5['abs_frac'] if args.abs_frac is None else args.abs_frac
    eps_r = CONFIG5['eps_r'] if args.eps_r is None else args.eps_r
    regions = args.regions or CONFIG5['regions']
    n_cells = args.cells or world
    mpc = examples.pwa4_mpc(N=CONFIG5['N'], seed=args.seed)
    half = examples.theta_box(mpc)
    V = examples.box_vertices(half)
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=8192, device=device_index)
    # the native driver (include/ehm_frontier.h) owns device tables of its own; the Python oracle
    # stays for the cells the native driver hands back (failed vertex solves: none so far)
    native = None
    natives = []
    if args.driver == 'native':
        from explicit_hybrid_mpc_amd import frontier
        # --host-streams S: S driver handles (own device tables, streams and host threads each) grow
        # different roots at the same time from S interpreter threads -- the native calls release
        # the GIL --, so that one handle's host bookkeeping between its solver calls (a quarter of a
        # deep root's time) and the tails of its launches are filled with the other's kernels
        natives = [frontier.NativeFrontier(mpc, 1., 1., slots=16384, device=device_index)
                   for _ in range(max(1, args.host_streams))]
        native = natives[0]
    t_eps = time.perf_counter()
    # eps_a by the reference's rule (lib/examples.py:42-46): 2^p P_theta searches in lockstep
    eps_a = float(np.max([j for _, _, j in (native.p_theta(abs_frac * V) if native else
                                            bnb_frontier.p_theta_many(orc, abs_frac * V))]))
    t_eps = time.perf_counter() - t_eps
    orc.eps_a, orc.eps_r = eps_a, eps_r
    orc.table.set_eps(eps_a, eps_r)
    for nat in natives:
        nat.set_eps(eps_a, eps_r)
    if args.roots == 'delaunay':
        from explicit_hybrid_mpc_amd import tools as ehm_tools
        root_cells, _ = ehm_tools.delaunay_roots(V)
        n_roots_total = len(root_cells)
        cell_vertices = lambda c: root_cells[c % n_roots_total].copy()
    else:
        n_roots_total = 8
        cell_vertices = lambda c: kuhn_cell(half, c)
    my_cells = list(range(rank, n_cells, world))
    cells_log = []

    # the device table(s): one, or -- where the full model needs the wide kernels -- a table of the
    # short horizon for the prefixes of few steps next to it (sequences.SplitPrefixTable)
    parts = [('short', orc.table.short), ('long', orc.table.long)] if hasattr(orc.table, 'short') \
        else [('long', orc.table)]
    # accounting by table = by horizon: 'h4' ... 'h8'; (n, m, p) of the table's blocks
    hname = lambda tb: 'h%d' % tb.mpc.N
    table_dims = {hname(tb): (tb.gp.can.n, tb.gp.can.m, tb.gp.can.p) for _, tb in parts}

    if native:
        for hz in native.horizons:
            if 'h%d' % hz not in table_dims:
                G_h = frontier.condense_native(mpc, (), hz)[0]
                table_dims['h%d' % hz] = (G_h.shape[1], G_h.shape[0], mpc.n_x)
    nat_acc = dict(calls=dict(P_theta=0, V_R=0, bar_E=0, bar_D=0), expanded=0, stalled=0,
                   solver_seconds=0., driver_seconds=0., open_cells=0, launches=0)

    def snapshot():
        per = {name: dict(lp=0, iters=0, secs=[0., 0.], launches=[0, 0],
                          hist=np.zeros((5, mpc.N + 1), dtype=np.int64)) for name in table_dims}
        for _, tb in parts:             # the Python oracle's tables (eps_a, cells handed back)
            st, e = tb.gp.stats(), per[hname(tb)]
            e['lp'] += tb.lp_solves
            e['iters'] += st['ipm_iters']
            e['secs'] = [a + b for a, b in zip(e['secs'], st['batch_seconds'])]
            e['launches'] = [a + b for a, b in zip(e['launches'], st['batch_launches'])]
            e['hist'][:, :tb.by_length.shape[1]] += tb.by_length
        for nat in natives:
            counts = nat.lp_counts()
            for t, nt in enumerate(nat.table_stats()):
                e = per['h%d' % nt['horizon']]
                e['lp'] += int(counts[t].sum())
                e['iters'] += nt['ipm_iters']
                e['secs'] = [a + b for a, b in zip(e['secs'], nt['batch_seconds'])]
                e['launches'] = [a + b for a, b in zip(e['launches'], nt['batch_launches'])]
                e['hist'] += counts[t]
        return dict(lp=sum(v['lp'] for v in per.values()), per=per,
                    calls={k: orc.calls[k] + nat_acc['calls'][k] for k in orc.calls},
                    expanded=orc.n_expanded + nat_acc['expanded'],
                    hist=sum(v['hist'] for v in per.values()),
                    stalled=orc.table.stalled + nat_acc['stalled'])

    import threading
    acc_lock = threading.Lock()

    # several handles: a worker that finds no root left takes pending cells from one that is in the
    # middle of a root (frontier.LocalExchange on ehm_frontier_take / _give), so ONE expensive
    # root is shared by all handles; the sub-trees are attached to the owner's tree after the step
    exchange = [None]

    def grow_group(part, nat=None, cells=None):
        nat = nat or native
        if native is None:
            return bnb_frontier.grow_frontier(orc, part, 'ecc', order=args.order,
                                              table_backoff=True, round_cap=args.round_cap,
                                              min_regions=None if args.regions else regions,
                                              max_visits=args.max_visits)
        if len(natives) == 1:
            orc.table.forget()
        st = frontier.grow_cells(nat, part, slow_oracle=lambda: orc, round_cap=args.round_cap,
                                 max_visits=args.max_visits or 0, deadline=step_deadline[0],
                                 # (a soft limit is honoured, and a request for cells answered,
                                 # at the end of a slice of visits)
                                 slice_visits=(args.steal_slice
                                               if exchange[0] is not None else
                                               40000 if len(natives) > 1 else 100000),
                                 cells=cells,
                                 between_slices=exchange[0].serve if exchange[0] is not None
                                 else None,
                                 max_depth=args.max_depth or 0,
                                 min_regions=0 if args.regions or regions >= (1 << 30) else regions,
                                 slow_opts=dict(order=args.order, table_backoff=True,
                                                round_cap=args.round_cap))
        with acc_lock:
            nat_acc['calls']['P_theta'] += st['calls_p_theta']
            nat_acc['calls']['V_R'] += st['calls_v_r']
            nat_acc['calls']['bar_E'] += st['calls_bar_e']
            nat_acc['calls']['bar_D'] += st['calls_bar_d']
            nat_acc['expanded'] += st['prefixes_expanded']
            nat_acc['stalled'] += st['stalled']
            nat_acc['solver_seconds'] += st['seconds_solvers']
            nat_acc['driver_seconds'] += st['seconds_total']
            nat_acc['open_cells'] += st['slow_path_cells']
            nat_acc['launches'] += st['launches']
        return dict(host_visits=st['visits'] + st['slow_path_visits'], rounds=st['rounds'],
                    regions=st['regions'], truncated=bool(st['truncated']), handoffs=0,
                    given=[(pc['id'], leaves) for pc, leaves in st['given_away']],
                    cells_given_away=sum(len(pc['node']) for pc, _ in st['given_away']),
                    native_visits=st['visits'], slow_path_cells=st['slow_path_cells'],
                    depth_limited_leaves=int(st.get('depth_limited_leaves', 0)),
                    depth_limited_without_commutation=int(
                        st.get('depth_limited_without_commutation', 0)))

    step_deadline = [None]

    def merge(stats, st):
        if stats is None:
            return dict(st)
        for k, v in st.items():         # counters add up, flags combine
            if isinstance(v, bool):
                stats[k] = bool(stats.get(k)) or v
            elif isinstance(v, (int, float)):
                stats[k] = stats.get(k, 0) + v
        return stats

    def step():
        step_deadline[0] = time.perf_counter() + args.seconds if args.seconds > 0 else None
        # with a target of regions the cells are grown one group after the other, EACH TO
        # COMPLETION (every leaf eps-suboptimal), until the target is reached
        group = args.cells_at_once if args.cells_at_once > 0 else (
            1 if args.regions else max(len(my_cells), 1))
        box = dict(stats=None, trees=[], next=0, error=None, given=[], adopted={})
        exchange[0] = frontier.LocalExchange(len(natives)) \
            if len(natives) > 1 and args.steal else None

        def over():
            """No more work is started: a limit of the step is reached (under no lock)."""
            if box['error'] is not None:
                return True
            if args.regions and box['stats'] is not None and \
                    box['stats'].get('regions', 0) >= regions:
                return True
            return step_deadline[0] is not None and time.perf_counter() >= step_deadline[0]

        def claim():
            """Next group of this rank's cells, or None when a limit is reached (under the lock)."""
            with acc_lock:
                if box['error'] is not None or box['next'] >= len(my_cells):
                    return None
                if args.regions and box['stats'] is not None and \
                        box['stats'].get('regions', 0) >= regions:
                    return None
                if step_deadline[0] is not None and time.perf_counter() >= step_deadline[0]:
                    return None
                g0 = box['next']
                box['next'] += group
                return g0
        # (N > 1 with --regions: the cells of a rank are the static deal k mod world; the dynamic
        # deal over ranks -- distributed.claim_roots, a counter in the process group's store -- is
        # what distributed.grow_roots_sharded(deal='dynamic') does and tests/test_distributed_gloo.py
        # covers; the handles of ONE rank claim from its list here)

        def work(nat):
            try:
                while True:
                    g0 = claim()
                    if g0 is None:
                        break
                    part = [Tree(NodeData(vertices=cell_vertices(c)))
                            for c in my_cells[g0:g0 + group]]
                    t_g = time.perf_counter()
                    st = grow_group(part, nat)
                    with acc_lock:
                        box['given'] += st['given']
                        box['trees'] += part
                        cells_log.append(dict(cells=my_cells[g0:g0 + group],
                                              seconds=time.perf_counter() - t_g,
                                              regions=int(st.get('regions', 0)),
                                              rounds=int(st['rounds']),
                                              visits=int(st['host_visits']),
                                              truncated=bool(st['truncated'])))
                        if args.progress_file and rank == 0:
                            with open(args.progress_file, 'a') as f:
                                f.write(json.dumps(cells_log[-1]) + '\n')
                        box['stats'] = merge(box['stats'], st)
                # no root left: cells of the roots the other handles are still growing
                while exchange[0] is not None:
                    parcel = exchange[0].wait_for_work(stop=over)
                    if parcel is None:
                        break
                    sub = [Tree(NodeData(vertices=R.copy())) for R in parcel['vertices']]
                    st = grow_group(sub, nat, cells=parcel)
                    with acc_lock:
                        box['adopted'][parcel['id']] = dict(trees=sub, given=st['given'])
                        st['cells_adopted'] = len(sub)
                        box['stats'] = merge(box['stats'], st)
            except BaseException as e:      # a worker's failure ends the step
                with acc_lock:
                    box['error'] = e

        if len(natives) > 1:
            threads = [threading.Thread(target=work, args=(nat,)) for nat in natives]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if box['error'] is not None:
                raise box['error']
            if box['stats'] is not None:
                waiting = []
                box['stats']['subtrees_attached'] = frontier.attach_adopted(
                    box['given'], box['adopted'], waiting)
                box['stats']['cells_taken_and_never_grown'] = len(waiting)
            return box['stats'] or dict(host_visits=0, handoffs=0, truncated=False), box['trees']
        stats, trees = None, []
        for g0 in range(0, len(my_cells), group):
            if args.regions and stats is not None and stats.get('regions', 0) >= regions:
                break
            if step_deadline[0] is not None and time.perf_counter() >= step_deadline[0]:
                break
            orc.table.forget()
            part = [Tree(NodeData(vertices=cell_vertices(c))) for c in my_cells[g0:g0 + group]]
            t_g, lp_g = time.perf_counter(), snapshot()['lp']
            st = grow_group(part)
            trees += part
            cells_log.append(dict(cells=my_cells[g0:g0 + group],
                                  seconds=time.perf_counter() - t_g,
                                  regions=int(st.get('regions', 0)), rounds=int(st['rounds']),
                                  visits=int(st['host_visits']),
                                  lp=int(snapshot()['lp'] - lp_g),
                                  truncated=bool(st['truncated'])))
            if args.progress_file and rank == 0:
                with open(args.progress_file, 'a') as f:
                    f.write(json.dumps(cells_log[-1]) + '\n')
            stats = merge(stats, st)
        return stats or dict(host_visits=0, handoffs=0, truncated=False), trees

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    failure = None
    try:
        for _ in range(args.warmup):
            step()
    except Exception as e:
        failure = e
    barrier()
    n_log_warm = len(cells_log)
    nat_acc.update(solver_seconds=0., driver_seconds=0., open_cells=0, launches=0)
    s0 = snapshot()
    t0 = time.perf_counter()
    runs = []
    try:
        if failure is None:
            runs = [step() for _ in range(args.steps)]
    except Exception as e:
        failure = e
    barrier()
    elapsed = time.perf_counter() - t0
    s1 = snapshot()
    red_dev = ('cuda:%d' % device_index) if backend == 'nccl' else 'cpu'
    _, any_failed = distributed.allreduce_counters([0. if failure is None else 1.], device=red_dev)
    if any_failed[0] > 0:
        sys.stderr.write('bench.py rank %d: %s\n' % (
            rank, failure if failure is not None else 'another rank failed'))
        orc.close()
        raise RuntimeError('config5 failed: %s' % (failure,))
    K = args.steps
    nodes = leaves = closed = depth = 0
    for _, trees in runs:
        for t in trees:
            for nd, loc in t.walk():
                nodes += 1
                if nd.is_leaf():
                    leaves += 1
                    closed += bool(nd.data.is_epsilon_suboptimal)
                    depth = max(depth, len(loc))
    calls = {k: s1['calls'][k] - s0['calls'][k] for k in s1['calls']}
    hist = s1['hist'] - s0['hist']
    # per table: LPs, iterations, kernel seconds (HIP events around every batched launch) and the
    # SURVEY 8(d) flops of its problems over a simplex, priced at the table's own dimensions
    tables = {}
    for name, (n_t, m_t, p_t) in table_dims.items():
        a0, a1 = s0['per'][name], s1['per'][name]
        d_lp, d_it = a1['lp'] - a0['lp'], a1['iters'] - a0['iters']
        mean_it_t = d_it / max(d_lp, 1.)
        h = a1['hist'] - a0['hist']
        dims_t = {2: (n_t + p_t + 1, m_t + p_t + 2), 3: (n_t + p_t, m_t + p_t + 1),
                  4: (n_t + p_t + 1, m_t + p_t + 3)}
        tables[name] = dict(
            lp=float(d_lp), iters=float(d_it), mean_it=mean_it_t, dims=dims_t,
            point_s=a1['secs'][0] - a0['secs'][0], simplex_s=a1['secs'][1] - a0['secs'][1],
            simplex_launches=float(a1['launches'][1] - a0['launches'][1]),
            n_simplex=float(sum(h[k].sum() for k in dims_t)),
            flops=sum(float(h[k].sum()) * mean_it_t * flops_per_iteration(*dims_t[k])
                      for k in dims_t))
    dom = max(tables, key=lambda k: tables[k]['simplex_s'])
    # the dominant table runs on the wide kernels unless its blocks fit the shared-block ones
    # (columns n + p + 1 <= 32 and rows m + p + 3 <= 256: csrc/ehm_capi.hip)
    dom_wide = table_dims[dom][0] + table_dims[dom][2] + 1 > 32 or \
        table_dims[dom][1] + table_dims[dom][2] + 3 > 256
    local = [float(s1['lp'] - s0['lp']), float(sum(t['iters'] for t in tables.values())),
             float(nodes), float(closed), float(sum(calls.values())), elapsed,
             sum(t['point_s'] for t in tables.values()),
             sum(t['simplex_s'] for t in tables.values()),
             tables[dom]['simplex_launches'],
             float(sum(st['host_visits'] for st, _ in runs))]
    tot, mx = distributed.allreduce_counters(local, device=red_dev)
    out = None
    if rank == 0:
        can = orc.table.gp.can
        n, m, p = can.n, can.m, can.p
        lp, iters, nodes_t, closed_t, micp, elapsed_max = (tot[0], tot[1], tot[2], tot[3], tot[4],
                                                          float(mx[5]))
        point_s, simplex_s = float(mx[6]), float(mx[7])
        mean_it = iters / max(lp, 1.)
        # dominant kernel: the batched problems over a simplex of the table that spends the most
        # kernel time -- the wide kernels (k3_simplex_batch) for the full model, the shared-block
        # kernels (k2_simplex_batch) for the short-horizon table; iterations by kind are not
        # counted separately: every kind is priced at the mean over the table's LPs
        T = tables[dom]
        dims = T['dims']
        flops, n_sx = T['flops'], T['n_simplex']
        hbm_alg = n_sx * (12 + 8 * ((p + 1) * p + (p + 1)) + 8 + 8 * (p + 1) + 4)
        dom_s = T['simplex_s']
        achieved = flops / max(dom_s, 1e-12) / 1e12
        c5_traffic, c5_traffic_src = pmc_traffic(
            wide_kernel('simplex_batch') if dom_wide else 'k2_simplex_batch',
            'pmc_summary_config5.json')
        out = {
            'metric': 'oracle LP solves/sec + final regions/sec, 4-state 2-input N=5 hybrid MPC',
            'value': lp / elapsed_max, 'unit': 'LP solves/s',
            'regions_per_s': closed_t / elapsed_max,
            'oracle_calls_answered_per_s': micp / elapsed_max,
            'n_gpus': min(world, max(torch.cuda.device_count(), 1)), 'host_processes': world,
            'steps': K, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed_max / K, 'ms_export': None,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': 'configs[4] (NOT the headline configuration): n_x=8 n_u=3 N=8 p=8, 4 '
                            'modes = %d mode sequences (searched by branch-and-bound over mode '
                            'prefixes, never enumerated), inf-norm LP oracle (relaxation blocks '
                            'n=%d m=%d), seed %d, eps_r=%g, eps_a=%.6g (abs_frac=%g), %s, %s' % (
                                mpc.delta_size ** mpc.N, n, m, args.seed, eps_r, eps_a, abs_frac,
                                ('the first %d of the %d Delaunay root simplices of the box '
                                 '(tools.delaunay_roots, Qhull order: the reference\'s roots, '
                                 'lib/tools.py:152-189)' % (n_cells, n_roots_total))
                                if args.roots == 'delaunay' else
                                '%d Kuhn cell(s) of the box' % n_cells,
                                ('cell after cell, each to completion, until %d regions are closed '
                                 'per rank' % regions) if args.regions else
                                'grown to completion'),
                'cells_grown_per_step': sum(len(c['cells']) for c in cells_log[n_log_warm:]) / K,
                'cells_log': cells_log[n_log_warm:],
                'regions_per_step': closed_t / K, 'nodes_per_step': nodes_t / K,
                'open_leaves_per_step': (leaves - closed) / K if world == 1 else None,
                'depth_limit': args.max_depth or None,
                # open leaves the depth limit left unbisected (flag EHM_FR_DEPTH; the rest of the
                # open leaves were pending when a time / region limit cut the step)
                'depth_limited_leaves_per_step':
                    sum(st.get('depth_limited_leaves', 0) for st, _ in runs) / K,
                'depth_limited_leaves_without_a_commutation_per_step':
                    sum(st.get('depth_limited_without_commutation', 0) for st, _ in runs) / K,
                'tree_depth': depth,
                'node_visits_per_step': tot[9] / K,
                'lp_solves_per_step': lp / K,
                'mixed_integer_oracle_calls_per_step': {k: v / K for k, v in calls.items()},
                'lp_solves_per_mixed_integer_oracle_call': lp / max(micp, 1.),
                'reference_equivalent_solves_per_step': micp / K,
                'mean_ipm_iterations': mean_it,
                'lp_solves_by_kind_and_prefix_length_per_step': {
                    kind: (row / float(K)).tolist() for kind, row in zip(
                        ('point_phase_one', 'point', 'simplex_phase_one', 'min_over_simplex',
                         'slack'), hist)},
                'prefixes_expanded_per_step': (s1['expanded'] - s0['expanded']) / K,
                'stalled_device_solves': s1['stalled'] - s0['stalled'],
                'handoffs_to_the_enumerating_engine_per_step':
                    sum(st['handoffs'] for st, _ in runs) / K,
                'eps_a_seconds': t_eps,
                'kernels': ('prefixes of <= %d steps: blocks of the law with that horizon on the '
                            'shared-block kernels (one wavefront per LP); longer prefixes and '
                            'full sequences: ' % orc.table.short_len
                            if hasattr(orc.table, 'short') else '') +
                           'wide kernels (one workgroup per LP, MFMA normal matrix); both through '
                           'the batched oracles ehm_simplex_idx_batch / ehm_point_idx_batch',
                'engine': ('native driver (include/ehm_frontier.h, csrc/ehm_frontier.cpp): round '
                           'loop, searches, block condensation and launches in C++ behind one '
                           'call per group of cells; cells bar_E leaves open go back to '
                           'bnb_frontier (%d of %d visits)' % (
                               nat_acc['open_cells'], int(tot[9])) if native else
                           'host-driven searches (bnb_frontier.grow_frontier: all pending nodes '
                           'share the launches; native memo of phase-one verdicts, '
                           'csrc/ehm_search.cpp)') + ', LPs on the device',
                'native_driver': None if not native else {
                    'seconds_inside_ehm_frontier_run_per_step': nat_acc['driver_seconds'] / K,
                    'of_them_inside_the_batched_solver_calls': nat_acc['solver_seconds'] / K,
                    'solver_calls_per_step': nat_acc['launches'] / K,
                    'cells_handed_back_open_per_step': nat_acc['open_cells'] / K},
                'order': {'lcss-first': 'cells that hold a commutation first, deepest first',
                          'fifo': 'level by level', 'deepest': 'deepest first'}[args.order] +
                         '; rounds of %d nodes' % args.round_cap,
                'stopped_early': bool(any(st['truncated'] for st, _ in runs)),
                'parallelism': '%d cell(s) over %d host process(es) on %d GPU(s): cell k on rank k '
                               'mod world, %s, no data-path collective' % (
                                   n_cells, world, min(world, max(torch.cuda.device_count(), 1)),
                                   'all of a rank\'s cells together' if args.cells_at_once <= 0
                                   else '%d at a time' % args.cells_at_once) + (
                    '; %d driver handles per process grow different cells at the same time '
                    '(--host-streams)' % len(natives) if len(natives) > 1 else ''),
                'host_streams': max(1, len(natives)),
                'cells_moved_between_handles_per_step': sum(
                    st.get('cells_given_away', 0) for st, _ in runs) / K,
                'subtrees_attached_per_step': sum(
                    st.get('subtrees_attached', 0) for st, _ in runs) / K,
                'cells_taken_and_never_grown_per_step': sum(
                    st.get('cells_taken_and_never_grown', 0) for st, _ in runs) / K,
            },
            'roofline': {
                'bound': 'mfma' if dom_wide else 'valu-fp64',
                'kernel': wide_kernel('simplex_batch') if dom_wide else 'k2_simplex_batch',
                'note': 'LPs over a simplex of the %s table (slack n=%d m=%d) on the %s; flops = '
                        'its LPs by kind x the MEAN iteration count of its LPs x SURVEY 8(d) '
                        'flops per iteration; kernel seconds by HIP events around every launch '
                        '(ehm_counters.batch_seconds)' % (
                            dom, dims[4][0], dims[4][1],
                            'wide kernels (normal matrix on v_mfma_f64_16x16x4_f64)'
                            if dom_wide else
                            'shared-block kernels (one wavefront per LP, FP64 vector FMA)'),
                'tables': {name: {'lp_solves': t['lp'] / K, 'mean_ipm_iterations': t['mean_it'],
                                  'simplex_kernel_seconds': t['simplex_s'],
                                  'point_kernel_seconds': t['point_s'],
                                  'slack_lp': '%d x %d' % t['dims'][4],
                                  'simplex_tflops': t['flops'] / max(t['simplex_s'], 1e-12) / 1e12}
                           for name, t in tables.items()},
                'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / FP64_PEAK_TFLOPS,
                'flop_per_ipm_iteration': flops_per_iteration(*dims[4]),
                'traffic': c5_traffic, 'traffic_source': c5_traffic_src,
                'traffic_unit': 'HBM bytes per launch of the kernel, mean over the launches of '
                                'one cell (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)',
                'algorithmic_bytes_per_launch': hbm_alg / max(tot[8], 1.),
                'kernel_seconds': dom_s, 'launches': tot[8],
                'point_kernel_seconds': point_s, 'simplex_kernel_seconds_all_tables': simplex_s,
                'device_share_of_the_step': (simplex_s + point_s) / elapsed_max,
                'device_share_note': 'kernel seconds summed over the tables; the native driver '
                                     'runs the tables of one solver call on streams of their own, '
                                     'so concurrent kernels count twice and the share can exceed 1',
                'hbm': {'achieved': hbm_alg / max(dom_s, 1e-12) / 1e9, 'peak': HBM_PEAK_GBS,
                        'unit': 'GB/s',
                        'frac': hbm_alg / max(dom_s, 1e-12) / 1e9 / HBM_PEAK_GBS},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_config5(args.seed, eps_a, eps_r, args.cpu_seconds,
                                                       cell_vertices)
        else:
            out['cpu_baseline'] = None
    for nat in natives:
        nat.close()
    orc.close()
    return out


# what the default invocation measures after the headline: (workload, steps, warmup)
# ---- f2: batched evaluation of the explicit law ---------------------------
def _explicit_cpu_worker(job):
    """cpu_baseline leg of --workload explicit (test infrastructure: oracle/explicit_cpu.py)."""
    path, seconds, seed = job
    from oracle.explicit_cpu import ExplicitFlatCPU
    d = np.load(path)
    cpu = ExplicitFlatCPU(d['vertices'], d['vinput'], d['left'], d['right'], int(d['n_roots']))
    X = d['X']
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        cpu(X[rng.integers(X.shape[0])])
        n += 1
    return n, time.perf_counter() - t0


def measure_explicit(args, ctx):
    """
    Evaluations per second of the explicit law (csrc/ehm_explicit.hip; the reference's
    ExplicitMPC.__call__, lib/mpc_library.py:737-792, whose evaluation time is the headline of the
    reference's README): the partition of the headline workload (1.6 M nodes, 22 roots, p = 4),
    and the root location over the 34 871 Delaunay roots of a p = 8 box (the spine of configs[4]).
    """
    import tempfile
    from explicit_hybrid_mpc_amd import engine, examples, explicit
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    device_index = ctx['device_index']
    out_cases = []
    n_q = args.queries
    rng = np.random.default_rng(args.seed)

    def run_case(name, flat, half, note):
        ex = explicit.ExplicitMPC(flat, device=device_index)
        X = rng.uniform(-1, 1, (n_q, half.size)) * half
        ex.evaluate(X[:1024])                                   # warm-up
        secs, walls, visited = [], [], None
        for _ in range(max(1, args.steps)):
            t0 = time.perf_counter()
            u, leaf, visited, ks = ex.evaluate(X, return_info=True)
            walls.append(time.perf_counter() - t0)
            secs.append(ks)
        ks = float(np.mean(secs))
        p, n_u = ex.p, ex.n_u
        rec_bytes = 8 * (((p + p * p + 7) // 8) * 8)
        # algorithmic bytes: one record per containment test, the child pair per level, the leaf's
        # record + vertex inputs, the query in and the input out
        tests = float(np.sum(visited))
        alg = tests * (rec_bytes + 8) + n_q * (rec_bytes + 8 * (p + 1) * n_u + 8 * p + 8 * n_u)
        case = {
            'name': name, 'note': note, 'queries': n_q, 'nodes': int(flat.n_nodes),
            'roots': int(flat.info['n_roots']), 'p': p, 'n_u': n_u,
            'evaluations_per_s': n_q / ks, 'kernel_ms': ks * 1e3,
            'evaluations_per_s_with_host_copies': n_q / float(np.mean(walls)),
            'containment_tests_per_query': tests / n_q,
            'roofline': {'bound': 'hbm', 'kernel': 'k_explicit_eval' +
                         (' + k_explicit_locate' if flat.info['n_roots'] >= 128 else ''),
                         'achieved': alg / ks / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': alg / ks / 8e12, 'traffic': None,
                         'algorithmic_bytes_per_launch': alg,
                         'note': 'a walk reads one 64-byte-aligned record per test: latency-bound '
                                 'pointer chasing through L2 / MALL, not a stream'}}
        if not args.no_cpu_baseline:
            import multiprocessing as mp
            cores = usable_cores()
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, 'tree.npz')
                np.savez(path, vertices=flat.vertices, vinput=flat.vertex_inputs, left=flat.left,
                         right=flat.right, n_roots=flat.info['n_roots'], X=X[:4096])
                with mp.get_context('spawn').Pool(cores) as pool:
                    res = pool.map(_explicit_cpu_worker,
                                   [(path, args.cpu_seconds, args.seed + k) for k in range(cores)])
            case['cpu_baseline'] = {
                'value': sum(r[0] for r in res) / max(r[1] for r in res), 'unit': 'evaluations/s',
                'cores': cores, 'kind': 'port',
                'sample': '%d processes, each the walk of lib/mpc_library.py:737-792 restated on '
                          'the flat arrays (oracle/explicit_cpu.ExplicitFlatCPU, numpy) for %.0f s '
                          'over queries drawn from the same batch' % (cores, args.cpu_seconds)}
        ex.close()
        return case

    # (a) the headline partition
    mpc = make_mpc('config2', args.seed)
    gp = engine.GpuProblem(mpc.compile(), 1., 1., device=device_index)
    half = examples.theta_box(mpc)
    V = examples.box_vertices(half)
    J_abs, _, _ = gp.solve_pt(0.02 * V)
    gp.set_eps(float(np.max(J_abs)), 1e-2)
    roots, _ = ehm_tools.delaunay_roots(V)
    flat = gp.partition(roots, action='ecc', export=True, with_volume=False)
    gp.close()
    out_cases.append(run_case('headline_tree', flat, half,
                              'partition of configs[1] (eps_r 0.01, abs_frac 0.02), uniform states'))
    # (b) root location over the spine of a p = 8 box (the roots of configs[4], none grown: what is
    # measured is the walk over the right spine, lib/mpc_library.py:760-766)
    mpc8 = examples.pwa4_mpc(N=CONFIG5['N'], seed=args.seed)
    half8 = examples.theta_box(mpc8)
    roots8, _ = ehm_tools.delaunay_roots(examples.box_vertices(half8))
    roots8 = np.asarray(roots8, dtype=np.float64)
    K, p8 = roots8.shape[0], roots8.shape[2]
    n_u8 = mpc8.n_u if hasattr(mpc8, 'n_u') else 3
    leafs = -np.ones(K, dtype=np.int32)
    flat8 = engine.FlatTree(roots8, leafs, leafs.copy(), np.zeros(K, dtype=np.int32),
                            np.zeros((K, p8 + 1)), np.zeros((K, p8 + 1, n_u8)),
                            np.zeros(K, dtype=np.uint8), np.zeros(K), {'n_roots': K}, None)
    out_cases.append(run_case('p8_spine', flat8, half8,
                              'the %d Delaunay roots of the p = 8 box of configs[4] as leaves: root '
                              'location only (one wavefront per query, 64 roots per step)' % K))
    head = out_cases[0]
    return {
        'metric': 'explicit-law evaluations/s (SURVEY 8 f2; NOT the headline metric of BASELINE.json)',
        'value': head['evaluations_per_s'], 'unit': 'evaluations/s', 'n_gpus': 1,
        'steps': max(1, args.steps), 'warmup': 1, 'ms_per_step': head['kernel_ms'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': 'explicit: %d uniform states per step through the partition of '
                               'configs[1] (%d nodes)' % (n_q, head['nodes'])},
        'roofline': head['roofline'], 'cpu_baseline': head.get('cpu_baseline'),
        'cases': out_cases}


SECONDARY = (('config3', 2, 1), ('config4', 2, 1), ('config2q', 5, 2), ('config5', 1, 0),
             ('config5_scale', 1, 0))
SECONDARY_KEYS = ('value', 'unit', 'ms_per_step', 'regions_per_s', 'oracle_calls_answered_per_s',
                  'steps', 'warmup', 'roofline', 'cpu_baseline')
SECONDARY_CONFIG_KEYS = ('workload', 'regions_per_step', 'nodes_per_step', 'lp_solves_per_step',
                         'open_leaves_at_max_depth_per_step', 'open_leaves_per_step',
                         'depth_limit', 'depth_limited_leaves_per_step', 'cells_grown_per_step',
                         'depth_limited_leaves_without_a_commutation_per_step', 'cells_log',
                         'host_streams', 'parallelism', 'cells_moved_between_handles_per_step',
                         'subtrees_attached_per_step', 'cells_taken_and_never_grown_per_step',
                         'tree_depth', 'mean_ipm_iterations',
                         'midpoint_optima_taken_from_the_table_per_step',
                         'lp_solves_per_mixed_integer_oracle_call',
                         'mixed_integer_oracle_calls_per_step', 'node_visits_per_step', 'engine',
                         'kernels')


def secondary_line(args, ctx, workload, steps, warmup):
    """One of the other BASELINE.json configurations, for a few steps, as an entry of the
    headline line's "secondary" list."""
    import copy
    a = copy.copy(args)
    a.workload, a.steps, a.warmup = workload, steps, warmup
    a.abs_frac = a.eps_r = a.max_depth = None
    a.cpu_seconds = args.secondary_cpu_seconds
    a.regions = a.cells = 0
    a.progress_file = None
    a.seconds = 0.
    if workload == 'config5_scale':
        # configs[4] towards its stated size: the Delaunay roots in Qhull order, each to completion
        # under a depth limit, until 1e6 regions are closed or the soft time limit cuts the root in
        # progress (its pending cells stay open leaves and are reported as such)
        if args.scale_seconds <= 0:
            return {'name': workload, 'skipped': '--scale-seconds 0'}
        a.workload, a.regions, a.cells = 'config5', 10 ** 6, 400
        a.seconds, a.max_depth = float(args.scale_seconds), 26
        # (measured, 480-s box: 1 / 2 / 3 / 4 / 6 handles -> 462 / 611 / 675 / 703 / 721 k LP/s,
        # 446 / 578 / 649 / 724 / 850 k regions: profiles/r6/c5_scale_host_streams.txt)
        a.host_streams = max(6, args.host_streams)
    a.driver = 'native'
    a.order, a.max_visits, a.round_cap = 'lcss-first', None, 4096
    a.status_dir = None
    a.engine, a.solver, a.decide_full = 1, 2, False
    a.no_mid_first = a.no_inherit_witness = False
    t0 = time.perf_counter()
    try:
        full = (measure_config5 if a.workload == 'config5' else measure)(a, ctx)
    except (Exception, SystemExit) as e:       # a failing secondary workload must not cost the headline
        return {'workload': workload, 'error': '%s: %s' % (type(e).__name__, e)}
    line = {k: full.get(k) for k in SECONDARY_KEYS}
    line['name'] = workload
    if workload == 'config5_scale':
        line['limits'] = {'soft_seconds': a.seconds, 'regions_target': a.regions,
                          'max_depth': a.max_depth, 'roots_offered': a.cells,
                          'host_streams': a.host_streams}
    line['workload'] = full['config']['workload']
    line['config'] = {k: full['config'][k] for k in SECONDARY_CONFIG_KEYS if k in full['config']}
    line['wall_seconds'] = time.perf_counter() - t0
    return line


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import distributed
    # "nccl" is RCCL on ROCm; EHM_BENCH_BACKEND=gloo lets several ranks share one GPU (tests)
    backend = os.environ.get('EHM_BENCH_BACKEND', 'nccl')
    rank, local_rank, world = distributed.init_process_group(backend)
    n_dev = max(torch.cuda.device_count(), 1)
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    if world != args.gpus and rank == 0:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE\n' %
                         (args.gpus, world))
    ctx = dict(backend=backend, rank=rank, world=world, device_index=device_index)
    rc = 0
    try:
        out = (measure_config5 if args.workload == 'config5' else
               measure_explicit if args.workload == 'explicit' else measure)(args, ctx)
        if (rank == 0 and world == 1 and args.workload == 'config2' and not args.no_secondary
                and not args.status_dir):
            out['secondary'] = [secondary_line(args, ctx, *w) for w in SECONDARY]
        if rank == 0:
            print(json.dumps(out))
    except SystemExit as e:
        rc = e.code or 1
    if world > 1:
        if rc == 0:
            dist.barrier()
        dist.destroy_process_group()
    if rc:
        raise SystemExit(rc)


if __name__ == '__main__':
    main()
