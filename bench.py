#!/usr/bin/env python
"""
Benchmark of the partitioning hot path on MI355X.

A "step" is one complete partition of the synthetic config-2 instance
(n_x=4, n_u=2, N=5, p=4 linear MPC, infinity-norm LP cost, eps_r=1e-2; BASELINE.json
configs[1]): feasible-commutation pass over the 22 Delaunay root simplices, then
epsilon-suboptimal refinement until every leaf is closed.  Problem constants and the
root simplices are resident in HBM before the timed region; the timed region contains
everything else (all frontier sweeps, LP solves, child construction).

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line on rank 0.  For N > 1 it is launched by torch.distributed.run with
one rank per GPU; every rank grows the same top of the tree until the frontier holds 1024
nodes per rank, keeps the positions k % N == rank, and grows its share in one launch of the
persistent frontier kernel -- no collective in the data path (`--balance static`, the default;
`--balance dynamic` sweeps level by level and rebalances the frontiers with an all-gather of
their sizes and point-to-point node transfers, explicit_hybrid_mpc_amd/distributed.py).  The
total work is fixed ("strong" scaling) and `value` is the whole-job LP-solve rate.
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

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = FP32 vector / 2 (157.3 TF, MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def flops_per_iteration(n, m):
    """dense primal-dual IPM iteration on  min c^T z, G z <= h."""
    return 2. * m * n * n + n ** 3 / 3. + 8. * m * n + 4. * n * n


def node_bytes(p, n_u, delta_len):
    """node payload + 16 B topology/flags."""
    return 8 * ((p + 1) * p + (p + 1) + (p + 1) * n_u) + delta_len + 16


def pmc_traffic(kernel, summary='pmc_summary_bench.json'):
    """
    HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of THIS workload
    (profiles/r1/pmc_summary_bench.json, tools/profile.sh): FETCH_SIZE / WRITE_SIZE are in KiB
    and come from separate passes; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64
    bytes, so it is doubled (MI355X_MICROARCH.md, HBM section).  None if no profile is there.
    """
    path = os.path.join(ROOT, 'profiles', 'r1', summary)
    try:
        c = json.load(open(path))['counters'][kernel]
        n_f, n_w = c['_dispatches_pmc3'], c['_dispatches_pmc4']
        return (2. * c['FETCH_SIZE'] / n_f + c['WRITE_SIZE'] / n_w) * 1024.
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def cpu_baseline(mpc, eps_a, eps_r, seconds):
    """Oracle (CPU restatement, HiGHS) timed on a bounded prefix of the same partition."""
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    from oracle import geometry
    from explicit_hybrid_mpc_amd import examples
    V = examples.box_vertices(examples.theta_box(mpc))
    roots, locs = geometry.delaunay_simplices(V)
    orc = OracleCPU(mpc, eps_a, eps_r)
    part = PartitionCPU(orc, max_nodes=50)
    t0 = time.perf_counter()
    visits = 0
    while time.perf_counter() - t0 < seconds:
        # extend the bounded prefix in chunks of node visits
        part.max_nodes = part.visits + 50
        if part.visits == 0:
            part.run(roots, locs, 'ecc')
        else:
            part.resume()
        visits = part.visits
        if not part.truncated:
            break
    dt = time.perf_counter() - t0
    return dict(value=orc.n_solves / dt, unit='LP solves/s', cores=1, kind='port',
                sample='first %d node visits of the same partition (%d HiGHS LP solves, '
                       '%.1f s), oracle/partition_cpu.py' % (visits, orc.n_solves, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--workload', choices=['config2', 'config4', 'config2q'], default='config2',
                    help='config2 = the BASELINE.json metric (default); config4 = n_x=6 n_u=3 N=10 '
                         'box-constrained instance on the wide kernels (a parity-test '
                         'configuration, timed for the record); config2q = config 2 with the '
                         'quadratic cost of the same weights (convex QP / QCQP oracles, the '
                         "reference's cvx.quad_form cost class; not the headline metric)")
    ap.add_argument('--abs-frac', type=float, default=None,
                    help='eps_a rule of lib/examples.py:42-46 (default 0.02; config4: 0.4)')
    ap.add_argument('--eps-r', type=float, default=None, help='default 1e-2; config4: 0.25')
    ap.add_argument('--max-nodes', type=int, default=1 << 22)
    ap.add_argument('--shard-min-frontier', type=int, default=0,
                    help='frontier size at which it is dealt over the ranks (0 = 64 per rank)')
    ap.add_argument('--balance', choices=['static', 'dynamic'], default='static',
                    help='N > 1: static = the frontier is dealt round-robin once it holds 1024 '
                         'nodes per rank and every rank grows its share in ONE launch of the '
                         'persistent frontier kernel, no collective in the data path (measured '
                         'load imbalance 1.04 at 8 shards); dynamic = level-synchronous sweeps '
                         'with an all-gather of the frontier sizes and point-to-point node '
                         'transfers every --sweeps-per-round sweeps')
    ap.add_argument('--sweeps-per-round', type=int, default=2,
                    help='frontier sweeps between two rebalancing rounds (N > 1)')
    ap.add_argument('--status-dir', default=None,
                    help='write status.txt / statistics.pkl (reference formats) there; adds one '
                         'progress read-back (and, N > 1, one all-gather) per round -- off by '
                         'default, the headline number is measured without it')
    ap.add_argument('--cpu-seconds', type=float, default=15.)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--solver', type=int, default=2, help='kernel generation (1 or 2)')
    ap.add_argument('--no-mid-first', action='store_true',
                    help='persistent kernel WITHOUT the midpoint solve before the suboptimality '
                         'test (the round-1 flow; A/B measurements)')
    ap.add_argument('--engine', type=int, default=1,
                    help='1 = persistent frontier kernel (one launch per partition; single rank, '
                         'shared-block kernels), 0 = level-synchronous sweeps')
    ap.add_argument('--decide-full', action='store_true',
                    help='solve the suboptimality-test LPs to full accuracy (no sign-only stop)')
    args = ap.parse_args()

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
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools

    wide = args.workload == 'config4'
    if args.abs_frac is None:
        args.abs_frac = 0.4 if wide else 0.02
    if args.eps_r is None:
        args.eps_r = 0.25 if wide else 1e-2
    quad = args.workload == 'config2q'
    mpc = examples.integrator_chain_mpc() if wide else \
        examples.linear_mpc(seed=args.seed, cost='quadratic' if quad else 'inf')
    can = mpc.compile()
    gp = engine.GpuProblem(can, 1., 1., device=device_index)
    if not wide:
        gp.set_solver(args.solver)
    gp.set_option('decide_full', 1 if args.decide_full else 0)
    if args.no_mid_first:
        gp.set_option('mid_first', 0)
    static = args.balance == 'static' and not args.status_dir
    persistent = (args.engine == 1 and (world == 1 or static) and args.solver == 2 and
                  not wide and not args.status_dir)
    # (the persistent kernel exists at one solver width, k2_persist, and -- where a pair of
    # instances is compiled, as for this workload -- at two, kp_persist; the library picks)
    kname = 'k3_lcss_decide' if wide else (('kp_persist' if not quad else 'k2_persist')
                                           if persistent else
                                           'k2_lcss_decide' if args.solver == 2
                                           else 'k_lcss_decide')
    pmc_file = 'pmc_summary_wide.json' if wide else 'pmc_summary_bench.json'
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
                                shard=shard, with_volume=False, engine=args.engine)
        info, log, rounds = distributed.run_balanced(
            gp, roots, action='ecc', max_nodes=args.max_nodes,
            min_frontier=args.shard_min_frontier, sweeps_per_round=args.sweeps_per_round,
            device=xdev, export=False, status=publisher,
            publish_status=bool(args.status_dir))
        info['rounds'] = rounds
        info['moved'] = sum(len(e['ids']) for e in log if e['kind'] == 'give')
        return info

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    infos = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    # totals over ranks (max time, summed work)
    keys = ['lp_solves', 'ipm_iters', 'n_nodes', 'n_closed', 'ref_solves', 'decide_solves',
            'decide_iters', 'cert_closed', 'witness_open']
    if rank > 0:
        # the top of the tree is grown identically on every rank: count it once (rank 0)
        for i in infos:
            i['lp_solves'] -= i['replicated_solves']
            i['n_closed'] -= i['replicated_closed']
            i['n_nodes'] -= i['replicated_nodes']
    local = [float(sum(i[k] for i in infos)) for k in keys] + \
        [elapsed, sum(i['decide_seconds'] for i in infos),
         sum(i['expand_seconds'] for i in infos),
         float(sum(i.get('moved', 0) for i in infos)),
         float(sum(i.get('rounds', 0) for i in infos))]
    red_dev = ('cuda:%d' % device_index) if backend == 'nccl' else 'cpu'
    tot, mx = distributed.allreduce_counters(local, device=red_dev)
    per_rank = distributed.allgather_counts([int(local[0])], device=red_dev)[:, 0]
    elapsed_max = float(mx[len(keys)])
    agg = dict(zip(keys, tot[:len(keys)]))
    decide_s = float(mx[len(keys) + 1])
    expand_s = float(mx[len(keys) + 2])
    if rank == 0:
        K = args.steps
        info0 = infos[-1]
        n_slack, m_slack = can.n + can.p + 1, can.m + can.p + 3
        n_pt, m_pt = can.n, can.m
        decide_flops = agg['decide_iters'] * flops_per_iteration(n_slack, m_slack)
        expand_flops = (agg['ipm_iters'] - agg['decide_iters']) * flops_per_iteration(n_pt, m_pt)
        B = node_bytes(can.p, can.n_u, can.deltas.shape[1])
        if agg['cert_closed'] > 0:
            # the run keeps the vertex gradients of the optimal cost next to every record
            B += 8 * (can.p + 1) * can.p
        closed, nodes = agg['n_closed'], agg['n_nodes']
        splits = (nodes - K * len(roots)) / 2.
        if persistent:
            # ONE kernel per partition: suboptimality tests AND splits / midpoint solves
            achieved = (decide_flops + expand_flops) / decide_s / 1e12
            hbm_alg = (agg['decide_solves'] + agg['cert_closed']) * (B + 8) + \
                splits * (B + 2 * B + 16)
        else:
            # dominant kernel = the suboptimality-test sweep (k_lcss_decide)
            achieved = decide_flops / decide_s / 1e12
            hbm_alg = agg['decide_solves'] * (B + 8)      # decide: read record, write verdict
        out = {
            'metric': 'oracle LP solves/sec + final regions/sec, 4-state 2-input N=5 hybrid MPC',
            'value': agg['lp_solves'] / elapsed_max,
            'unit': 'LP solves/s',
            'regions_per_s': closed / elapsed_max,
            'n_gpus': world, 'steps': K, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed_max / K,
            'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': ('configs[3] (NOT the headline configuration): n_x=6 n_u=3 N=10 p=6 '
                             'box-constrained chain' if wide else
                             'configs[1]: n_x=4 n_u=2 N=5 p=4 linear MPC') +
                            ', %s (n=%d m=%d), seed %d, eps_r=%g, eps_a=%.6g '
                            '(abs_frac=%g), %d Delaunay roots' % (
                                'quadratic-cost QP / QCQP oracle (NOT the headline metric)'
                                if quad else 'inf-norm LP oracle',
                                can.n, can.m, args.seed, args.eps_r, eps_a, args.abs_frac,
                                len(roots)),
                'regions_per_step': closed / K,
                'nodes_per_step': nodes / K,
                'lp_solves_per_step': agg['lp_solves'] / K,
                'reference_equivalent_solves_per_step': agg['ref_solves'] / K,
                'reference_equivalent_solves_per_s': agg['ref_solves'] / elapsed_max,
                'leaves_closed_without_lp_per_step': agg['cert_closed'] / K,
                'nodes_proved_open_by_midpoint_per_step': agg['witness_open'] / K,
                'mean_ipm_iterations': agg['ipm_iters'] / max(agg['lp_solves'], 1),
                'sweeps': info0['sweeps'], 'tree_depth': info0['max_depth'],
                'min_decision_margin': info0['min_margin'],
                'kernels': 'wide (one workgroup per LP, MFMA normal matrix)' if wide else
                           'generation %d%s' % (args.solver, ' with the quadratic block' if quad
                                                else ''),
                'engine': 'persistent frontier kernel (one launch per partition)' if persistent
                          else 'level-synchronous sweeps',
                'suboptimality_test': 'full accuracy' if (args.decide_full or args.solver == 1) else
                                      'sign-only stop (lower bound of |t*| recorded)',
                'parallelism': 'frontier dealt round-robin over %d GPU(s)' % world +
                               ('' if world == 1 else
                                ' at %d nodes, every rank grows its share in one persistent '
                                'launch (static; no data-path collective)' %
                                args.shard_min_frontier if static else
                                ', rebalanced every %d sweeps (all-gather of frontier sizes + '
                                'point-to-point node records)' % args.sweeps_per_round),
                'rebalance_rounds_per_step': float(mx[len(keys) + 4]) / K,
                'nodes_moved_per_step': float(tot[len(keys) + 3]) / K,
                'lp_solves_per_rank': [int(v) for v in per_rank],
                'load_imbalance_max_over_mean': distributed.imbalance(per_rank),
            },
            'roofline': {
                'bound': 'mfma', 'kernel': kname,
                'note': ('normal matrix on v_mfma_f64_16x16x4_f64 (57 columns = 4 tiles); peak = '
                         'FP64 matrix = vector peak of MI355X' if wide else
                         'one launch holds the slack LPs (n=%d m=%d) and the midpoint LPs (n=%d '
                         'm=%d); FP64 vector FMA bound; peak = FP64 vector = matrix peak of MI355X'
                         % (n_slack, m_slack, n_pt, m_pt) if persistent else
                         'FP64 vector FMA bound (no f64 contraction >= 32 wide at n=25); peak = '
                         'FP64 vector = matrix peak of MI355X'),
                'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / FP64_PEAK_TFLOPS, 'traffic': pmc_traffic(kname, pmc_file),
                'traffic_unit': 'HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, '
                                'profiles/r1/%s)' % pmc_file,
                'algorithmic_bytes_per_launch': hbm_alg / max(info0['decide_launches'] * K, 1),
                'flop_per_ipm_iteration': flops_per_iteration(n_slack, m_slack),
                'kernel_seconds': decide_s, 'launches': info0['decide_launches'] * K,
                'expand_kernel': None if persistent else
                                 {'achieved': expand_flops / max(expand_s, 1e-12) / 1e12,
                                  'kernel_seconds': expand_s},
                'hbm': {'achieved': hbm_alg / decide_s / 1e9, 'peak': HBM_PEAK_GBS,
                        'unit': 'GB/s', 'frac': hbm_alg / decide_s / 1e9 / HBM_PEAK_GBS,
                        'bytes_per_node': B + 8},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(mpc, eps_a, args.eps_r, args.cpu_seconds)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    gp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
