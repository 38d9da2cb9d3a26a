"""Static dealing over `world` ranks, measured on ONE GPU by running the shards one after the
other: per-rank work and time, speed-up = full run / slowest shard.
  mode "deal": one persistent launch per rank from the roots, dealt at a tree depth by path code
  (ehm_run_opts.deal_depth -- what bench.py --gpus N runs);
  mode "sweeps": the round-1 scheme (sweeps until the frontier holds min_frontier nodes, deal by
  position, then one persistent launch per share).
usage: shard_balance.py [deal|sweeps] [abs_frac] [config2|config4] [log2 of the node pool, 24]
  config4 = BASELINE configs[3] (n_x = 6, N = 10: the wide kernels, level-synchronous sweeps, 652
  roots, eps_r 0.25, abs_frac 0.4 as bench.py --workload config4): only the "sweeps" dealing
  exists there."""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from explicit_hybrid_mpc_amd import engine, examples, distributed
from explicit_hybrid_mpc_amd import tools as ehm_tools

mode = sys.argv[1] if len(sys.argv) > 1 else 'deal'
workload = sys.argv[3] if len(sys.argv) > 3 else 'config2'
wide = workload == 'config4'
abs_frac = float(sys.argv[2]) if len(sys.argv) > 2 else (0.4 if wide else 0.02)
mpc = examples.integrator_chain_mpc() if wide else examples.linear_mpc(0)
if wide:
    mode = 'sweeps'
gp = engine.GpuProblem(mpc.compile(), 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
gp.set_eps(float(np.max(gp.solve_pt(abs_frac * V)[0])), 0.25 if wide else 0.01)
roots, _ = ehm_tools.delaunay_roots(V)
cap = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 24)
gp.partition(roots, export=False, with_volume=False, max_nodes=cap)
full = gp.partition(roots, export=False, with_volume=False, max_nodes=cap)
print('full: nodes', full['n_nodes'], 'LPs', full['lp_solves'], 'ms', 1e3 * full['device_seconds'], flush=True)
for world in (2, 4, 8):
    variants = ([('depth %d' % (distributed.deal_depth_for(len(roots), world, pr)),
                  dict(deal_depth=distributed.deal_depth_for(len(roots), world, pr),
                       shard=(None, world, 0))) for pr in (128, 1024, 2048, 4096)]
                if mode == 'deal' else
                [('min_frontier %d' % mf, dict(shard=(None, world, mf))) for mf in (64 * world, 1024 * world, 4096 * world)])
    for label, kw in variants:
        lp, ms, rep, nodes = [], [], 0, 0
        for r in range(world):
            kw2 = dict(kw)
            kw2['shard'] = (r, world, kw['shard'][2])
            info = gp.partition(roots, export=False, with_volume=False, max_nodes=cap,
                                engine=0 if wide else 1, **kw2)
            ms.append(1e3 * info['device_seconds'])
            lp.append(info['lp_solves'] - info['replicated_solves'])
            rep = info['replicated_solves']
            nodes += info['n_nodes'] - (info['replicated_nodes'] if r else 0)
        lp = np.array(lp, dtype=float)
        print('world %d %-18s: nodes %d replicated LPs %7d  per-rank LPs max/mean %.3f  time max %.1f ms mean %.1f ms  speedup vs full %.2f'
              % (world, label, nodes, rep, lp.max() / lp.mean(), max(ms), np.mean(ms), 1e3 * full['device_seconds'] / max(ms)), flush=True)
gp.close()
