"""Static dealing of the frontier over `world` ranks: per-rank work, measured on ONE GPU by running
the shards one after the other (engine 0 so that the deal happens inside the sweeps)."""
import sys, time
import numpy as np
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools

mpc = examples.linear_mpc(0)
gp = engine.GpuProblem(mpc.compile(), 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
gp.set_eps(float(np.max(gp.solve_pt(0.02 * V)[0])), 0.01)
roots, _ = ehm_tools.delaunay_roots(V)
full = gp.partition(roots, export=False, with_volume=False, max_nodes=1 << 22)
print('full', full['lp_solves'], 'ms', 1e3 * full['device_seconds'], flush=True)
for world in (2, 4, 8):
    for mf in (64 * world, 1024 * world, 8192 * world):
        lp, ms, rep = [], [], 0
        for r in range(world):
            t0 = time.perf_counter()
            info = gp.partition(roots, export=False, with_volume=False, max_nodes=1 << 22,
                                shard=(r, world, mf), engine=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
            ms.append(1e3 * (time.perf_counter() - t0))
            lp.append(info['lp_solves'] - info['replicated_solves'])
            rep = info['replicated_solves']
        lp = np.array(lp, dtype=float)
        print('world %d min_frontier %6d: replicated %7d  per-rank LPs max/mean %.3f  time max %.1f ms mean %.1f ms  speedup vs full %.2f'
              % (world, mf, rep, lp.max() / lp.mean(), max(ms), np.mean(ms), 1e3 * full['device_seconds'] / max(ms)), flush=True)
gp.close()
