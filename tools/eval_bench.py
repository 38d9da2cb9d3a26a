#!/usr/bin/env python
"""Throughput of the batched explicit-MPC evaluation on the bench partition (needs a GPU)."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import engine, examples, explicit          # noqa: E402
from explicit_hybrid_mpc_amd import tools as ehm_tools                  # noqa: E402

mpc = examples.linear_mpc(seed=0)
can = mpc.compile()
gp = engine.GpuProblem(can, 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
roots, _ = ehm_tools.delaunay_roots(V)
abs_frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
gp.set_eps(float(np.max(gp.solve_pt(abs_frac * V)[0])), 1e-2)
flat = gp.partition(roots, action='ecc', max_nodes=1 << 23)
gp.close()
ex = explicit.ExplicitMPC(flat)
rng = np.random.default_rng(0)
half = examples.theta_box(mpc)
for n in (1, 1000, 1000000, 4000000):
    X = rng.uniform(-1, 1, (n, can.p)) * half
    ex.evaluate(X[:min(n, 1000)])
    t0 = time.perf_counter()
    u, leaf, visited, secs = ex.evaluate(X, return_info=True)
    wall = time.perf_counter() - t0
    rec_bytes = 8 * ((can.p + can.p * can.p + 7) // 8 * 8) + 8
    traffic = visited.sum() * rec_bytes + n * (8 * can.p + 8 * can.n_u + 8 * (can.p + 1) * can.n_u)
    print('n=%8d  nodes %d  tests/query %.1f  kernel %.3f ms  %.3g queries/s  %.1f GB/s algorithmic  '
          '(wall incl. PCIe %.3f ms)' % (n, flat.n_nodes, visited.mean(), secs * 1e3, n / max(secs, 1e-9),
                                         traffic / max(secs, 1e-9) / 1e9, wall * 1e3))
