import sys, time
import numpy as np
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools
mpc = examples.linear_mpc(0)
gp = engine.GpuProblem(mpc.compile(), 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
gp.set_eps(float(np.max(gp.solve_pt(0.02 * V)[0])), 0.01)
roots, _ = ehm_tools.delaunay_roots(V)
for k in range(3):
    t0 = time.perf_counter()
    info = gp.partition(roots, max_nodes=1 << 22, export=False, with_volume=False)
    dt = time.perf_counter() - t0
print('sign-only', info['n_nodes'], info['n_closed'], info['decide_iters'] / info['decide_solves'], '%.2f ms' % (1e3 * dt), info['min_margin'])
a = gp.partition(roots, max_nodes=1 << 22)
gp.set_option('decide_full', 1)
b = gp.partition(roots, max_nodes=1 << 22, engine=0)
print('same as full-accuracy sweeps:', a.n_nodes == b.n_nodes and np.array_equal(a.left, b.left) and np.array_equal(a.flags & 1, b.flags & 1) and np.array_equal(a.vertices, b.vertices))
gp.close()
