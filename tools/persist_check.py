"""Persistent frontier kernel (engine=1) vs level-synchronous sweeps (engine=0): identity + time."""
import sys, time
import numpy as np
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools

which = sys.argv[1] if len(sys.argv) > 1 else 'lp'
af = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
er = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
mpc = examples.linear_mpc(0, cost='quadratic' if which == 'qp' else 'inf')
gp = engine.GpuProblem(mpc.compile(), 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
gp.set_eps(float(np.max(gp.solve_pt(af * V)[0])), er)
roots, _ = ehm_tools.delaunay_roots(V)
trees = {}
for eng in (0, 1, 1, 0):
    t0 = time.perf_counter()
    info = gp.partition(roots, action='ecc', max_nodes=1 << 22, engine=eng, export=False, with_volume=False)
    dt = time.perf_counter() - t0
    print('engine', eng, 'nodes', info['n_nodes'], 'closed', info['n_closed'], 'lp', info['lp_solves'],
          'depth', info['max_depth'], '%.2f ms' % (1e3 * dt), 'dev %.2f ms' % (1e3 * info['device_seconds']),
          '%.3g LP/s' % (info['lp_solves'] / dt), flush=True)
for eng in (0, 1):
    trees[eng] = gp.partition(roots, action='ecc', max_nodes=1 << 22, engine=eng)
a, b = trees[0], trees[1]
print('identical:', a.n_nodes == b.n_nodes and np.array_equal(a.vertices, b.vertices) and
      np.array_equal(a.left, b.left) and np.array_equal(a.flags & 1, b.flags & 1) and
      np.array_equal(a.vertex_costs, b.vertex_costs) and np.array_equal(a.tstar, b.tstar),
      'vol', a.info['volume_closed'], b.info['volume_closed'])
for name in ('vertices', 'left', 'flags', 'vertex_costs', 'tstar', 'vertex_inputs'):
    x, y = getattr(a, name), getattr(b, name)
    if name == 'flags':
        x, y = x & 1, y & 1
    if not np.array_equal(x, y):
        d = np.abs(x.astype(float) - y.astype(float))
        print('  differs:', name, 'max abs', d.max(), 'count', int((d > 0).sum()))
gp.close()
