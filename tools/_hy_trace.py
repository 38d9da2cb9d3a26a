import sys, time, json, os
import numpy as np
sys.path.insert(0, '.')
from tests import helpers
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools
os.environ['EHM_HY_TRACE'] = '1'
af, er = float(sys.argv[1]), float(sys.argv[2])
mpc = helpers.make_instance('pwa', 0)
can = mpc.compile()
gp = engine.GpuProblem(can, 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
roots, _ = ehm_tools.delaunay_roots(V)
J = gp.solve_pt(af * V)[0]
eps_a = float(np.max(J[np.isfinite(J)])); gp.set_eps(eps_a, er)
print('eps_a', eps_a, 'box', examples.theta_box(mpc))
t0 = time.perf_counter()
try:
    info = gp.partition(roots, action='ecc', max_nodes=int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20, export=False, with_volume=False)
    print(info)
except RuntimeError as e:
    print('stopped', e)
print(time.perf_counter() - t0)
