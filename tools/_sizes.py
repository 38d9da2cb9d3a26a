import sys, time, numpy as np
sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools
mpc = examples.linear_mpc(seed=0); can = mpc.compile()
gp = engine.GpuProblem(can, 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
roots, _ = ehm_tools.delaunay_roots(V)
for af in (0.03, 0.025, 0.02, 0.017, 0.015):
    eps_a = float(np.max(gp.solve_pt(af * V)[0])); gp.set_eps(eps_a, 1e-2)
    for rep in range(2):
        t0 = time.perf_counter()
        i = gp.partition(roots, action='ecc', max_nodes=1 << 23, export=False, with_volume=False)
        dt = time.perf_counter() - t0
    print('abs_frac %.3f eps_a %.5f: nodes %d leaves %d LP %d sweeps %d depth %d wall %.3fs LP/s %.3g margin %.2e' % (
        af, eps_a, i['n_nodes'], i['n_closed'], i['lp_solves'], i['sweeps'], i['max_depth'], dt, i['lp_solves'] / dt, i['min_margin']))
