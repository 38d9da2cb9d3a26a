import sys, time, numpy as np
sys.path.insert(0, '.')
from tests import helpers
from explicit_hybrid_mpc_amd import engine, examples, partition
from explicit_hybrid_mpc_amd import tools as ehm_tools
mpc = helpers.make_instance('pwa', 0)
can = mpc.compile()
print('pwa: n=%d m=%d p=%d n_delta=%d' % (can.n, can.m, can.p, can.n_delta))
gp = engine.GpuProblem(can, 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
roots, _ = ehm_tools.delaunay_roots(V)
for af, er in ((0.5, 1.0), (0.25, 0.3)):
    J = gp.solve_pt(af * V)[0]
    eps_a = float(np.max(J[np.isfinite(J)])); gp.set_eps(eps_a, er)
    s0 = gp.stats()
    t0 = time.perf_counter()
    try:
        flat = partition.grow_hybrid(gp, roots, action='ecc', max_nodes=150000)
    except RuntimeError as e:
        print('  (stopped:', e, ')'); flat = None
    dt = time.perf_counter() - t0
    s1 = gp.stats()
    print('abs_frac %.2f eps_r %.2f: nodes %s  LP %d  launches %d  wall %.3fs  %.3g LP/s' % (
        af, er, (flat.n_nodes if flat else '>150000'), s1['lp_solves'] - s0['lp_solves'],
        s1['kernel_launches'] - s0['kernel_launches'], dt, (s1['lp_solves'] - s0['lp_solves']) / dt),
        'stalled', s1['stalled'] - s0['stalled'], 'fallbacks', s1['fallbacks'] - s0['fallbacks'])
