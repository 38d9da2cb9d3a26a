import sys, time, json, os
import numpy as np
sys.path.insert(0, '.')
from tests import helpers
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools
af, er, md = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
mpc = helpers.make_instance('pwa', 0)
can = mpc.compile()
gp = engine.GpuProblem(can, 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
roots, _ = ehm_tools.delaunay_roots(V)
J = gp.solve_pt(af * V)[0]
eps_a = float(np.max(J[np.isfinite(J)])); gp.set_eps(eps_a, er)
flat = gp.partition(roots, action='ecc', max_nodes=8000000, max_depth=md)
print(flat.info)
leaf = flat.left < 0
openl = leaf & ((flat.flags & 1) == 0) & ((flat.flags & 2) != 0)
print('open leaves at max depth', openl.sum(), 'closed', (leaf & ((flat.flags & 1) != 0)).sum())
idx = np.where(openl)[0]
Vt = flat.vertices[idx]
ctr = Vt.mean(axis=1)
print('centre x1 range', ctr[:, 0].min(), ctr[:, 0].max(), 'abs x1 quantiles', np.quantile(np.abs(ctr[:, 0]), [0.1, 0.5, 0.9]))
print('centre norm quantiles', np.quantile(np.abs(ctr).max(axis=1), [0.1, 0.5, 0.9]))
print('cost quantiles', np.quantile(flat.vertex_costs[idx], [0.1, 0.5, 0.9]), 'eps_a', eps_a)
print('tstar quantiles', np.quantile(flat.tstar[idx], [0.1, 0.5, 0.9]))
print('didx histogram', np.bincount(flat.delta_idx[idx], minlength=32))
sel = idx[:8]
for k in sel:
    R = flat.vertices[k]; Vb = flat.vertex_costs[k]
    t = []
    feas = gp.feas_all(R)
    for d in range(can.n_delta):
        tt, al, st = gp.slack(R[None], Vb[None], can.deltas[d])
        t.append(float(tt[0]) if st[0] == 0 else np.nan)
    t = np.array(t)
    print('node', k, 'didx', flat.delta_idx[k], 'x1 of vertices', np.round(R[:, 0], 4), 'Vbar', np.round(Vb, 4))
    print('   t*', np.round(t, 4))
    print('   vfeas all', feas.all(axis=0).astype(int), 'any', feas.any(axis=0).astype(int))
    print('   deltas best', can.deltas[int(np.nanargmax(t))], 'own', can.deltas[flat.delta_idx[k]])
