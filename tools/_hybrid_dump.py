import sys, re, numpy as np
sys.path.insert(0, '.')
from tests import helpers
from explicit_hybrid_mpc_amd import engine, examples, partition, _capi
from explicit_hybrid_mpc_amd import tools as ehm_tools
mpc = helpers.make_instance('pwa', 0)
can = mpc.compile()
gp = engine.GpuProblem(can, 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
roots, _ = ehm_tools.delaunay_roots(V)
af, er = float(sys.argv[1]), float(sys.argv[2])
J = gp.solve_pt(af * V)[0]
eps_a = float(np.max(J[np.isfinite(J)])); gp.set_eps(eps_a, er)
for name in ('bar_e', 'bar_d'):
    orig = getattr(gp, name)
    def wrapped(R, Vb, *a, _o=orig, _n=name):
        try:
            return _o(R, Vb, *a)
        except _capi.EhmError as e:
            m = re.search(r'instance (\d+), commutation (\d+)', str(e))
            if m:
                k, d = int(m.group(1)), int(m.group(2))
                np.savez('gpurun_out/fail_slack.npz', R=R[k], V=Vb[k], d=d, eps_a=eps_a, eps_r=er)
                print('saved', _n, k, d)
            raise
    setattr(gp, name, wrapped)
try:
    partition.grow_hybrid(gp, roots, action='ecc', max_nodes=150000)
except Exception as e:
    print('stopped:', e)
