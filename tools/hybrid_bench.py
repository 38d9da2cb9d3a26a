"""Hybrid device engine: config 3 (pwa, 32 commutations) at a few tolerances, and the reference's
cwh_z jobs.  gpurun -- 'PYTHONPATH=. python tools/hybrid_bench.py [njobs]'"""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from tests import helpers
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools
from oracle import geometry

njobs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mpc = helpers.make_instance('pwa', 0)
can = mpc.compile()
print('pwa: n=%d m=%d p=%d n_delta=%d' % (can.n, can.m, can.p, can.n_delta), flush=True)
gp = engine.GpuProblem(can, 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
roots, _ = ehm_tools.delaunay_roots(V)
for af, er in ((0.5, 0.5), (0.25, 0.3), (0.1, 0.1), (0.05, 0.05)):
    J = gp.solve_pt(af * V)[0]
    eps_a = float(np.max(J[np.isfinite(J)])); gp.set_eps(eps_a, er)
    for rep in range(2):
        s0 = gp.stats()
        t0 = time.perf_counter()
        try:
            info = gp.partition(roots, action='ecc', max_nodes=1 << 22, export=False,
                                with_volume=False)
        except RuntimeError as e:
            print('  (stopped:', e, ')'); info = None
        dt = time.perf_counter() - t0
        s1 = gp.stats()
    lp = s1['lp_solves'] - s0['lp_solves']
    print(json.dumps(dict(abs_frac=af, eps_r=er, eps_a=eps_a, wall=dt, lp=lp, lp_per_s=lp / dt,
                          launches=s1['kernel_launches'] - s0['kernel_launches'],
                          info=info)), flush=True)
    if info is None or dt > 20:
        break
gp.close()

known = json.load(open('tests/golden/known_answers.json'))['runs']
fracs = [0.5, 0.25, 0.1, 0.03, 0.01]
for k in range(njobs):
    r = known[k]
    full_set, part, oracle = examples.example('cwh_z', abs_frac=fracs[k], rel_err=float(r['rel_err']))
    roots, locs = geometry.delaunay_simplices(full_set)
    g = oracle.gpu
    for rep in range(2 if k < 4 else 1):
        s0 = g.stats()
        t0 = time.perf_counter()
        info = g.partition(np.array(roots), action='ecc', max_nodes=(1 << 22) if k < 4 else (1 << 23),
                           export=False, with_volume=False)
        dt = time.perf_counter() - t0
        s1 = g.stats()
    rec = dict(abs_frac=fracs[k], rel_err=r['rel_err'], eps_a=oracle.eps_a, eps_a_ref=r['eps_a'],
               nodes=info['n_nodes'], leaves=info['n_leaves'], leaves_ref=r['leaves'],
               depth=info['max_depth'], depth_ref=r['tree_depth'], seconds=dt,
               device_seconds=info['device_seconds'], sweeps=info['sweeps'],
               lp_solves=int(info['lp_solves']), min_margin=float(info['min_margin']),
               swaps=info['swaps'], launches=s1['kernel_launches'] - s0['kernel_launches'])
    rec.update({k2: int(v) for k2, v in g.stats().items() if k2 in ('slivers', 'fallbacks', 'stalled')})
    print(json.dumps(rec), flush=True)
    oracle.close()
