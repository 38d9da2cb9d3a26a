import sys, time, json, os
import numpy as np
sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import examples
from oracle import geometry
os.environ['EHM_HY_TRACE'] = '1'
known = json.load(open('tests/golden/known_answers.json'))['runs']
r = known[4]
full_set, part, oracle = examples.example('cwh_z', abs_frac=0.01, rel_err=float(r['rel_err']))
roots, locs = geometry.delaunay_simplices(full_set)
g = oracle.gpu
t0 = time.perf_counter()
info = g.partition(np.array(roots), action='ecc', max_nodes=1 << 24, export=False, with_volume=False)
dt = time.perf_counter() - t0
print(json.dumps(dict(eps_a=oracle.eps_a, eps_a_ref=r['eps_a'], leaves_ref=r['leaves'], depth_ref=r['tree_depth'], seconds=dt, info=info, stats=g.stats())))
