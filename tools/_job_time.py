"""Device time of one of the reference's cwh_z jobs (index 0..4) on the multi-commutation engine."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import examples
from oracle import geometry
known = json.load(open('tests/golden/known_answers.json'))['runs']
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
fracs = [0.5, 0.25, 0.1, 0.03, 0.01]
r = known[k]
full_set, part, oracle = examples.example('cwh_z', abs_frac=fracs[k], rel_err=float(r['rel_err']))
roots, locs = geometry.delaunay_simplices(full_set)
g = oracle.gpu
for timing in (0, 0, 1):
    g.set_option('timing', timing)
    t0 = time.perf_counter()
    info = g.partition(np.array(roots), action='ecc', max_nodes=1 << 23, export=False, with_volume=False)
    dt = time.perf_counter() - t0
    print('job', k + 1, 'timing', timing, 'wall %.3f' % dt, 'device %.3f' % info['device_seconds'], 'leaves', info['n_leaves'],
          'lp', info['lp_solves'], 'sweeps', info['sweeps'], 'margin %.3g' % info['min_margin'], flush=True)
print({q: info[q] for q in ('decide_seconds', 'expand_seconds', 'kind_solves', 'kind_iters')})
oracle.close()
