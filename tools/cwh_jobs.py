"""The reference's cwh_z jobs (lib/post_process.py:484-526, make_jobs.sh:60-66) on the device."""
import sys, time, json
import numpy as np
from explicit_hybrid_mpc_amd import examples, partition
from oracle import geometry

known = json.load(open('tests/golden/known_answers.json'))['runs']
fracs = [0.5, 0.25, 0.1, 0.03, 0.01]
njobs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out = []
for k in range(njobs):
    r = known[k]
    full_set, part, oracle = examples.example('cwh_z', abs_frac=fracs[k], rel_err=float(r['rel_err']))
    roots, locs = geometry.delaunay_simplices(full_set)
    t0 = time.time()
    flat = partition.run_engine(oracle, np.array(roots), action='ecc', max_nodes=1 << 22)
    dt = time.time() - t0
    leaves = int(sum(flat.is_leaf(i) for i in range(flat.n_nodes)))
    loc = flat.locations(locs)
    depth = max(len(l) for l in loc) - 1
    rec = dict(abs_frac=fracs[k], rel_err=r['rel_err'], eps_a=oracle.eps_a, eps_a_ref=r['eps_a'],
               nodes=flat.n_nodes, leaves=leaves, leaves_ref=r['leaves'], depth=depth,
               depth_ref=r['tree_depth'], seconds=dt, lp_solves=int(flat.info.get('lp_solves', 0)),
               min_margin=float(flat.info.get('min_margin', 0)))
    rec.update({k: int(v) for k, v in oracle.gpu.stats().items() if k in ('slivers', 'fallbacks', 'stalled')})
    print(json.dumps(rec), flush=True)
    oracle.close()
