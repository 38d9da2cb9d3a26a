"""The reference's cwh_z jobs (lib/post_process.py:484-526, make_jobs.sh:60-66) on the device.

    python tools/cwh_jobs.py [n_jobs=3] [n_seeds=0]

n_seeds > 0: after the canonical run of every job, n_seeds runs under option "any_admissible"
(include/ehmpc.h: V_R and bar_D return a hashed draw among the admissible commutations -- what the
reference's Minimize(0) leaves to MOSEK), seeds 0 .. n_seeds-1 (the largest jobs take fewer: see
SEED_CAP).  One JSON line per run; leaves / depth next to the reference's published figures.
"""
import os
import sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from explicit_hybrid_mpc_amd import examples, partition
from oracle import geometry

known = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'tests', 'golden', 'known_answers.json')))['runs']
fracs = [0.5, 0.25, 0.1, 0.03, 0.01]
SEED_CAP = {3: 4, 4: 2}          # job index -> at most this many drawn runs (minutes each)
njobs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for k in range(njobs):
    r = known[k]
    full_set, part, oracle = examples.example('cwh_z', abs_frac=fracs[k], rel_err=float(r['rel_err']))
    roots, locs = geometry.delaunay_simplices(full_set)
    # 'last', 'smallest', 'last+smallest': deterministic extremes (include/ehmpc.h, option
    # any_admissible = 2^30 + m) -- the envelope of what the choice can do
    extremes = [('V_R last', 1), ('bar_D smallest slack', 2), ('V_R last + bar_D smallest slack', 3)] \
        if nseeds > 0 else []
    for seed in [None] + extremes + list(range(min(nseeds, SEED_CAP.get(k, nseeds)))):
        mode = None
        if isinstance(seed, tuple):
            mode, seed = seed[0], None
            oracle.gpu.set_option('any_admissible', float((1 << 30) + dict(extremes)[mode]))
        else:
            oracle.gpu.set_option('any_admissible', 0 if seed is None else seed + 1)
        t0 = time.time()
        try:
            flat = partition.run_engine(oracle, np.array(roots), action='ecc', max_nodes=1 << 24)
        except Exception as e:
            print(json.dumps(dict(job=k + 1, rule=mode or ('canonical' if seed is None else
                                                           'any admissible'),
                                  seed=seed, error=str(e)[:200])), flush=True)
            continue
        dt = time.time() - t0
        leaves = int(np.sum(flat.left < 0))
        loc = flat.locations(locs)
        depth = max(len(l) for l in loc) - 1
        rec = dict(job=k + 1, rule=mode or ('canonical' if seed is None else 'any admissible'), seed=seed,
                   abs_frac=fracs[k], rel_err=r['rel_err'], eps_a=oracle.eps_a, eps_a_ref=r['eps_a'],
                   nodes=flat.n_nodes, leaves=leaves, leaves_ref=r['leaves'], depth=depth,
                   depth_ref=r['tree_depth'], seconds=dt, lp_solves=int(flat.info.get('lp_solves', 0)),
                   min_margin=float(flat.info.get('min_margin', 0)))
        rec.update({q: int(v) for q, v in oracle.gpu.stats().items()
                    if q in ('slivers', 'fallbacks', 'stalled')})
        print(json.dumps(rec), flush=True)
        del flat
    oracle.close()
