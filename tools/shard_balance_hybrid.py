"""Sharded runs of the multi-commutation engine, the shards one after the other on one GPU
(fourth cwh_z job: 79 468 leaves, 81 commutations).  Speed-up = full run / slowest shard."""
import sys, json, time
import numpy as np
sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import examples, distributed
from oracle import geometry
known = json.load(open('tests/golden/known_answers.json'))['runs'][3]
full_set, part, oracle = examples.example('cwh_z', abs_frac=0.03, rel_err=float(known['rel_err']))
roots, locs = geometry.delaunay_simplices(full_set)
g = oracle.gpu
R = np.array(roots)
g.partition(R, export=False, with_volume=False, max_nodes=1 << 22)
full = g.partition(R, export=False, with_volume=False, max_nodes=1 << 22)
print('full: nodes', full['n_nodes'], 'solves', full['lp_solves'], 'ms %.1f' % (1e3 * full['device_seconds']), flush=True)
for world in (2, 4, 8):
    for per_rank in (16, 64, 256):
        d = distributed.deal_depth_for(len(roots), world, per_rank)
        ms, lp, nodes = [], [], 0
        for r in range(world):
            info = g.partition(R, export=False, with_volume=False, max_nodes=1 << 22,
                               shard=(r, world, 0), deal_depth=d)
            ms.append(1e3 * info['device_seconds'])
            lp.append(info['lp_solves'] - info['replicated_solves'])
            nodes += info['n_nodes'] - (info['replicated_nodes'] if r else 0)
        lp = np.array(lp, dtype=float)
        print('world %d depth %2d: nodes %d  per-rank solves max/mean %.3f  time max %.1f ms mean %.1f ms  speedup %.2f'
              % (world, d, nodes, lp.max() / lp.mean(), max(ms), np.mean(ms), 1e3 * full['device_seconds'] / max(ms)), flush=True)
oracle.close()
