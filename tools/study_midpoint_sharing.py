"""
Study (CPU, test infrastructure -- it grows the tree with the oracle): how many of the midpoint
problems of a partition are the SAME problem.

Every split solves P_theta_delta at the midpoint of the split edge (lib/worker.py:406-407).  The
simplices around an edge all bisect it at the same point, and on a single-commutation instance (the
bench workload) the problem at that point is the same LP whoever asks -- so a partition solves one
midpoint LP per split where one per distinct midpoint would do.  This script grows the config-2
instance with the CPU oracle (one process per Delaunay root) and counts both.

    PYTHONPATH=. python tools/study_midpoint_sharing.py [abs_frac=0.1] [eps_r=0.01] [procs=8] [max_visits_per_root]
    EHM_STUDY_INSTANCE=config4 selects the six-dimensional single-commutation instance of
    bench.py --workload config4 (a budget of visits per root keeps it within minutes).
"""
import multiprocessing as mp
import sys
import time

import numpy as np


def _instance(seed):
    import os
    from explicit_hybrid_mpc_amd import examples
    if os.environ.get('EHM_STUDY_INSTANCE') == 'config4':
        return examples.integrator_chain_mpc()
    return examples.linear_mpc(seed=seed)


def _grow(job):
    seed, eps_a, eps_r, root, loc, max_visits = job
    from explicit_hybrid_mpc_amd import examples
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    orc = OracleCPU(_instance(seed), eps_a, eps_r)
    part = PartitionCPU(orc, max_nodes=max_visits)
    part.run([root], [loc], 'ecc')
    mids, depth = [], []
    for name, nd in part.nodes.items():
        if nd['leaf']:
            continue
        a, b = part.nodes[name + '0']['vertices'], part.nodes[name + '1']['vertices']
        diff = np.flatnonzero(np.any(a != nd['vertices'], axis=1))
        mids.append(a[diff[0]].tobytes())
        depth.append(len(name) - len(loc))
    return mids, depth, len(part.nodes), orc.n_solves, part.truncated


def main():
    abs_frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
    eps_r = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    max_visits = int(sys.argv[4]) if len(sys.argv) > 4 else None
    from explicit_hybrid_mpc_amd import examples
    from oracle.oracle_cpu import OracleCPU
    from oracle import geometry
    mpc = _instance(0)
    V = examples.box_vertices(examples.theta_box(mpc))
    eps_a = max(OracleCPU(mpc, 1., 1.).P_theta(abs_frac * v)[2] for v in V)
    roots, locs = geometry.delaunay_simplices(V)
    t0 = time.time()
    with mp.get_context('spawn').Pool(procs) as pool:
        res = pool.map(_grow, [(0, eps_a, eps_r, R, loc, max_visits) for R, loc in zip(roots, locs)],
                       chunksize=1)
    mids = [m for r in res for m in r[0]]
    depth = np.array([d for r in res for d in r[1]])
    nodes = sum(r[2] for r in res)
    uniq = len(set(mids))
    print('abs_frac %g eps_r %g eps_a %.6g: %d roots, %d nodes, %d LP solves, %.0f s%s' % (
        abs_frac, eps_r, eps_a, len(roots), nodes, sum(r[3] for r in res), time.time() - t0,
        ' (TRUNCATED per root)' if any(r[4] for r in res) else ''))
    print('splits (midpoint problems solved) %d, distinct midpoints %d: %.2f splits per midpoint' % (
        len(mids), uniq, len(mids) / max(uniq, 1)))
    seen = set()
    for lo in range(0, int(depth.max()) + 1, 4):
        sel = [m for m, d in zip(mids, depth) if lo <= d < lo + 4]
        if sel:
            print('  depth %2d-%2d: %7d splits, %7d distinct' % (lo, lo + 3, len(sel), len(set(sel))))


if __name__ == '__main__':
    main()
