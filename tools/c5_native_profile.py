"""
configs[4] on the native driver (include/ehm_frontier.h), cell by cell in slices of node visits:
where the time of ``ehm_frontier_run`` goes (inside the batched solver calls / kernel seconds by
HIP events / the driver's own bookkeeping), launches, LPs by table and kind, cells handed back.

    python tools/c5_native_profile.py --cells 0,3 --slice 20000 --seconds 120
"""

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cells', default='0')
    ap.add_argument('--slice', type=int, default=20000, help='node visits per call')
    ap.add_argument('--seconds', type=float, default=120., help='per cell')
    ap.add_argument('--round-cap', type=int, default=4096)
    ap.add_argument('--launch-target', type=int, default=65536)
    ap.add_argument('--speculate', type=int, default=0)
    ap.add_argument('--horizons', default='', help='e.g. 4,8 or 4,5,6,7,8 (default: one per '
                                                   'horizon above the short one)')
    ap.add_argument('--slots', type=int, default=16384)
    ap.add_argument('--max-depth', type=int, default=0)
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier, examples, frontier
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    mpc = examples.pwa4_mpc(N=8, seed=0)
    V = examples.box_vertices(examples.theta_box(mpc))
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=1024, device=0)
    eps_a = float(np.max([j for _, _, j in bnb_frontier.p_theta_many(orc, 0.2 * V)]))
    orc.close()
    roots, _ = ehm_tools.delaunay_roots(V)
    horizons = [int(x) for x in args.horizons.split(',')] if args.horizons else None
    nat = frontier.NativeFrontier(mpc, eps_a, 1e-3, slots=args.slots, horizons=horizons)
    print('tables of horizons %s' % nat.horizons, flush=True)
    out = []
    for c in [int(x) for x in args.cells.split(',')]:
        nat.reset()
        nat.add_roots(roots[c:c + 1])
        t0 = time.perf_counter()
        tab0 = nat.table_stats()
        visits = 0
        while True:
            visits += args.slice
            st = nat.run(round_cap=args.round_cap, launch_target=args.launch_target,
                         max_visits=visits, speculate=args.speculate, max_depth=args.max_depth)
            el = time.perf_counter() - t0
            print('cell %d: %6.1f s  visits %8d  nodes %8d  regions %8d  at depth limit %6d  depth %2d  LPs '
                  '%9d  solver calls %6d  in solvers %.1f s' % (
                      c, el, st['visits'], st['n_nodes'], st['regions'], st['depth_limited'],
                      st['depth'], st['lp_solves'], st['launches'], st['seconds_solvers']),
                  flush=True)
            if not st['truncated'] or el > args.seconds:
                break
        tab1 = nat.table_stats()
        kern = {'h%d' % b['horizon']: [y - x for x, y in zip(a['batch_seconds'], b['batch_seconds'])]
                for a, b in zip(tab0, tab1)}
        launches = {'h%d' % b['horizon']: [y - x for x, y in zip(a['batch_launches'],
                                                                  b['batch_launches'])]
                    for a, b in zip(tab0, tab1)}
        kern['evicted'] = {'h%d' % b['horizon']: b['evicted'] for b in tab1}
        rec = dict(cell=c, wall=el, finished=not st['truncated'], stats=st,
                   kernel_seconds_point_simplex=kern, kernel_launches_point_simplex=launches,
                   lp_by_table_kind_length=nat.lp_counts().tolist())
        out.append(rec)
        print(json.dumps({k: v for k, v in rec.items() if k != 'lp_by_table_kind_length'}))
    nat.close()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
