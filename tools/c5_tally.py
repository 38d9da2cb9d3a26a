"""configs[4]: slack problems by caller (EHM_FR_TALLY=1 of csrc/ehm_frontier.cpp) on one Delaunay
root -- how the LPs of a deep root split between the suboptimality test (bar_E: a closed cell
needs every prefix refuted) and the best-slack search of the open cells (bar_D).

    python tools/c5_tally.py [--root 3] [--max-depth 26]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--root', type=int, default=3)
    ap.add_argument('--max-depth', type=int, default=26)
    ap.add_argument('--max-visits', type=int, default=0)
    args = ap.parse_args()
    os.environ['EHM_FR_TALLY'] = '1'
    from explicit_hybrid_mpc_amd import examples, frontier
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    mpc = examples.pwa4_mpc(N=8, seed=0)
    V = examples.box_vertices(examples.theta_box(mpc))
    nat = frontier.NativeFrontier(mpc, 1., 1.)
    eps_a = max(j for _, _, j in nat.p_theta(0.2 * V))
    nat.set_eps(eps_a, 1e-3)
    roots, _ = ehm_tools.delaunay_roots(V)
    nat.reset()
    nat.add_roots(roots[args.root:args.root + 1])
    t0 = time.perf_counter()
    st = nat.run(max_depth=args.max_depth, max_visits=args.max_visits)
    dt = time.perf_counter() - t0
    kinds = nat.lp_counts().sum(axis=0)
    print('root %d: %.1f s, %d regions, %d nodes, %d LPs; calls V_R %d P_theta %d bar_E %d bar_D %d; '
          'LPs by kind (point phase one / point optimum / simplex phase one / min over simplex / '
          'slack): %s' % (args.root, dt, st['regions'], st['n_nodes'], st['lp_solves'],
                          st['calls_v_r'], st['calls_p_theta'], st['calls_bar_e'],
                          st['calls_bar_d'], kinds.sum(axis=1).tolist()))
    print('slack LPs by prefix length: %s' % kinds[4].tolist())
    nat.close()


if __name__ == '__main__':
    main()
