"""Where the host time of one config-5 cell goes, on the GPU box: bnb_frontier.grow_frontier on
the whole-box Kuhn cell at the stated tolerances (bench.py --workload config5) under cProfile.
    python tools/config5_host_profile.py [top=45] [cold|sweep]
Prints wall seconds, the device's share (ehm_counters.batch_seconds) and the profile sorted by
own time."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from explicit_hybrid_mpc_amd import bnb, bnb_frontier, examples
from explicit_hybrid_mpc_amd.tree import NodeData, Tree

top = int(sys.argv[1]) if len(sys.argv) > 1 else 45
mpc = examples.pwa4_mpc(N=8)
half = examples.theta_box(mpc)
R = np.array([-half + 2 * half * (np.arange(8) < k) for k in range(9)])
orc = bnb.PrefixOracle(mpc, 1., 1., slots=8192)
V = examples.box_vertices(half)
eps_a = float(np.max([j for _, _, j in bnb_frontier.p_theta_many(orc, 0.2 * V)]))
orc.eps_a, orc.eps_r = eps_a, 1e-3
orc.table.set_eps(eps_a, 1e-3)
if len(sys.argv) > 2 and sys.argv[2] == 'sweep':
    # round size x problems a best-first step aims at per launch
    for cap in (2048, 4096, 8192, 16384):
        for target in (4096, 16384):
            bnb_frontier.LAUNCH_TARGET = target
            root = Tree(NodeData(vertices=R.copy()))
            lp0 = orc.table.lp_solves
            t0 = time.perf_counter()
            stats = bnb_frontier.grow_frontier(orc, root, 'ecc', round_cap=cap, order='lcss-first',
                                               table_backoff=True)
            print('round_cap %5d launch target %5d: %.2f s, %d regions, %d rounds, %d LPs'
                  % (cap, target, time.perf_counter() - t0, stats['regions'], stats['rounds'],
                     orc.table.lp_solves - lp0), flush=True)
    orc.close()
    sys.exit(0)
cold = len(sys.argv) > 2 and sys.argv[2] == 'cold'           # profile the FIRST pass instead
for rep in range(1 if cold else 2):                   # second pass: warm tables, profiled
    root = Tree(NodeData(vertices=R.copy()))
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if rep or cold:
        pr.enable()
    stats = bnb_frontier.grow_frontier(orc, root, 'ecc', round_cap=2048, order='lcss-first',
                                       table_backoff=True)
    if rep or cold:
        pr.disable()
    print('pass %d: %.2f s, %d regions, %d rounds' % (rep, time.perf_counter() - t0,
                                                      stats['regions'], stats['rounds']), flush=True)
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(top)
orc.close()
