"""
Config 5 (examples.pwa4_mpc(N=8): n_x = 8, n_u = 3, four modes, 65 536 mode sequences) at a STATED
tolerance on the device, with a clock on every layer: where do the seconds of
``bnb_frontier.grow_frontier`` go -- LP launches, block condensation / upload, hand-offs to the
multi-commutation engine, the interpreter in between.

    python tools/config5_probe.py ABS_FRAC EPS_R VISITS [TABLE_MAX] [ROUND_CAP] [backoff|-] [fifo|deepest|lcss-first] [MIN_REGIONS]

One cell of the box (the Kuhn simplex on the main diagonal, 1/8! of Theta); eps_a by
lib/examples.py:42-46 (largest P_theta cost at abs_frac x the box vertices).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import bnb, bnb_frontier, engine, examples   # noqa: E402
from explicit_hybrid_mpc_amd.tree import NodeData, Tree                    # noqa: E402

CLOCK = {}


def clocked(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or name

    def wrapper(*a, **kw):
        t = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            c = CLOCK.setdefault(label, [0, 0.])
            c[0] += 1
            c[1] += time.perf_counter() - t
    setattr(owner, name, wrapper)


def main():
    abs_frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
    eps_r = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
    visits = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    table_max = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    round_cap = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
    backoff = len(sys.argv) > 6 and sys.argv[6] == 'backoff'
    order = sys.argv[7] if len(sys.argv) > 7 else 'fifo'
    min_regions = int(sys.argv[8]) if len(sys.argv) > 8 else None
    for name in ('point_idx', 'simplex_idx', 'update_blocks', 'partition'):
        clocked(engine.GpuProblem, name)
    clocked(engine.GpuProblem, '__init__', 'problem_create')
    clocked(bnb_frontier, '_hand_off')
    clocked(bnb_frontier, 'bar_e_many')
    clocked(bnb_frontier, 'bar_d_many')
    clocked(bnb_frontier, 'region_tables_many')
    mpc = examples.pwa4_mpc(N=8)
    clocked(type(mpc), 'condense_prefix')
    clocked(type(mpc), 'restrict')
    clocked(type(mpc), 'compile')
    half = examples.theta_box(mpc)
    p = mpc.n_x
    import bench
    R = bench.kuhn_cell(half, int(os.environ.get('EHM_CELL', '0')))      # 0 = the main-diagonal cell
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=8192)
    t0 = time.perf_counter()
    V = examples.box_vertices(half)
    Jabs = [j for _, _, j in bnb_frontier.p_theta_many(orc, abs_frac * V)]
    eps_a = float(np.max(Jabs))
    print('eps_a %.6g (abs_frac %g: %d P_theta searches, %.2f s), eps_r %g' %
          (eps_a, abs_frac, len(V), time.perf_counter() - t0, eps_r), flush=True)
    orc.eps_a, orc.eps_r = eps_a, eps_r
    orc.table.set_eps(eps_a, eps_r)
    lp0 = orc.table.lp_solves
    orc.table.reset_counts()
    CLOCK.clear()
    t = time.perf_counter()
    branch = Tree(NodeData(vertices=R.copy()))
    stats = bnb_frontier.grow_frontier(
        orc, branch, 'ecc', max_visits=visits, round_cap=round_cap, table_max=table_max,
        table_backoff=backoff, order=order, min_regions=min_regions,
        log=lambda m: print('  ', m, '%.1fs' % (time.perf_counter() - t), flush=True))
    wall = time.perf_counter() - t
    leaves = list(branch.leaves())
    closed = sum(1 for n, _ in leaves if n.data.is_epsilon_suboptimal)
    print('visits %d nodes %d leaves %d closed %d depth %d wall %.2f s' %
          (stats['host_visits'], sum(1 for _ in branch.walk()), len(leaves), closed,
           max(len(loc) for _, loc in leaves), wall))
    sizes = stats.pop('table_sizes')
    print('stats', stats, 'table sizes (largest)', sorted(sizes)[-8:], 'mean',
          float(np.mean(sizes)) if sizes else None)
    print('calls', dict(orc.calls), 'LPs', orc.table.lp_solves - lp0, 'expanded', orc.n_expanded,
          'blocks', orc.table.blocks_loaded)
    print('stalled', orc.table.stalled, 'of them answered "no information"',
          orc.table.stalled_relaxations)
    for kind, row in zip(('point phase one', 'point optimum', 'simplex phase one',
                          'min over simplex', 'suboptimality test'), orc.table.by_length):
        print('  LPs by prefix length, %-18s' % kind, row.tolist())
    for k, (n, s) in sorted(CLOCK.items(), key=lambda kv: -kv[1][1]):
        print('  %-20s %8d calls %8.2f s' % (k, n, s))
    orc.close()


if __name__ == '__main__':
    main()
