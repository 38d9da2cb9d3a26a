"""Exploration: bnb.grow on a Kuhn simplex of the whole box at N = 8 (config 5's top), to completion
or a budget of host visits."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import examples, bnb, bnb_frontier
from explicit_hybrid_mpc_amd.tree import Tree, NodeData
budget = int(sys.argv[1]) if len(sys.argv) > 1 else 600
mpc = examples.pwa4_mpc(N=8)
half = examples.theta_box(mpc)
p = 8
R = np.array([-half + 2 * half * (np.arange(p) < k) for k in range(p + 1)])
orc = bnb.PrefixOracle(mpc, 1., 1., slots=8192)
J = [orc.P_theta(v)[2] for v in R]
Jm = max(J)
orc.eps_a, orc.eps_r = 0.5 * Jm, 1.0
orc.table.set_eps(orc.eps_a, 1.0)
t = time.time()
branch = Tree(NodeData(vertices=R.copy()))
frontier = len(sys.argv) > 2 and sys.argv[2] == 'frontier'
if frontier:
    stats = bnb_frontier.grow_frontier(orc, branch, 'ecc', max_visits=budget, round_cap=16384,
                                       log=lambda m: print('  ', m, '%.1fs' % (time.time() - t), flush=True))
else:
    stats = bnb.grow(orc, branch, 'ecc', max_visits=budget)
leaves = list(branch.leaves())
depth = max(len(loc) for _, loc in leaves)
print('budget', budget, 'nodes', sum(1 for _ in branch.walk()), 'leaves', len(leaves), 'depth', depth,
      'closed', sum(1 for n, _ in leaves if n.data.is_epsilon_suboptimal),
      {k: v for k, v in stats.items() if k != 'table_sizes'}, 'tables', sorted(stats['table_sizes'])[-5:],
      'calls', orc.calls, 'LPs', orc.table.lp_solves, 'expanded', orc.n_expanded,
      'inherited', orc.n_inherited, 'blocks', orc.table.blocks_loaded,
      'memo (verdicts, points, pairs solved, pairs shared)', orc.table.search_counts(),
      'optima asked / solved', getattr(orc.table, 'optima_asked', 0),
      getattr(orc.table, 'optima_solved', 0), '%.1fs' % (time.time() - t), flush=True)
orc.close()
