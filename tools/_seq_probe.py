"""Exploration: the bnb driver at N = 8 on a Kuhn simplex of the whole box (config 5's top)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import examples, bnb
from explicit_hybrid_mpc_amd.tree import Tree, NodeData
mpc = examples.pwa4_mpc(N=8)
half = examples.theta_box(mpc)
p = 8
R = np.array([-half + 2 * half * (np.arange(p) < k) for k in range(p + 1)])
orc = bnb.PrefixOracle(mpc, 1., 1., slots=8192)
t = time.time()
J = [orc.P_theta(v)[2] for v in R]
print('vertex optimal costs', np.round(J, 4), 'LPs', orc.table.lp_solves, 'expanded', orc.n_expanded, '%.1fs' % (time.time() - t), flush=True)
Jm = max(J)
for eps_a_frac, eps_r in ((0.5, 1.0), (0.2, 0.3)):
    orc.eps_a, orc.eps_r = eps_a_frac * Jm, eps_r
    orc.table.set_eps(orc.eps_a, eps_r)
    lp0, ex0 = orc.table.lp_solves, orc.n_expanded
    t = time.time()
    branch = Tree(NodeData(vertices=R.copy()))
    stats = bnb.grow(orc, branch, 'ecc', max_visits=40, log=lambda s: print('  ', s, flush=True))
    leaves = list(branch.leaves())
    print(eps_a_frac, eps_r, 'nodes', sum(1 for _ in branch.walk()), 'leaves', len(leaves),
          'closed', sum(1 for n, _ in leaves if n.data.is_epsilon_suboptimal), stats,
          'calls', orc.calls, 'LPs', orc.table.lp_solves - lp0, 'expanded', orc.n_expanded - ex0,
          'blocks', orc.table.blocks_loaded, '%.1fs' % (time.time() - t), flush=True)
orc.close()
