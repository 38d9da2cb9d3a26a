"""
configs[4] (n_x = 8, n_u = 3, 4 modes, N = 8: 65 536 sequences) against a checker that shares no
code with the product's search -- the CI form of tools/config5_independent_check.py (whose full
run, 240 nodes of 6 Delaunay roots, is profiles/r5/config5_independent_check.json): a small
Delaunay root is grown by the native driver and a handful of its nodes are re-decided by ONE
mixed-integer LP per oracle call (HiGHS branch-and-bound on the reference's big-M statement of the
law, oracle/milp_check.py):

* cells split without a commutation: V_R's MILP (a trajectory copy per vertex, shared mode
  indicators) is infeasible;
* closed leaves: vertex costs = the uncondensed fixed-sequence LP; where the commutation was
  adopted at the cell it is the lexicographic minimum of V_R's MILP; bar_E's MILP has max t < 0.
"""

import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                'tools'))


def test_nodes_of_a_native_tree_against_one_milp_per_oracle_call():
    import config5_independent_check as c5
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier, examples, frontier
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from explicit_hybrid_mpc_amd.tree import NodeData, Tree
    mpc = examples.pwa4_mpc(N=c5.N_STEPS, seed=c5.SEED)
    V = examples.box_vertices(examples.theta_box(mpc))
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=1024)
    eps_a = float(np.max([j for _, _, j in bnb_frontier.p_theta_many(orc, c5.ABS_FRAC * V)]))
    orc.close()
    roots, _ = ehm_tools.delaunay_roots(V)
    nat = frontier.NativeFrontier(mpc, eps_a, c5.EPS_R, slots=4096)
    tree = Tree(NodeData(vertices=roots[2].copy()))          # ~1.5 k regions, 2 s
    st = frontier.grow_cells(nat, tree)
    nat.close()
    assert st['slow_path_cells'] == 0 and st['regions'] > 1000
    nodes = list(tree.walk())
    has = {loc: hasattr(nd.data, 'commutation') for nd, loc in nodes}
    rng = np.random.default_rng(0)
    ecc_splits = [i for i, (nd, loc) in enumerate(nodes) if not nd.is_leaf() and not has[loc]]
    own_leaves = [i for i, (nd, loc) in enumerate(nodes) if nd.is_leaf() and
                  (loc == '' or not has[loc[:-1]])]
    picks = list(rng.choice(ecc_splits, 3, replace=False)) + \
        list(rng.choice(own_leaves, 2, replace=False))
    seq_of = lambda d: tuple(int(i) for i in np.asarray(d).reshape(mpc.N, mpc.delta_size).argmax(1))
    nv, p = roots.shape[1], roots.shape[2]
    for k, i in enumerate(picks):
        nd, loc = nodes[i]
        d = nd.data
        leaf = nd.is_leaf()
        job = (k, 1 if leaf else 0, np.asarray(d.vertices, dtype=np.float64),
               np.array(seq_of(d.commutation) if leaf else [-1] * mpc.N),
               np.asarray(d.vertex_costs) if leaf else np.full(nv, np.nan), bool(leaf),
               np.full((2, nv, p), np.nan) if leaf else
               np.array([np.asarray(c.data.vertices) for c in (nd.left, nd.right)]),
               np.full((2, mpc.N), -1), eps_a, c5.EPS_R)
        res = c5._check_one(job)
        assert res['ok'] and not res['routed'], (loc, res['notes'])
        if leaf:
            assert res['t_max'] < 0. and res['max_cost_diff'] <= 1e-7
