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
  adopted at the cell it is the lexicographic minimum of V_R's MILP; bar_E's MILP has max t < 0;
* three lcss splits (round 6): bar_E's MILP has max t >= 0 (the largest over big-M 50 / 10 / 200 -- any
  run's point is feasible for the reference's problem, so that is a certified lower bound), the
  children are the longest-edge bisection and hold the node's commutation or bar_D's optimum --
  compared tie-aware: a child's sequence is also accepted where it is feasible at every vertex
  and its own fixed-sequence slack is within the canonical tie tolerance of the MILP's maximum.
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
    lcss_splits = [i for i, (nd, loc) in enumerate(nodes) if not nd.is_leaf() and has[loc]]
    assert len(lcss_splits) >= 3
    picks = list(rng.choice(ecc_splits, 3, replace=False)) + \
        list(rng.choice(own_leaves, 2, replace=False)) + \
        list(rng.choice(lcss_splits, 3, replace=False))     # (~3 min of HiGHS per lcss split)
    seq_of = lambda d: tuple(int(i) for i in np.asarray(d).reshape(mpc.N, mpc.delta_size).argmax(1))
    nv, p = roots.shape[1], roots.shape[2]
    jobs, meta = [], []
    for k, i in enumerate(picks):
        nd, loc = nodes[i]
        d = nd.data
        leaf = nd.is_leaf()
        kind = 1 if leaf else (2 if has[loc] else 0)
        kids_seq = np.full((2, mpc.N), -1)
        if kind == 2:
            kids_seq = np.array([seq_of(c.data.commutation) for c in (nd.left, nd.right)])
        job = (k, kind, np.asarray(d.vertices, dtype=np.float64),
               np.array(seq_of(d.commutation) if has[loc] else [-1] * mpc.N),
               np.asarray(d.vertex_costs) if has[loc] else np.full(nv, np.nan),
               bool(has[loc] and (loc == '' or not has[loc[:-1]])),
               np.full((2, nv, p), np.nan) if leaf else
               np.array([np.asarray(c.data.vertices) for c in (nd.left, nd.right)]),
               kids_seq, eps_a, c5.EPS_R)
        jobs.append(job)
        meta.append((loc, kind, leaf))
    # the checks are independent CPU work (HiGHS; the device handle is closed): one process each,
    # the lcss splits -- minutes of branch and bound each -- first
    import multiprocessing as mp
    order = sorted(range(len(jobs)), key=lambda k: -jobs[k][1])
    with mp.get_context('fork').Pool(min(len(jobs), os.cpu_count() or 1)) as pool:
        results = pool.map(c5._check_one, [jobs[k] for k in order], chunksize=1)
    for k, res in zip(order, results):
        loc, kind, leaf = meta[k]
        assert res['ok'] and not res['routed'], (loc, kind, res['notes'])
        if leaf:
            assert res['t_max'] < 0. and res['max_cost_diff'] <= 1e-7
        if kind == 2:
            assert res['t_max'] >= 0. and res['max_cost_diff'] <= 1e-7
