"""
The tangent-plane bound that lets the device close leaves without a suboptimality-test LP
(DESIGN.md section 3.3c), checked on the CPU with HiGHS: it is an UPPER bound of the true
optimum t* on every node (so a negative bound can only close what the LP would close), the
pairwise form is at least as tight as the single-plane form and no tighter than the exact
cutting-plane LP, and it closes most of the leaves a partition closes.
"""

import numpy as np

from tests import helpers
from explicit_hybrid_mpc_amd import examples
from oracle import cut_bound as cb
from oracle.oracle_cpu import OracleCPU
from oracle.partition_cpu import PartitionCPU


def test_bound_is_valid_and_closes_most_leaves():
    mpc = examples.linear_mpc(0)
    eps_r = 0.05
    eps_a = helpers.eps_a_rule(mpc, 0.15)
    roots, locs = helpers.roots_of(mpc)
    orc = OracleCPU(mpc, eps_a, eps_r)
    orc.memoize = True
    part = PartitionCPU(orc, max_nodes=700)
    part.run(roots, locs, 'ecc')
    nodes = [v for v in part.nodes.values() if v['vertex_costs'] is not None and
             (v['is_epsilon_suboptimal'] or not v['leaf'])]
    rng = np.random.default_rng(0)
    rng.shuffle(nodes)
    model = orc.models[0]
    n_closed = n_cert = n_open = 0
    for nd in nodes[:60]:
        R, V = nd['vertices'], nd['vertex_costs']
        g = np.array([cb.vertex_gradient(model, v)[1] for v in R])
        rows = cb.rows_at_vertices(R, V, g, eps_a, eps_r)
        b1, b2 = cb.bound_single(rows), cb.bound_pairs(rows)
        b_all = cb.bound_all_cuts(R, V, g, eps_a, eps_r)
        t_star, _ = orc.slack(R, V, 0)
        tol = 1e-8 * (1 + abs(t_star))
        assert b1 >= b2 - tol and b2 >= b_all - tol and b_all >= t_star - tol
        if nd['is_epsilon_suboptimal']:
            assert t_star < 0
            n_closed += 1
            n_cert += b2 < 0
        else:
            assert t_star >= 0 and b2 >= 0          # an open node is never closed by the bound
            n_open += 1
    assert n_closed >= 10 and n_open >= 10
    assert n_cert >= 0.7 * n_closed


def test_bound_is_valid_for_quadratic_costs():
    """Convexity is all the bound needs: same check with the quadratic cost (QCQP optimum)."""
    mpc = examples.linear_mpc(0, cost='quadratic')
    eps_r = 0.02
    eps_a = helpers.eps_a_rule(mpc, 0.05)
    orc = OracleCPU(mpc, eps_a, eps_r)
    model = orc.models[0]
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(3)
    n_neg = n_pos = 0
    for trial in range(14):
        scale = [0.03, 0.1, 0.3][trial % 3]
        R = rng.uniform(-0.6, 0.6, 4) * half + scale * rng.uniform(-1, 1, (5, 4)) * half
        VG = [cb.vertex_gradient_quadratic(model, v) for v in R]
        V = np.array([v for v, _ in VG])
        g = np.array([gr for _, gr in VG])
        # the gradient really is the gradient: finite difference along a random direction
        if trial < 3:
            d = rng.normal(size=4) * 1e-5 * half
            Vp = cb.vertex_gradient_quadratic(model, R[0] + d)[0]
            Vm = cb.vertex_gradient_quadratic(model, R[0] - d)[0]
            assert abs((Vp - Vm) / 2 - g[0] @ d) <= 1e-6 * (abs(g[0] @ d) + 1e-12) + 1e-12
        rows = cb.rows_at_vertices(R, V, g, eps_a, eps_r)
        b1, b2 = cb.bound_single(rows), cb.bound_pairs(rows)
        t_star, _ = orc.slack(R, V, 0)
        tol = 1e-7 * (1 + abs(t_star))
        assert b1 >= b2 - tol and b2 >= t_star - tol
        n_neg += t_star < 0
        n_pos += t_star >= 0
        if t_star >= 0:
            assert b2 >= -tol
    assert n_neg >= 2 and n_pos >= 2
