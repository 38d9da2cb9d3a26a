"""
CPU oracle self-consistency (no GPU): the HiGHS restatement of lib/oracle.py on the
uncondensed model against the condensed canonical LPs solved by the independent numpy
interior-point path, plus invariants of the CPU partition (lib/worker.py:241-417).
"""

import numpy as np
import pytest

from explicit_hybrid_mpc_amd import examples
from oracle import ipm_numpy as ipm
from oracle.oracle_cpu import OracleCPU
from oracle.partition_cpu import PartitionCPU
from tests import helpers

RTOL = 1e-7


@pytest.mark.parametrize('kind,seed', [('di', 0), ('lin', 0), ('lin', 3), ('pwa', 0)])
def test_condensed_ipm_matches_uncondensed_highs(kind, seed):
    mpc = helpers.make_instance(kind, seed)
    can = mpc.compile()
    orc = OracleCPU(mpc, 1e-2, 1e-2)
    rng = np.random.default_rng(11)
    half = examples.theta_box(mpc)
    n_checked = 0
    for _ in range(6):
        theta = rng.uniform(-1, 1, half.size) * half
        for d in range(0, can.n_delta, max(1, can.n_delta // 4)):
            ok, u, J = orc._point(theta, d)
            c, A, b = ipm.assemble_feasibility(can, d, theta)
            tau = ipm.solve_lp(c, A, b).obj
            assert (tau <= 1e-8) == ok
            if ok:
                r = ipm.solve_lp(*ipm.assemble_point(can, d, theta))
                assert r.status == 0
                assert abs(r.obj - J) <= RTOL * (1 + abs(J))
                n_checked += 1
    assert n_checked >= 6


def test_slack_and_min_simplex_forms_agree():
    mpc = helpers.make_instance('lin', 1)
    can = mpc.compile()
    orc = OracleCPU(mpc, 0.05, 0.1)
    rng = np.random.default_rng(12)
    R = helpers.random_simplices(mpc, rng, 8)
    for Rk in R:
        Vb = np.array([orc.P_theta_delta(v, orc.deltas[0])[1] for v in Rk])
        t_ref, alpha = orc.slack(Rk, Vb, 0)
        r = ipm.solve_lp(*ipm.assemble_bar_E(can, 0, Rk, Vb, orc.eps_a, orc.eps_r))
        assert abs(-r.obj - t_ref) <= RTOL * (1 + abs(t_ref))
        res = orc._solve(orc.models[0].lp_min_over_simplex(Rk))
        r2 = ipm.solve_lp(*ipm.assemble_min_simplex(can, 0, Rk))
        assert abs(r2.obj - res.fun) <= RTOL * (1 + abs(res.fun))
        assert abs(alpha.sum() - 1) < 1e-9


def test_oracle_return_conventions():
    """Shapes / None conventions of lib/oracle.py:104-173, 175-218, 311-414."""
    mpc = helpers.make_instance('pwa', 0)
    orc = OracleCPU(mpc, 0.1, 0.5)
    half = examples.theta_box(mpc)
    u, delta, J, t = orc.P_theta(0.3 * half)
    assert u.shape == (2,) and delta.shape == (10,) and J >= 0 and t >= 0
    assert set(np.unique(delta)) <= {0., 1.} and delta.reshape(5, 2).sum(axis=1).tolist() == [1.] * 5
    assert orc.P_theta(0.3 * half, check_feasibility=True) is True
    assert orc.P_theta(50 * half, check_feasibility=True) is False
    u2, J2, _ = orc.P_theta_delta(0.3 * half, delta)
    assert abs(J2 - J) <= 1e-9 * (1 + J)
    # a simplex far on the positive side of the switching surface: mode-1-first sequences
    # are infeasible there, V_R must skip them
    R = 0.5 * half + 0.05 * half * np.vstack([np.zeros(4), np.eye(4)])
    d_feas, vx = orc.V_R(R)
    assert d_feas is not None and len(vx) == 5 and len(vx[0]) == 3
    assert d_feas[0] == 1.          # step 0 in mode 0 (x_1 >= -overlap)
    out = orc.bar_D_delta_R(R, np.array([v[1] for v in vx]), d_feas)
    assert len(out) == 4


def test_cpu_partition_invariants():
    mpc = helpers.make_instance('lin', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.5)
    orc = OracleCPU(mpc, eps_a, 1.0)
    roots, locs = helpers.roots_of(mpc)
    part = PartitionCPU(orc)
    nodes = part.run(roots, locs, 'ecc')
    leaves = part.leaves()
    assert all(v['is_epsilon_suboptimal'] for v in leaves.values())
    total = np.prod(2 * examples.theta_box(mpc))
    assert abs(part.volume_closed - total) <= 1e-9 * total
    # internal nodes have exactly the two children, children share all but one vertex
    for loc, nd in nodes.items():
        if not nd['leaf']:
            a, b = nodes[loc + '0'], nodes[loc + '1']
            same_a = (a['vertices'] == nd['vertices']).all(axis=1).sum()
            same_b = (b['vertices'] == nd['vertices']).all(axis=1).sum()
            assert same_a == same_b == nd['vertices'].shape[0] - 1
    # every closed leaf passes an independent re-check at random interior points
    rng = np.random.default_rng(3)
    for loc in list(leaves)[::9]:
        nd = leaves[loc]
        al = rng.dirichlet(np.ones(5))
        theta = al @ nd['vertices']
        J = orc.P_theta(theta)[2]
        gap = al @ nd['vertex_costs'] - J
        assert gap <= max(orc.eps_a, orc.eps_r * J) + 1e-7
    # bounded run + resume reproduces the same tree
    part2 = PartitionCPU(OracleCPU(mpc, eps_a, 1.0), max_nodes=40)
    part2.run(roots, locs, 'ecc')
    assert part2.truncated
    part2.max_nodes = None
    part2.resume()
    assert set(part2.nodes) == set(nodes)
