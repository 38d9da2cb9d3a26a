"""
GPU parity of the batched oracles (through the C-ABI) against the CPU oracle.
Tolerance on optimal costs: 1e-7 relative to (1+|J|) (north_star: "within a stated FP
tolerance"); geometry is bit-exact.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

RTOL = 1e-7


@pytest.fixture(scope='module')
def lin():
    from explicit_hybrid_mpc_amd import engine
    from oracle.oracle_cpu import OracleCPU
    mpc = helpers.make_instance('lin', 0)
    eps_a, eps_r = 0.05, 0.1
    gp = engine.GpuProblem(mpc.compile(), eps_a, eps_r)
    orc = OracleCPU(mpc, eps_a, eps_r)
    yield mpc, gp, orc
    gp.close()


def test_split_matches_reference_golden(golden_geometry):
    from explicit_hybrid_mpc_amd import engine
    z = golden_geometry
    for key in [k for k in z.files if k.endswith('_R')]:
        pre = key[:-2]
        S1, S2, ij = engine.split_batch(z[key])
        assert np.array_equal(ij, z[pre + '_ij']), pre
        assert np.array_equal(S1, z[pre + '_S1']), pre
        assert np.array_equal(S2, z[pre + '_S2']), pre
        vol = engine.volume_batch(z[key])
        assert np.allclose(vol, z[pre + '_vol'], rtol=1e-12, atol=0.)


def test_ptd_matches_oracle(lin):
    from explicit_hybrid_mpc_amd import examples
    mpc, gp, orc = lin
    rng = np.random.default_rng(7)
    half = examples.theta_box(mpc)
    theta = np.vstack([examples.box_vertices(half), rng.uniform(-1, 1, (184, 4)) * half])
    J, u0, status, iters = gp.solve_ptd(theta, orc.deltas[0])
    assert (status == 0).all()
    can = mpc.compile()
    for k in range(theta.shape[0]):
        u_ref, J_ref, _ = orc.P_theta_delta(theta[k], orc.deltas[0])
        assert abs(J[k] - J_ref) <= RTOL * (1 + abs(J_ref)), (k, J[k], J_ref)
    assert iters.max() <= 25
    # u0 belongs to an optimal solution: re-solve with u0 fixed and compare the optimum
    from scipy.optimize import linprog
    for k in range(0, theta.shape[0], 10):
        h = can.w[0] + can.S[0] @ theta[k]
        Aeq = np.zeros((can.n_u, can.n))
        Aeq[:, :can.n_u] = np.eye(can.n_u)
        res = linprog(can.c, A_ub=can.G[0], b_ub=h + 1e-9, A_eq=Aeq, b_eq=u0[k],
                      bounds=(None, None), method='highs')
        assert res.status == 0
        assert abs(res.fun - J[k]) <= 1e-6 * (1 + abs(J[k]))


def test_feasibility_form(lin):
    from explicit_hybrid_mpc_amd import examples
    mpc, gp, orc = lin
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(8)
    theta = rng.uniform(-1, 1, (64, 4)) * half * 2.5      # in- and outside the feasible set
    feas, tau = gp.feasible_ptd(theta, orc.deltas[0])
    ref = np.array([orc.P_theta_delta(t, orc.deltas[0], check_feasibility=True)
                    for t in theta])
    assert feas.any() and (~feas).any()
    assert np.array_equal(feas, ref)


def test_slack_and_min_simplex_match_oracle(lin):
    mpc, gp, orc = lin
    rng = np.random.default_rng(9)
    R = helpers.random_simplices(mpc, rng, 60)
    Vbar = np.array([[orc.P_theta_delta(v, orc.deltas[0])[1] for v in Rk] for Rk in R])
    t, alpha, status = gp.slack(R, Vbar, orc.deltas[0])
    assert (status == 0).all()
    Jmin, st2 = gp.min_simplex(R, orc.deltas[0])
    assert (st2 == 0).all()
    for k in range(R.shape[0]):
        t_ref, a_ref = orc.slack(R[k], Vbar[k], 0)
        assert abs(t[k] - t_ref) <= RTOL * (1 + abs(t_ref)), (k, t[k], t_ref)
        assert abs(alpha[k].sum() - 1) < 1e-9 and (alpha[k] > -1e-9).all()
        res = orc._solve(orc.models[0].lp_min_over_simplex(R[k]))
        assert abs(Jmin[k] - res.fun) <= RTOL * (1 + abs(res.fun))
    assert (t > 0).any() and (t < 0).any()
