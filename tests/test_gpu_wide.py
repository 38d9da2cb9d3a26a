"""
The wide kernels (ehm_k3.hip: one workgroup per LP, constant block from L2, normal matrix on
the matrix cores) against the CPU oracle, through the C-ABI: BASELINE.json config 4
(n_x = 6, n_u = 3, N = 10, box constraints -- LPs of 50..57 columns and 360..369 rows) and a
smaller member of the same family whose normal matrix needs three tiles instead of four.
Same bars as the narrow kernels: optimal values within 1e-7 (1 + |value|), feasibility
verdicts equal, identical tree on a sub-forest the CPU oracle finishes in about a minute.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

RTOL = 1e-7


@pytest.fixture(scope='module', params=['chain', 'chain_small'])
def wide(request):
    from explicit_hybrid_mpc_amd import engine
    from oracle.oracle_cpu import OracleCPU
    mpc = helpers.make_instance(request.param, 0)
    eps_a, eps_r = 0.05, 0.1
    can = mpc.compile()
    assert can.n + can.p + 1 > 32          # beyond the wave-local kernels
    gp = engine.GpuProblem(can, eps_a, eps_r)
    orc = OracleCPU(mpc, eps_a, eps_r)
    yield mpc, gp, orc
    gp.close()


def test_selftest_covers_the_wide_instances():
    from explicit_hybrid_mpc_amd import engine
    out = engine.selftest()
    assert out.shape[0] == 31
    assert np.abs(out - np.array([1072., 99., 25., 1. / 3., -1.])).max() < 1e-13


def test_ptd_and_feasibility_match_oracle(wide):
    from explicit_hybrid_mpc_amd import examples
    mpc, gp, orc = wide
    rng = np.random.default_rng(11)
    half = examples.theta_box(mpc)
    p = half.size
    V = examples.box_vertices(half)
    theta = np.vstack([V[::max(1, len(V) // 8)], rng.uniform(-1, 1, (24, p)) * half])
    J, u0, status, iters = gp.solve_ptd(theta, orc.deltas[0])
    assert (status == 0).all()
    for k in range(theta.shape[0]):
        _, J_ref, _ = orc.P_theta_delta(theta[k], orc.deltas[0])
        assert abs(J[k] - J_ref) <= RTOL * (1 + abs(J_ref)), (k, J[k], J_ref)
    assert iters.max() <= 30
    theta2 = rng.uniform(-1, 1, (32, p)) * half * 2.0      # in- and outside the feasible set
    feas, tau = gp.feasible_ptd(theta2, orc.deltas[0])
    ref = np.array([orc.P_theta_delta(t, orc.deltas[0], check_feasibility=True)
                    for t in theta2])
    assert feas.any() and (~feas).any()
    assert np.array_equal(feas, ref)


def test_slack_and_min_simplex_match_oracle(wide):
    mpc, gp, orc = wide
    rng = np.random.default_rng(12)
    R = helpers.random_simplices(mpc, rng, 24)
    p = R.shape[2]
    Vbar, st = gp.solve_ptd(R.reshape(-1, p), orc.deltas[0])[0::2]
    assert (st == 0).all()
    Vbar = Vbar.reshape(R.shape[0], p + 1)
    t, alpha, status = gp.slack(R, Vbar, orc.deltas[0])
    assert (status == 0).all()
    Jmin, st2 = gp.min_simplex(R, orc.deltas[0])
    assert (st2 == 0).all()
    for k in range(R.shape[0]):
        t_ref, _ = orc.slack(R[k], Vbar[k], 0)
        assert abs(t[k] - t_ref) <= RTOL * (1 + abs(t_ref)), (k, t[k], t_ref)
        assert abs(alpha[k].sum() - 1) < 1e-9 and (alpha[k] > -1e-9).all()
        res = orc._solve(orc.models[0].lp_min_over_simplex(R[k]))
        assert abs(Jmin[k] - res.fun) <= RTOL * (1 + abs(res.fun))
    assert (t > 0).any() and (t < 0).any()


def test_sub_forest_identical_to_cpu_partition(wide):
    from explicit_hybrid_mpc_amd import examples
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    from oracle import geometry
    mpc, gp, _ = wide
    eps_a = helpers.eps_a_rule(mpc, 0.5)
    eps_r = 1.0
    roots, locs = helpers.roots_of(mpc)
    # every 40th root simplex of config 4 (652 roots), every 2nd of the small instance
    step = 40 if len(roots) > 100 else 2
    roots, locs = roots[::step], locs[::step]
    orc = OracleCPU(mpc, eps_a, eps_r)
    cpu = PartitionCPU(orc)
    cpu.run(roots, locs, 'ecc')
    gp.set_eps(eps_a, eps_r)
    for decide_full in (0, 1):
        gp.set_option('decide_full', decide_full)
        flat = gp.partition(np.array(roots), action='ecc')
        loc = flat.locations(locs)
        assert set(loc) == set(cpu.nodes.keys())
        for k, name in enumerate(loc):
            ref = cpu.nodes[name]
            assert np.array_equal(flat.vertices[k], ref['vertices']), name
            assert flat.is_leaf(k) == ref['leaf'], name
            assert bool(flat.flags[k] & 1) == ref['is_epsilon_suboptimal'], name
            assert np.allclose(flat.vertex_costs[k], ref['vertex_costs'], rtol=RTOL, atol=RTOL)
        total = sum(geometry.simplex_volume(R) for R in roots)
        assert abs(flat.info['volume_closed'] - total) <= 1e-9 * total
        assert flat.info['min_margin'] > 1e-6
    gp.set_option('decide_full', 0)


def test_wide_hybrid_batches_match_oracle():
    """
    Multi-commutation instance on the wide kernels (2-mode PWA, N = 8: 256 commutations, LPs of
    32..37 columns x 264..271 rows): instances arrive unsorted, the batch kernels walk the
    commutation segments.
    """
    from explicit_hybrid_mpc_amd import engine, examples
    from oracle.oracle_cpu import OracleCPU
    mpc = examples.pwa_mpc(seed=0, n_x=4, n_u=2, N=8)
    can = mpc.compile()
    assert can.n_delta == 256 and can.n + can.p + 1 > 32
    gp = engine.GpuProblem(can, 0.05, 0.1)
    orc = OracleCPU(mpc, 0.05, 0.1)
    rng = np.random.default_rng(21)
    half = examples.theta_box(mpc, scale=0.45)
    n = 96
    theta = rng.uniform(-1, 1, (n, can.p)) * half
    # commutations that stay in one mode for long stretches are the feasible ones
    pick = rng.integers(can.n_delta, size=n)
    pick[::3] = 0
    pick[1::3] = can.n_delta - 1
    delta = can.deltas[pick]
    feas, _ = gp.feasible_ptd(theta, delta)
    ref = np.array([orc.P_theta_delta(theta[k], delta[k], check_feasibility=True)
                    for k in range(n)])
    assert np.array_equal(feas, ref)
    assert feas.any() and (~feas).any()
    J, _, st, _ = gp.solve_ptd(theta[feas], delta[feas])
    assert (st == 0).all()
    for k, kk in enumerate(np.where(feas)[0]):
        J_ref = orc.P_theta_delta(theta[kk], delta[kk])[1]
        assert abs(J[k] - J_ref) <= RTOL * (1 + abs(J_ref)), (kk, J[k], J_ref)
    # P_theta: minimum over all 256 commutations per parameter
    Jp, _, didx = gp.solve_pt(theta[:6])
    for k in range(6):
        _, _, J_ref, _ = orc.P_theta(theta[k])
        assert didx[k] >= 0 and abs(Jp[k] - J_ref) <= RTOL * (1 + abs(J_ref))
    gp.close()
