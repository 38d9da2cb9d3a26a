"""
Quadratic-cost half of the CPU oracle (every MPC law of the reference has a
``cvx.quad_form`` cost, lib/mpc_library.py:180-183, :515-517).

* KNOWN ANSWERS OF THE REFERENCE: ``lib/post_process.py:484-485`` lists the absolute-error
  tolerances of its cwh_z runs, i.e. (lib/examples.py:42-45) the largest ``P_theta`` optimal
  cost over the ``abs_frac``-scaled vertices of the partitioned box, for the job parameters
  of ``make_jobs.sh:60-66`` (N = 4; abs_frac 0.5 and 0.25; the other three entries of the list
  correspond to abs_frac 0.1, 0.03, 0.01 -- oracle/satellite_cpu.py).  The restated
  ``SatelliteZ`` law + mixed-integer QP oracle must reproduce all five.  Tolerance 1e-7
  ABSOLUTE (observed <= 6.7e-8; relative 1.4e-6 .. 5e-5 as the costs shrink): the size of
  MOSEK's own feasibility / optimality tolerances, which is what separates the reference's
  printed values from the exact optima.
* the product's condensed canonical QP (``mpc_library.SatelliteZ`` / quadratic ``PWAMPC``)
  and the kernel's algorithm in numpy (``oracle/ipm_numpy.py::solve_cp``) against the
  uncondensed oracle models solved by ``oracle/qp_numpy.py``;
* ``oracle/qp_numpy.py`` against SciPy's SLSQP as a second opinion.
"""

import numpy as np
import pytest
from scipy.optimize import minimize

from explicit_hybrid_mpc_amd import examples
from explicit_hybrid_mpc_amd.mpc_library import SatelliteZ
from oracle import ipm_numpy as ip
from oracle import qp_numpy
from oracle.oracle_cpu import OracleCPU
from oracle.satellite_cpu import (SatelliteZCPU, KNOWN_EPS_A, KNOWN_EPS_A_INFERRED,
                                  KNOWN_EPS_A_ABS_TOL)

RTOL = 1e-7


@pytest.fixture(scope='module')
def sat():
    return SatelliteZ(4), SatelliteZCPU(4)


def test_known_answers_fixture_matches_the_constants():
    """tests/golden/known_answers.json is what make_known_answers.py read from the reference."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                        'known_answers.json')
    runs = json.load(open(path))['runs']
    assert len(runs) == 5 and all(r['example'] == 'cwh_z' and r['N'] == 4 for r in runs)
    pinned = {(r['N'], r['abs_frac']): r['eps_a'] for r in runs if r['abs_frac'] is not None}
    assert pinned == KNOWN_EPS_A
    assert sorted(r['eps_a'] for r in runs if r['abs_frac'] is None) == \
        sorted(KNOWN_EPS_A_INFERRED.values())


@pytest.mark.parametrize('key', sorted(KNOWN_EPS_A) + sorted(KNOWN_EPS_A_INFERRED))
def test_reference_known_eps_a(key):
    N, abs_frac = key
    known = {**KNOWN_EPS_A, **KNOWN_EPS_A_INFERRED}[key]
    cpu = SatelliteZCPU(N)
    orc = OracleCPU(cpu, 1., 1.)                       # lib/examples.py:43
    eps_a = max(orc.P_theta(theta=abs_frac * v)[2] for v in cpu.box_vertices)
    assert abs(eps_a - known) <= KNOWN_EPS_A_ABS_TOL
    assert abs(eps_a - known) <= 1e-4 * known
    # far out every step fires (against the position error); close to the origin coasting
    # (no input piece active at any step) is optimal
    u, delta, J, _ = orc.P_theta(theta=abs_frac * cpu.box_vertices[0])
    assert delta.sum() == (N if abs_frac >= 0.25 else 0 if abs_frac <= 0.01 else delta.sum())


def test_commutation_layout_and_sizes(sat):
    mpc, cpu = sat
    can = mpc.compile()
    assert (can.n, can.m, can.p, can.n_u, can.n_delta) == (12, 40, 2, 1, 81)
    assert can.quadratic and can.deltas.shape == (81, 8)
    assert mpc.mode_sequences() == cpu.mode_sequences()
    for s in ((0, 0, 0, 0), (1, 2, 0, 1)):
        assert np.array_equal(mpc.sequence_to_delta(s), cpu.sequence_to_delta(s))
    d = mpc.sequence_to_delta((1, 2, 0, 1))           # lib/mpc_library.py:160 layout
    assert d.tolist() == [1, 0, 0, 1, 0, 0, 1, 0]
    assert np.all(can.deltas.reshape(81, 4, 2).sum(axis=2) <= 1)   # :211-215


def _feasible_everywhere(orc, R, d):
    return all(orc._point(v, d)[0] for v in R)


def test_condensed_model_and_kernel_algorithm_match_uncondensed_oracle(sat):
    mpc, cpu = sat
    can = mpc.compile()
    orc = OracleCPU(cpu, 0.005, 0.1)
    orc.memoize = True
    rng = np.random.default_rng(1)
    V = mpc.box_vertices()
    n_pt = n_sx = 0
    for trial in range(60):
        d = int(rng.integers(can.n_delta))
        th = V[rng.integers(4)] * rng.uniform(0, 0.9)
        ok, u, J = orc._point(th, d)
        if ok:
            out = ip.solve_cp(ip.assemble_point_quad(can, d, th))
            assert out.status == 0
            assert abs(out.obj - J) <= RTOL * (1 + abs(J))
            assert abs(out.x[0] - u[0]) <= 1e-7
            assert abs(can.cost_value(d, out.x, th) - J) <= RTOL * (1 + abs(J))
            n_pt += 1
        R = V[rng.integers(4)] * rng.uniform(0, 0.8) + \
            rng.normal(size=(3, 2)) * [0.02, 2e-4] * rng.uniform(0.05, 1)
        if not _feasible_everywhere(orc, R, d):
            continue
        Vb = np.array([orc._point(v, d)[2] for v in R])
        ref = orc._solve(orc.models[d].lp_min_over_simplex(R))
        out = ip.solve_cp(ip.assemble_min_simplex_quad(can, d, R))
        assert out.status == 0 and abs(out.obj - ref.fun) <= RTOL * (1 + abs(ref.fun))
        t_ref, alpha = orc.slack(R, Vb, d)
        out = ip.solve_cp(ip.assemble_bar_E_quad(can, d, R, Vb, orc.eps_a, orc.eps_r))
        assert out.status == 0 and abs(-out.obj - t_ref) <= RTOL * (1 + abs(t_ref))
        n_sx += 1
    assert n_pt >= 15 and n_sx >= 10


def test_quadratic_pwa_condensation():
    mpc = examples.pwa_mpc(seed=0, n_x=2, n_u=1, N=3, n_random=4, overlap=0.3,
                           cost='quadratic')
    can = mpc.compile()
    assert can.quadratic and can.n == 3 and can.n_delta == 8
    orc = OracleCPU(mpc, 0.05, 0.2)
    rng = np.random.default_rng(7)
    n = 0
    for trial in range(40):
        d = int(rng.integers(8))
        R = rng.uniform(-0.3, 0.3, 2) + rng.uniform(-0.1, 0.1, (3, 2))
        if not _feasible_everywhere(orc, R, d):
            continue
        Vb = np.array([orc._point(v, d)[2] for v in R])
        for v, J in zip(R, Vb):
            out = ip.solve_cp(ip.assemble_point_quad(can, d, v))
            assert out.status == 0 and abs(out.obj - J) <= RTOL * (1 + abs(J))
        t_ref, _ = orc.slack(R, Vb, d)
        out = ip.solve_cp(ip.assemble_bar_E_quad(can, d, R, Vb, orc.eps_a, orc.eps_r))
        assert out.status == 0 and abs(-out.obj - t_ref) <= RTOL * (1 + abs(t_ref))
        n += 1
    assert n >= 8


def test_qp_solver_against_slsqp(sat):
    """Second opinion on oracle/qp_numpy.py (QP and QCQP), in scaled variables for SLSQP."""
    mpc, cpu = sat
    orc = OracleCPU(cpu, 0.005, 0.1)
    R = np.array([[0.02, 1e-4], [0.03, 2e-4], [0.025, -1e-4]])
    # a commutation with active input pieces that is feasible on all of R
    d = next(k for k in range(80, -1, -1) if _feasible_everywhere(orc, R, k))
    assert sum(cpu.mode_sequences()[d]) > 0
    mdl = orc.models[d]
    Vb = []
    for v in R:
        lp = mdl.lp_point(v)
        ours = qp_numpy.solve(lp['c'], lp['A_ub'], lp['b_ub'], lp['A_eq'], lp['b_eq'], P=lp['P'])
        assert ours.status == 0 and max(ours.res_p, ours.res_d, ours.gap) <= 1e-10
        Vb.append(ours.fun)
        sc = np.where(np.abs(ours.x) > 0, np.abs(ours.x), 1.)          # x = sc * y, y ~ +-1
        P_, A_, Ae_ = lp['P'], lp['A_ub'], lp['A_eq']
        res = minimize(lambda y: 0.5 * (sc * y) @ P_ @ (sc * y), ours.x / sc * 0.9,
                       jac=lambda y: sc * (P_ @ (sc * y)), method='SLSQP',
                       constraints=[dict(type='ineq', fun=lambda y: lp['b_ub'] - A_ @ (sc * y),
                                         jac=lambda y: -A_ * sc),
                                    dict(type='eq', fun=lambda y: Ae_ @ (sc * y) - lp['b_eq'],
                                         jac=lambda y: Ae_ * sc)],
                       options=dict(ftol=1e-15, maxiter=500))
        assert abs(res.fun - ours.fun) <= 1e-6 * (1 + abs(ours.fun))
    # the suboptimality-test QCQP: our optimum must be feasible and no SLSQP point may beat it
    lp = mdl.lp_bar_E(R, np.array(Vb), 0.005, 0.1)
    ours = qp_numpy.solve(lp['c'], lp['A_ub'], lp['b_ub'], lp['A_eq'], lp['b_eq'], quad=lp['quad'])
    assert ours.status == 0
    for (Pi, qi, ri) in lp['quad']:
        assert 0.5 * ours.x @ Pi @ ours.x + qi @ ours.x + ri <= 1e-9
    assert np.all(lp['A_ub'] @ ours.x - lp['b_ub'] <= 1e-9)
    assert np.allclose(lp['A_eq'] @ ours.x, lp['b_eq'], atol=1e-9)
    # KKT: stationarity of the Lagrangian with the returned multipliers
    lam_lin, lam_q = ours.lam[:lp['A_ub'].shape[0]], ours.lam[lp['A_ub'].shape[0]:]
    grad = lp['c'] + lp['A_ub'].T @ lam_lin + lp['A_eq'].T @ ours.nu
    for l, (Pi, qi, ri) in zip(lam_q, lp['quad']):
        grad = grad + l * (Pi @ ours.x + qi)
    scale = np.maximum(np.abs(lp['A_ub']).max(axis=0), np.abs(lp['A_eq']).max(axis=0))
    assert np.max(np.abs(grad) / np.maximum(scale, 1.)) <= 1e-8
    assert np.all(ours.lam >= 0)
