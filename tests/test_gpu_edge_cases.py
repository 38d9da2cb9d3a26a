"""
Edge cases of the C-ABI on the device: empty batches, size limits, capacity and depth limits,
infeasible parameter sets (the reference's ``RuntimeError`` at lib/worker.py:266), infeasible
single solves (``None`` returns, lib/oracle.py:136-139, 170-173), degenerate simplices and
non-finite input.  The reference has no tests of its own; these pin the error behaviour
documented in include/ehmpc.h and INTEGRATION.md.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lin():
    from explicit_hybrid_mpc_amd import engine
    mpc = helpers.make_instance('lin', 0)
    gp = engine.GpuProblem(mpc.compile(), 0.05, 0.1)
    yield mpc, gp
    gp.close()


@pytest.mark.parametrize('gen', [2, 1])
def test_empty_batches(lin, gen):
    mpc, gp = lin
    gp.set_solver(gen)
    p, n_u = 4, 2
    J, u0, st, it = gp.solve_ptd(np.zeros((0, p)))
    assert J.shape == (0,) and u0.shape == (0, n_u) and st.shape == (0,)
    feas, tau = gp.feasible_ptd(np.zeros((0, p)))
    assert feas.shape == (0,)
    J, u0, didx = gp.solve_pt(np.zeros((0, p)))
    assert didx.shape == (0,)
    R0, V0 = np.zeros((0, p + 1, p)), np.zeros((0, p + 1))
    assert gp.v_r(R0)[0].shape == (0,)
    assert gp.slack(R0, V0)[0].shape == (0,)
    assert gp.bar_e(R0, V0)[0].shape == (0,)
    assert gp.min_simplex(R0)[0].shape == (0,)
    assert gp.bar_d(R0, V0, np.ones((0, 5)))[0].shape == (0,)
    gp.set_solver(2)


def test_empty_geometry_batches():
    from explicit_hybrid_mpc_amd import engine
    S1, S2, idx = engine.split_batch(np.zeros((0, 5, 4)))
    assert S1.shape == (0, 5, 4) and idx.shape[0] == 0
    assert engine.volume_batch(np.zeros((0, 5, 4))).shape == (0,)


def test_size_limits_are_refused():
    from explicit_hybrid_mpc_amd import engine
    from explicit_hybrid_mpc_amd._capi import EhmError, EHM_E_INVALID
    from explicit_hybrid_mpc_amd.mpc_library import CanonicalLP

    def can(n, m, p):
        return CanonicalLP(np.zeros((1, m, n)), np.ones((1, m)), np.zeros((1, m, p)),
                           np.ones(n), np.ones((1, 1)), 1, 1, 1)
    for n, m, p in ((70, 10, 2), (10, 1100, 2), (4, 10, 9)):
        with pytest.raises(EhmError) as e:
            engine.GpuProblem(can(n, m, p), 0.1, 0.1)
        assert e.value.code == EHM_E_INVALID
    # a quadratic cost on a problem beyond the wave-local kernels
    big = can(40, 100, 4)
    big.quadratic = True
    big.H, big.F, big.f0 = np.eye(40)[None], np.zeros((1, 40, 4)), np.zeros((1, 40))
    big.C, big.c1, big.c0 = np.zeros((1, 4, 4)), np.zeros((1, 4)), np.zeros(1)
    with pytest.raises(EhmError) as e:
        engine.GpuProblem(big, 0.1, 0.1)
    assert e.value.code == EHM_E_INVALID


def test_not_an_admissible_commutation(lin):
    from explicit_hybrid_mpc_amd._capi import EhmError, EHM_E_INVALID
    mpc, gp = lin
    with pytest.raises(EhmError) as e:
        gp.solve_ptd(np.zeros((1, 4)), np.zeros(5))      # all-zero is no mode sequence
    assert e.value.code == EHM_E_INVALID


def test_infeasible_and_non_finite_parameters(lin):
    from explicit_hybrid_mpc_amd import examples
    mpc, gp = lin
    half = examples.theta_box(mpc)
    theta = np.array([0.2 * half, 50 * half, np.full(4, np.nan), np.full(4, np.inf)])
    feas, tau = gp.feasible_ptd(theta)
    assert feas.tolist() == [True, False, False, False]
    J, u0, st, it = gp.solve_ptd(theta)
    assert st[0] == 0 and (st[1:] != 0).all()            # stalled, never a hang or a crash
    J, u0, didx = gp.solve_pt(theta)
    assert didx[0] == 0 and (didx[1:] == -1).all() and np.isinf(J[1:]).all()


def test_infeasible_parameter_set_raises_like_the_reference(lin):
    """lib/worker.py:266: ecc raises RuntimeError when Theta contains infeasible regions."""
    from explicit_hybrid_mpc_amd import examples
    mpc, gp = lin
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    V = examples.box_vertices(30 * examples.theta_box(mpc))
    roots, _ = ehm_tools.delaunay_roots(V)
    from explicit_hybrid_mpc_amd._capi import EHM_E_INFEASIBLE
    with pytest.raises(RuntimeError, match='Theta contains infeasible regions') as e:
        gp.partition(roots, action='ecc')
    assert e.value.code == EHM_E_INFEASIBLE


def test_capacity_and_depth_limits(lin):
    from explicit_hybrid_mpc_amd import examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from explicit_hybrid_mpc_amd._capi import EhmError, EHM_E_CAPACITY
    mpc, gp = lin
    V = examples.box_vertices(examples.theta_box(mpc))
    roots, _ = ehm_tools.delaunay_roots(V)
    eps_a = float(np.max(gp.solve_pt(0.25 * V)[0]))
    gp.set_eps(eps_a, 0.05)
    full = gp.partition(roots, action='ecc')
    assert full.info['truncated'] == 0 and full.n_nodes > 1500
    for eng in (1, 0):      # persistent frontier kernel / level-synchronous sweeps
        with pytest.raises(EhmError) as e:
            gp.partition(roots, action='ecc', max_nodes=1000, engine=eng)
        assert e.value.code == EHM_E_CAPACITY
        cut = gp.partition(roots, action='ecc', max_depth=3, engine=eng)
        assert cut.info['truncated'] == 1 and cut.info['max_depth'] <= 3
        assert cut.n_nodes < full.n_nodes
        # what was grown is the top of the full tree
        n = cut.n_nodes
        assert np.array_equal(cut.vertices, full.vertices[:n])
    gp.set_eps(0.05, 0.1)


def test_degenerate_simplices():
    from explicit_hybrid_mpc_amd import engine
    R = np.zeros((2, 5, 4))
    R[0] = np.eye(5, 4)
    R[1] = np.eye(5, 4)
    R[1, 4] = R[1, 3]                                     # repeated vertex: zero volume
    vol = engine.volume_batch(R)
    assert vol[0] > 0 and vol[1] == 0.0
    S1, S2, idx = engine.split_batch(R)
    assert idx.shape == (2, 2) and (idx[:, 0] < idx[:, 1]).all()
