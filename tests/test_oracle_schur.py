"""
The block elimination of csrc/ehm_ipm2.h (round 4: the epigraph columns of the z-block leave the
Newton system) stated in numpy, oracle/schur_numpy.py, against the dense interior-point method of
oracle/ipm_numpy.py: the reduction changes how the Newton systems are SOLVED, not the systems --
same optima, same iteration counts, for every kind of LP the oracles pose (point, phase one,
suboptimality test with its two dense rows).  CPU only.
"""
import numpy as np
import pytest

from explicit_hybrid_mpc_amd import examples
from oracle import ipm_numpy as ip
from oracle import schur_numpy as sc


def _law(name):
    mpc = getattr(examples, name)()
    return mpc, mpc.compile()


# integrator_chain_mpc = BASELINE configs[3] (n_x = 6, n_u = 3, N = 10: the wide kernels, which
# still factorise all 57 columns -- DESIGN.md section 8 item 2): 30 inputs stay, 20 epigraphs go
@pytest.mark.parametrize('name, nd0_expected', [('linear_mpc', 10), ('pwa_mpc', 10),
                                                ('integrator_chain_mpc', 30)])
def test_the_epigraph_columns_are_the_singleton_tail(name, nd0_expected):
    """lib/mpc_library.py:530-560 appends the epigraph variables ex_k, eu_k to the inputs: every
    MPC row holds at most one of them, in every commutation."""
    _, can = _law(name)
    assert sc.singleton_tail(can.G) == nd0_expected
    G = np.asarray(can.G)
    tail = np.abs(G[:, :, nd0_expected:]) > 0
    assert (tail.sum(axis=2) <= 1).all()
    # one column more is no singleton block any longer
    wider = np.abs(G[:, :, nd0_expected - 1:]) > 0
    assert (wider.sum(axis=2) > 1).any()


def test_no_tail_without_epigraphs():
    """A dense block has no such range: the solver keeps every column."""
    rng = np.random.default_rng(3)
    G = rng.standard_normal((2, 30, 6))
    assert sc.singleton_tail(G) == 6


def test_reduced_newton_step_equals_the_dense_one():
    """One Newton system with random positive weights: reduction against numpy's dense solve."""
    _, can = _law('linear_mpc')
    rng = np.random.default_rng(1)
    n, p = can.n, can.p
    R = 0.2 * rng.standard_normal((p + 1, p))
    c, A, b = ip.assemble_bar_E(can, 0, R, rng.random(p + 1) + 1., 0.05, 0.01)
    m = A.shape[0]
    for trial in range(20):
        d = np.exp(rng.uniform(-18., 18., size=m))      # lambda / s over 16 orders of magnitude
        r = rng.standard_normal(A.shape[1])
        M = A.T @ (d[:, None] * A)
        cols_E = np.arange(10, n)
        cols_D = np.array([j for j in range(A.shape[1]) if j < 10 or j >= n])
        F = sc.reduced_factor(A[:m - 2], d[:m - 2], A[m - 2:], d[m - 2:], cols_D, cols_E)
        x = sc.reduced_solve(F, r)
        # compare in the metric of the system: residual relative to the right-hand side
        assert np.linalg.norm(M @ x - r) <= 1e-7 * np.linalg.norm(r) * np.sqrt(np.linalg.cond(M)) \
            or np.allclose(x, np.linalg.solve(M, r), rtol=1e-6)


@pytest.mark.parametrize('name, nd0, trials, least', [('linear_mpc', 10, 12, 60),
                                                      ('integrator_chain_mpc', 30, 3, 22)])
def test_whole_solves_same_optimum_same_iterations(name, nd0, trials, least):
    mpc, can = _law(name)
    rng = np.random.default_rng(0)
    n, p = can.n, can.p
    half = examples.theta_box(mpc)
    worst, mismatches, total = 0., 0, 0
    for trial in range(trials):
        R = (0.6 * rng.random((p + 1, p)) - 0.3) * half
        Vb = []
        for v in R:
            c, A, b = ip.assemble_point(can, 0, v)
            o = ip.solve_lp(c, A, b, step_frac=0.999)
            x2, obj2, it2, cv = sc.solve_lp_reduced(c, A, b, 0, range(nd0, n))
            assert cv and o.status == 0
            worst = max(worst, abs(obj2 - o.obj) / (1. + abs(o.obj)))
            mismatches += int(it2 != o.iters)
            total += 1
            Vb.append(o.obj)
        for (c, A, b), k in ((ip.assemble_bar_E(can, 0, R, Vb, 0.05, 0.01), 2),
                             (ip.assemble_feasibility(can, 0, R[0]), 1),
                             (ip.assemble_min_simplex(can, 0, R), 0)):
            o = ip.solve_lp(c, A, b, step_frac=0.999)
            x2, obj2, it2, cv = sc.solve_lp_reduced(c, A, b, k, range(nd0, n))
            if o.status != 0 or not cv:
                continue
            worst = max(worst, abs(obj2 - o.obj) / (1. + abs(o.obj)))
            mismatches += int(it2 != o.iters)
            total += 1
    assert total >= least
    assert worst <= 1e-11
    assert mismatches == 0
