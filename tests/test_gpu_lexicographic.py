"""
Lexicographic second stage of the vertex inputs (explicit_hybrid_mpc_amd/lexicographic.py: n_u
more LPs per vertex through ehm_problem_create / ehm_solve_ptd_batch) against the oracle's
statement of the same rule on the uncondensed model with HiGHS
(oracle/oracle_cpu.py: lexicographic_u0).  What the explicit law interpolates
(lib/mpc_library.py:786-789) is then a function of the parameter also where the LP optimum is a
face in u_0 -- the double integrator, which tests/test_gpu_partition.py has to leave out of its
comparison of the raw inputs.

Tolerance: 1e-6 absolute on the inputs (both sides solve the same LPs to ~1e-9; measured
agreement 1e-10 .. 1e-8).
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

ATOL = 1e-6


@pytest.mark.parametrize('kind', ['di', 'lin', 'pwa_small'])
def test_lexicographic_inputs_equal_the_oracle(kind):
    from explicit_hybrid_mpc_amd import engine, examples, lexicographic
    from oracle.oracle_cpu import OracleCPU
    mpc = helpers.make_instance(kind, 0)
    can = mpc.compile()
    gp = engine.GpuProblem(can, 1., 1.)
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(1)
    theta = (rng.random((300, can.p)) * 2 - 1) * half * 0.8
    J, _, d = gp.solve_pt(theta)
    keep = np.isfinite(J) & (d >= 0)
    theta, d = theta[keep], d[keep]
    J, u_ipm, status, _ = gp.solve_ptd(theta, can.deltas[d])
    assert np.all(status == 0) and len(theta) >= 100
    lex = lexicographic.LexicographicInputs(can)
    U = lex.solve(theta, d, J)
    lex.close()
    gp.close()
    orc = OracleCPU(mpc, 1., 1.)
    n = 120
    ref = np.array([orc.lexicographic_u0(t, can.deltas[k], tol=lexicographic.TOL)
                    for t, k in zip(theta[:n], d[:n])])
    assert np.allclose(U[:n], ref, rtol=0., atol=ATOL), np.abs(U[:n] - ref).max()
    # the rule picks the smallest first component of the optimal face: never above the
    # interior-point limit by more than the solvers' accuracy
    assert np.all(U[:, 0] <= u_ipm[:, 0] + 1e-7)
    if kind == 'di':
        # ... and on the double integrator the face really is wide somewhere
        assert np.abs(U - u_ipm).max() > 1e-2


def test_refined_tree_holds_the_oracles_lexicographic_inputs():
    """The double integrator (u_0 free on a face): the device partition, refined, carries at every
    vertex of every node the input the oracle's lexicographic rule gives there."""
    from explicit_hybrid_mpc_amd import engine, lexicographic
    from oracle.oracle_cpu import OracleCPU
    mpc = helpers.make_instance('di', 0)
    can = mpc.compile()
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    gp = engine.GpuProblem(can, eps_a, 0.1)
    flat = gp.partition(np.array(roots), action='ecc')
    gp.close()
    raw = flat.vertex_inputs.copy()
    lex = lexicographic.LexicographicInputs(can)
    pairs = lex.refine(flat)
    lex.close()
    has = flat.delta_idx >= 0
    assert 0 < pairs < int(has.sum()) * (can.p + 1)          # shared vertices are solved once
    assert np.abs(flat.vertex_inputs[has] - raw[has]).max() > 1e-2
    orc = OracleCPU(mpc, eps_a, 0.1)
    orc.memoize = True
    memo = {}
    checked = 0
    for k in np.nonzero(has)[0]:
        for v in range(can.p + 1):
            key = flat.vertices[k, v].tobytes()
            if key not in memo:
                memo[key] = orc.lexicographic_u0(flat.vertices[k, v], can.deltas[flat.delta_idx[k]],
                                                 tol=lexicographic.TOL)
            assert np.allclose(flat.vertex_inputs[k, v], memo[key], rtol=0., atol=ATOL), (k, v)
            checked += 1
    assert checked == int(has.sum()) * (can.p + 1) and len(memo) == pairs


def test_refined_inputs_feed_the_explicit_law():
    """f2 on the refined tree: the law evaluated on the device interpolates the lexicographic
    inputs (same leaf, same weights as oracle/explicit_cpu.py on the refined arrays)."""
    from explicit_hybrid_mpc_amd import engine, examples, explicit, lexicographic
    from oracle.explicit_cpu import ExplicitFlatCPU
    mpc = helpers.make_instance('di', 0)
    can = mpc.compile()
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, _ = helpers.roots_of(mpc)
    gp = engine.GpuProblem(can, eps_a, 0.1)
    flat = gp.partition(np.array(roots), action='ecc')
    gp.close()
    lex = lexicographic.LexicographicInputs(can)
    lex.refine(flat)
    lex.close()
    half = examples.theta_box(mpc)
    X = (np.random.default_rng(3).random((400, can.p)) * 2 - 1) * half * 0.95
    law = explicit.ExplicitMPC(flat)
    U, leaf, _, _ = law.evaluate(X, return_info=True)
    law.close()
    cpu = ExplicitFlatCPU(flat.vertices, flat.vertex_inputs, flat.left, flat.right,
                          flat.info['n_roots'])
    same = 0
    for k in range(len(X)):
        u_ref, k_ref = cpu(X[k])
        if k_ref == leaf[k]:        # (a state within rounding of a shared face may go either way)
            same += 1
            assert np.allclose(U[k], u_ref, rtol=1e-9, atol=1e-11)
    assert same >= 390
