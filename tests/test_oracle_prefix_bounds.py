"""
The bounds a branch-and-bound over mode prefixes would use (DESIGN.md section 7c item 1,
oracle/prefix_bb.py), checked against enumeration on instances small enough to enumerate, and
the search itself on config 5's shape at N = 8 (65 536 sequences).  CPU only.
"""

import itertools
import numpy as np

from explicit_hybrid_mpc_amd import examples
from oracle import prefix_bb
from oracle.oracle_cpu import OracleCPU
from tests import helpers


def test_prefix_relaxations_bound_every_completion():
    mpc = helpers.make_instance('pwa_small', 0)          # 2 modes, N = 3: 8 sequences
    orc = OracleCPU(mpc, 0.05, 0.2)
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(0)
    seqs = orc.sequences
    for trial in range(6):
        theta = rng.uniform(-0.9, 0.9, 2) * half
        J_full = {s: (orc._point(theta, d)[2] if orc._point(theta, d)[0] else np.inf)
                  for d, s in enumerate(seqs)}
        R = theta + 0.15 * half * rng.uniform(-1, 1, (3, 2))
        R = np.clip(R, -half, half)
        V_bar = rng.uniform(0.5, 1.5, 3) * max(0.1, min(v for v in J_full.values()
                                                        if np.isfinite(v)) if any(
            np.isfinite(v) for v in J_full.values()) else 0.1)
        t_full = {s: orc.slack(R, V_bar, d)[0] for d, s in enumerate(seqs)}
        for k in range(1, mpc.N + 1):
            for prefix in itertools.product(range(mpc.delta_size), repeat=k):
                comp = [s for s in seqs if s[:k] == prefix]
                lb = prefix_bb.prefix_cost(mpc, prefix, theta)
                assert lb <= min(J_full[s] for s in comp) + 1e-8
                ub = prefix_bb.prefix_slack(mpc, prefix, R, V_bar, orc.eps_a, orc.eps_r)
                assert ub >= max(t_full[s] for s in comp) - 1e-8
                if k == mpc.N:                       # a full sequence is no relaxation
                    assert abs(lb - J_full[prefix]) <= 1e-8 or not np.isfinite(J_full[prefix])


def test_best_first_search_equals_enumeration():
    mpc = examples.pwa4_mpc()                            # 4 modes, N = 4: 256 sequences
    orc = OracleCPU(mpc, 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    for v in V[[5, 130]]:
        u, delta, J, _ = orc.P_theta(v)
        Jb, seq, n_lp = prefix_bb.p_theta_bb(mpc, v)
        assert abs(Jb - J) <= 1e-7 * (1 + abs(J))
        assert np.array_equal(mpc.sequence_to_delta(seq).astype(int), delta.astype(int))
        assert n_lp < 0.5 * len(orc.sequences)           # and it looked at far fewer problems


def test_config5_scale_search():
    """n_x = 8, n_u = 3, 4 modes, N = 8: P_theta and a bar_E verdict among 65 536 sequences."""
    mpc = examples.pwa4_mpc(N=8)
    V = examples.box_vertices(examples.theta_box(mpc))
    theta = V[37]
    J, seq, n_lp = prefix_bb.p_theta_bb(mpc, theta)
    assert np.isfinite(J) and len(seq) == 8
    assert n_lp < 4 ** 8 / 50                            # a small fraction of the enumeration
    # its optimum is the optimum of the sequence it names
    from oracle.lp_models import FixedCommutationModel
    res = prefix_bb._solve(FixedCommutationModel(mpc, seq).lp_point(theta))
    assert res.status == 0 and abs(res.fun - J) <= 1e-8 * (1 + abs(J))
    # a simplex around theta with vertex costs far above the optimum: some sequence is better
    # by more than the tolerance somewhere (not closed); with huge tolerances: closed
    E = np.vstack([np.zeros(8), np.eye(8)]) - 1. / 9.
    R = 0.9 * theta + 0.02 * examples.theta_box(mpc) * E
    closed, n1 = prefix_bb.bar_e_bb(mpc, R, np.full(9, 3. * J), 0.1 * J, 0.1)
    assert not closed and n1 < 4 ** 8 / 50
    closed, n2 = prefix_bb.bar_e_bb(mpc, R, np.full(9, 1.01 * J), 10. * J, 10.)
    assert closed and n2 <= 4                            # every first-step prefix already has t* < 0


def test_region_table_keeps_every_oracle_answer():
    """
    The cost-pruned table of a region (sequences.relevant_sequences is the
    device version): on 4 modes x N = 4 (256 sequences, enumerable) the oracles restricted to the
    table return what they return on the full enumeration -- P_theta at points of the region,
    V_R on it, and the partition's first decisions (bar_E / bar_D with the canonical rule).
    """
    from oracle.partition_cpu import PartitionCPU
    mpc = examples.pwa4_mpc()
    half = examples.theta_box(mpc)
    E = np.vstack([np.zeros(8), np.eye(8)]) - 1. / 9.
    R = 0.8 * examples.box_vertices(half)[37] + 0.08 * half * E
    seqs, U, dive, n_lp = prefix_bb.relevant_sequences(mpc, [R])
    assert 2 <= len(seqs) < 64 and dive in seqs
    sub = mpc.restrict(seqs)
    assert sub.mode_sequences() == seqs and sub.compile().n_delta == len(seqs)
    assert mpc.compile().n_delta == 256                 # the original is untouched
    full = OracleCPU(mpc, 0.004, 0.01)
    part = OracleCPU(sub, 0.004, 0.01)
    rng = np.random.default_rng(1)
    for a in rng.dirichlet(np.ones(9), size=3):
        u1, d1, J1, _ = full.P_theta(a @ R)
        u2, d2, J2, _ = part.P_theta(a @ R)
        assert abs(J1 - J2) <= 1e-9 and np.array_equal(d1, d2) and J1 <= U + 1e-9
    d1, vx1 = full.V_R(R)
    d2, vx2 = part.V_R(R)
    assert np.array_equal(d1, d2)
    assert np.allclose([v[1] for v in vx1], [v[1] for v in vx2], atol=1e-9)
    # the first visits of the partition of R: ecc, then lcss (bar_E, bar_D) on the root
    trees = []
    for orc in (full, part):
        cpu = PartitionCPU(orc, max_nodes=3)
        cpu.run([R], [''], 'ecc')
        trees.append(cpu.nodes)
    assert set(trees[0]) == set(trees[1]) and len(trees[0]) >= 3
    for name in trees[0]:
        a, b = trees[0][name], trees[1][name]
        assert a['leaf'] == b['leaf'] and a['is_epsilon_suboptimal'] == b['is_epsilon_suboptimal']
        if a['commutation'] is not None:
            assert np.array_equal(a['commutation'], b['commutation'])
            assert np.allclose(a['vertex_costs'], b['vertex_costs'], atol=1e-9)


def test_p_theta_against_an_independent_milp():
    """
    P_theta three ways on instances HiGHS' own branch-and-bound handles in seconds: ONE
    mixed-integer LP with binary mode indicators and big-M dynamics (oracle/milp_check.py: the
    reference's formulation, lib/oracle.py:42-46), the enumeration of the fixed-sequence LPs
    (OracleCPU) and the best-first search over mode prefixes.  Same optimal cost; the MILP's
    sequence attains it.
    """
    from oracle import milp_check
    from oracle.lp_models import FixedCommutationModel
    cases = [(helpers.make_instance('pwa_small', 0), 6), (examples.pwa4_mpc(N=3), 3)]
    rng = np.random.default_rng(4)
    compared = 0
    for mpc, n_pts in cases:
        orc = OracleCPU(mpc, 1., 1.)
        half = examples.theta_box(mpc)
        for _ in range(n_pts):
            theta = rng.uniform(-0.9, 0.9, half.size) * half
            J_milp, seq = milp_check.p_theta_milp(mpc, theta)
            u, delta, J_enum, _ = orc.P_theta(theta)
            if delta is None:
                assert not np.isfinite(J_milp)
                continue
            assert abs(J_milp - J_enum) <= 1e-7 * (1 + abs(J_enum))
            J_bb, seq_bb, _ = prefix_bb.p_theta_bb(mpc, theta)
            assert abs(J_bb - J_enum) <= 1e-8 * (1 + abs(J_enum))
            res = prefix_bb._solve(FixedCommutationModel(mpc, seq).lp_point(theta))
            assert res.status == 0 and abs(res.fun - J_milp) <= 1e-7 * (1 + abs(J_milp))
            compared += 1
    assert compared >= 6


def test_bar_e_against_an_independent_milp():
    """
    The suboptimality test as the reference states it -- ONE mixed-integer problem over modes,
    parameter and trajectory (lib/oracle.py:89-97) -- solved by HiGHS' branch-and-bound
    (oracle/milp_check.bar_e_milp), against the enumeration of the fixed-sequence slack LPs
    (OracleCPU) and the prefix search: the same largest slack, hence the same verdict.
    """
    from oracle import milp_check
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    orc = OracleCPU(mpc, eps_a, 0.2)
    rng = np.random.default_rng(7)
    n_open = n_closed = 0
    for R in helpers.random_simplices(mpc, rng, 14, scale_lo=-1.5):
        delta, vx = orc.V_R(R)
        if delta is None:
            continue
        V = np.array([v[1] for v in vx])
        t_enum = max(orc.slack(R, V, d)[0] for d in range(len(orc.models)))
        t_milp, seq = milp_check.bar_e_milp(mpc, R, V, eps_a, 0.2)
        assert abs(t_milp - t_enum) <= 1e-7 * (1 + abs(t_enum)), (t_milp, t_enum)
        closed = orc.bar_E_delta_R(R, V)
        assert closed == (not (t_milp >= 0.))
        closed_bb, _ = prefix_bb.bar_e_bb(mpc, R, V, eps_a, 0.2)
        assert closed_bb == closed
        n_open += not closed
        n_closed += closed
    assert n_open >= 2 and n_closed >= 2


def test_v_r_and_bar_d_against_independent_milps():
    """
    V_R and bar_D as the reference states them -- ONE mixed-integer problem each, with a copy of
    the trajectory per vertex and one shared mode sequence (lib/oracle.py:57-66, 101-102) --
    solved by HiGHS' branch-and-bound (oracle/milp_check.v_r_milp / bar_d_milp), against the
    enumerating oracle: V_R is feasible for the same simplices and the solver's sequence is one
    the enumeration finds feasible at every vertex (the canonical answer is the FIRST of them);
    bar_D's optimum is the canonical rule's largest slack over the sequences feasible at every
    vertex, its sequence attains it, and "no better commutation" coincides.
    """
    from oracle import milp_check
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    orc = OracleCPU(mpc, eps_a, 0.2)
    rng = np.random.default_rng(21)
    n_feas = n_infeas = n_star = n_none = 0
    # small simplices (one sequence fits all vertices) and a few that span most of the box
    some = list(helpers.random_simplices(mpc, rng, 12, scale_lo=-1.5)) + \
        list(helpers.random_simplices(mpc, rng, 8, scale_lo=-0.15, scale_hi=0.)) + \
        [np.array(r) for r in helpers.roots_of(mpc)[0]]
    for R in some:
        R = np.asarray(R)
        delta, vx = orc.V_R(R)
        seq = milp_check.v_r_milp(mpc, R)
        assert (delta is None) == (seq is None)
        if delta is None:
            n_infeas += 1
            continue
        n_feas += 1
        ok = [d for d in range(len(orc.models)) if orc._feasible_on_vertices(R, d)]
        d_milp = next(d for d in range(len(orc.models))
                      if tuple(np.asarray(orc.deltas[d]).reshape(mpc.N, -1).argmax(axis=1)) == seq)
        assert d_milp in ok
        assert np.array_equal(np.asarray(delta).astype(int), np.asarray(orc.deltas[ok[0]]).astype(int))
        V = np.array([v[1] for v in vx])
        t_all = {d: orc.slack(R, V, d)[0] for d in ok}
        t_enum = max(t_all.values())
        t_milp, seq_d, theta = milp_check.bar_d_milp(mpc, R, V, eps_a, 0.2)
        assert abs(t_milp - t_enum) <= 1e-7 * (1 + abs(t_enum)), (t_milp, t_enum)
        d_star = next(d for d in ok
                      if tuple(np.asarray(orc.deltas[d]).reshape(mpc.N, -1).argmax(axis=1)) == seq_d)
        assert t_all[d_star] >= t_enum - 1e-6 * (1 + abs(t_enum))
        # theta* is a point of the simplex
        lam = np.linalg.lstsq(np.vstack([R.T, np.ones(R.shape[0])]),
                              np.concatenate([theta, [1.]]), rcond=None)[0]
        assert lam.min() >= -1e-7
        out = orc.bar_D_delta_R(R, V, delta)
        if t_enum < 0.:
            assert out[0] is None
            n_none += 1
        elif out[0] is not None:
            # the canonical answer attains the optimum the one-problem statement found
            d_can = next(d for d in ok if np.array_equal(np.asarray(orc.deltas[d]).astype(int),
                                                         np.asarray(out[0]).astype(int)))
            assert t_all[d_can] >= t_milp - 1e-6 * (1 + abs(t_milp))
            n_star += 1
    assert n_feas >= 4 and n_infeas >= 1 and n_star + n_none >= 2
