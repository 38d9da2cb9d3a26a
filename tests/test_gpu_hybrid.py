"""
Multi-commutation (hybrid) partitions on the device engine (csrc/ehm_hybrid.h behind
ehm_partition_run) against

* the CPU restatement of lib/worker.py:241-417 / lib/oracle.py:175-414 (``oracle/``): identical
  tree, commutations and closed leaves on a whole small partition, and -- at the dimensions of
  BASELINE.json's configs[2] (n_x = 4, n_u = 2, N = 5, 32 commutations) -- identical SUB-FORESTS
  below nodes sampled from the device tree;
* the round-1 host loop over the batched Level-2 oracles (tests/hybrid_host_loop.py), an
  independent route through ehm_lcss_batch / ehm_vr_batch / ehm_feas_all_batch.

Tolerances: vertices bit-identical, costs 1e-7 relative, decisions identical.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

RTOL = 1e-7
CONFIG3_PICKS, CONFIG3_VISITS = 8, 8
CONFIG3_ABS_FRAC, CONFIG3_EPS_R, CONFIG3_MAX_DEPTH = 0.1, 1e-2, 18     # = bench.CONFIG3


def by_location(flat, locs):
    loc = flat.locations(locs)
    return {name: k for k, name in enumerate(loc)}


def assert_same_flat(a, b, locs):
    """Two FlatTrees with possibly different node numbering describe the same tree."""
    la, lb = by_location(a, locs), by_location(b, locs)
    assert set(la) == set(lb)
    for name, ka in la.items():
        kb = lb[name]
        assert np.array_equal(a.vertices[ka], b.vertices[kb]), name
        assert a.is_leaf(ka) == b.is_leaf(kb), name
        assert (a.flags[ka] & 3) == (b.flags[kb] & 3), name
        assert a.delta_idx[ka] == b.delta_idx[kb], name
        assert np.allclose(a.vertex_costs[ka], b.vertex_costs[kb], rtol=RTOL, atol=RTOL), name
        assert np.allclose(a.vertex_inputs[ka], b.vertex_inputs[kb], rtol=1e-5, atol=1e-6), name


def test_device_engine_equals_host_loop_over_batched_oracles():
    from explicit_hybrid_mpc_amd import engine
    from tests.hybrid_host_loop import grow_hybrid_host
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    gp = engine.GpuProblem(mpc.compile(), eps_a, 0.2)
    dev = gp.partition(np.array(roots), action='ecc')
    host = grow_hybrid_host(gp, np.array(roots), action='ecc')
    # the same with tiny chunks: every chunk boundary of the sweep is exercised
    import os
    os.environ['EHM_HY_CHUNK'] = '64'
    try:
        dev2 = gp.partition(np.array(roots), action='ecc')
    finally:
        del os.environ['EHM_HY_CHUNK']
    gp.close()
    assert dev.n_nodes == host.n_nodes and dev.info['n_closed'] == host.info['n_closed']
    assert_same_flat(dev, host, locs)
    assert_same_flat(dev2, host, locs)
    assert dev.info['swaps'] >= 0 and dev.info['blacklisted'] == 0
    assert dev.info['max_depth'] == max(len(s) for s in dev.locations([''] * len(locs)))
    # the engine records a LOWER bound of |t*| (sign-only stops, inherited negative verdicts)
    assert 0 < dev.info['min_margin'] <= host.info['min_margin'] * (1 + 1e-6)


def test_lcss_resume_on_hybrid_leaf():
    """action='lcss' with initial data (lib/scheduler.py:633-639) on the device engine."""
    from explicit_hybrid_mpc_amd import engine
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    gp = engine.GpuProblem(mpc.compile(), eps_a, 0.2)
    coarse = gp.partition(np.array(roots), action='ecc', max_depth=3)
    assert coarse.info['truncated'] == 1
    open_leaves = [k for k in range(coarse.n_nodes)
                   if coarse.is_leaf(k) and not (coarse.flags[k] & 1) and (coarse.flags[k] & 2)]
    assert open_leaves
    init = dict(delta=coarse.deltas[coarse.delta_idx[open_leaves]],
                vertex_costs=coarse.vertex_costs[open_leaves],
                vertex_inputs=coarse.vertex_inputs[open_leaves])
    sub = gp.partition(coarse.vertices[open_leaves], action='lcss', init=init)
    full = gp.partition(np.array(roots), action='ecc')
    gp.close()
    # the resumed sub-forests are the subtrees of the full run below the same nodes
    loc_c = coarse.locations(locs)
    lf = by_location(full, locs)
    sub_loc = sub.locations([loc_c[k] for k in open_leaves])
    for k, name in enumerate(sub_loc):
        kf = lf[name]
        assert np.array_equal(sub.vertices[k], full.vertices[kf]), name
        assert sub.is_leaf(k) == full.is_leaf(kf), name
        assert (sub.flags[k] & 1) == (full.flags[kf] & 1), name
        assert sub.delta_idx[k] == full.delta_idx[kf], name


def test_blacklist_and_retry_after_a_failed_vertex_solve(monkeypatch):
    """
    a9 (lib/oracle.py:198-218, 406-414): a commutation whose vertex solves fail is blacklisted
    for the node and the oracle is asked again.  The failure is forced for one commutation
    (EHM_HY_FAIL_DELTA); the CPU oracle is given the same failure.
    """
    from explicit_hybrid_mpc_amd import engine
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    from tests.test_gpu_partition import compare_trees
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    gp = engine.GpuProblem(mpc.compile(), eps_a, 0.2)
    plain = gp.partition(np.array(roots), action='ecc')
    used = [d for d in np.unique(plain.delta_idx) if d >= 0]
    # a commutation the plain run adopts somewhere but not everywhere
    counts = {d: int(np.sum(plain.delta_idx == d)) for d in used}
    fail = min(counts, key=counts.get)
    monkeypatch.setenv('EHM_HY_FAIL_DELTA', str(fail))
    forced = gp.partition(np.array(roots), action='ecc')
    gp.close()
    assert forced.info['blacklisted'] > 0
    assert not np.any(forced.delta_idx == fail)
    orc = OracleCPU(mpc, eps_a, 0.2)
    orc.memoize = True
    orc.fail_vertex_solves_of = int(fail)
    cpu = PartitionCPU(orc)
    cpu.run(roots, locs, 'ecc')
    for nd in cpu.nodes.values():
        if nd['vertex_costs'] is None:
            nd['vertex_costs'] = np.zeros(nd['vertices'].shape[0])
    compare_trees(forced, cpu.nodes, locs)
    assert orc.n_blacklisted > 0


def test_any_admissible_commutation_rule_identical_to_cpu_oracle():
    """
    Option "any_admissible" (csrc/ehm_hybrid.h, hy_draw): V_R and bar_D return a hashed draw among
    the admissible commutations -- the reference's Minimize(0) leaves that choice to its solver
    (lib/oracle.py:201, 347).  The CPU oracle takes the same draws (rule 'hash', path codes from
    PartitionCPU): identical trees for every seed, and a tree different from the canonical rule's
    for at least one of them (the rule does something).
    """
    from explicit_hybrid_mpc_amd import engine
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    from tests.test_gpu_partition import compare_trees
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    gp = engine.GpuProblem(mpc.compile(), eps_a, 0.2)
    canonical = gp.partition(np.array(roots), action='ecc')
    differs = 0
    for seed in (0, 1, 2):
        gp.set_option('any_admissible', seed + 1)
        drawn = gp.partition(np.array(roots), action='ecc')
        again = gp.partition(np.array(roots), action='ecc')
        assert drawn.n_nodes == again.n_nodes           # a run repeats itself
        assert_same_flat(drawn, again, locs)
        orc = OracleCPU(mpc, eps_a, 0.2)
        orc.memoize = True
        orc.bar_d_rule = 'hash'
        orc.hash_seed = seed
        cpu = PartitionCPU(orc)
        cpu.run(roots, locs, 'ecc')
        for nd in cpu.nodes.values():
            if nd['vertex_costs'] is None:
                nd['vertex_costs'] = np.zeros(nd['vertices'].shape[0])
        compare_trees(drawn, cpu.nodes, locs)
        if drawn.n_nodes != canonical.n_nodes or \
                not np.array_equal(np.sort(drawn.delta_idx), np.sort(canonical.delta_idx)):
            differs += 1
    gp.set_option('any_admissible', 0)
    back = gp.partition(np.array(roots), action='ecc')
    gp.close()
    assert_same_flat(back, canonical, locs)
    assert differs > 0


def test_config3_subforests_identical_to_cpu_oracle():
    """
    BASELINE.json configs[2] dimensions (n_x = 4, n_u = 2, N = 5: 32 commutations, LPs of
    20..25 columns): nodes sampled from the device tree are grown again by the CPU oracle
    (HiGHS, action 'lcss' from the exported record, a bounded number of visits each) and the
    visited part of every sub-forest must equal the device's subtree below that node.
    """
    from explicit_hybrid_mpc_amd import engine, examples
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    mpc = helpers.make_instance('pwa', 0)
    can = mpc.compile()
    assert can.n_delta == 32 and can.p == 4
    gp = engine.GpuProblem(can, 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    eps_a = float(np.max(gp.solve_pt(CONFIG3_ABS_FRAC * V)[0]))
    eps_r = CONFIG3_EPS_R
    gp.set_eps(eps_a, eps_r)
    roots, locs = helpers.roots_of(mpc)
    # the partition is cut at a depth: the optimal cost of this system jumps where a mode stops
    # being admissible and the refinement along that surface never ends (bench.py --max-depth)
    flat = gp.partition(np.array(roots), action='ecc', max_nodes=1 << 21,
                        max_depth=CONFIG3_MAX_DEPTH)
    gp.close()
    total = np.prod(2 * examples.theta_box(mpc))
    assert flat.info['truncated'] == 1
    assert 0.995 * total < flat.info['volume_closed'] <= total * (1 + 1e-9)
    assert flat.info['min_margin'] > 1e-6
    depth = np.zeros(flat.n_nodes, dtype=int)
    for k in range(flat.n_nodes):               # parents precede children in the export
        if flat.left[k] >= 0:
            depth[flat.left[k]] = depth[flat.right[k]] = depth[k] + 1
    assert len(set(int(d) for d in flat.delta_idx if d >= 0)) >= 2
    loc = flat.locations(locs)
    pos = {name: k for k, name in enumerate(loc)}
    # sample internal lcss nodes (they carry data) spread over the tree, deterministic
    cand = [k for k in range(flat.n_nodes) if (flat.flags[k] & 2) and not flat.is_leaf(k)
            and depth[k] <= CONFIG3_MAX_DEPTH - 6]
    rng = np.random.default_rng(0)
    picks = rng.choice(cand, size=min(CONFIG3_PICKS, len(cand)), replace=False)
    orc = OracleCPU(mpc, eps_a, eps_r)
    orc.memoize = True
    decided, costs = 0, 0
    for k in picks:
        # the exported record is the node's FINAL one (after a possible swap in place); the CPU
        # oracle continues from it and has to make the device's split
        root = dict(vertices=flat.vertices[k].copy(),
                    commutation=flat.deltas[flat.delta_idx[k]].copy(),
                    vertex_costs=flat.vertex_costs[k].copy(),
                    vertex_inputs=flat.vertex_inputs[k].copy(),
                    is_epsilon_suboptimal=False, leaf=True)
        cpu = PartitionCPU(orc, max_nodes=CONFIG3_VISITS)
        cpu.run([root], [loc[k]], 'lcss')
        assert cpu.min_margin > 1e-6
        for name, ref in cpu.nodes.items():
            if name[:-1] in pos and depth[pos[name[:-1]]] >= CONFIG3_MAX_DEPTH:
                continue            # below the depth the device run was cut at
            kd = pos[name]          # KeyError = the CPU split a node the device did not
            if depth[kd] >= CONFIG3_MAX_DEPTH and not ref['is_epsilon_suboptimal']:
                continue
            assert np.array_equal(flat.vertices[kd], ref['vertices']), name
            same_delta = np.array_equal(flat.deltas[flat.delta_idx[kd]].astype(int),
                                        ref['commutation'].astype(int))
            if (not ref['leaf']) or ref['is_epsilon_suboptimal']:
                # the CPU run finished with this node: same fate, same final commutation
                assert flat.is_leaf(kd) == ref['leaf'], name
                assert bool(flat.flags[kd] & 1) == ref['is_epsilon_suboptimal'], name
                assert same_delta, name
                decided += 1
            if same_delta:
                assert np.allclose(flat.vertex_costs[kd], ref['vertex_costs'],
                                   rtol=RTOL, atol=RTOL), name
                costs += 1
    assert decided >= 3 * CONFIG3_PICKS and costs >= decided


def test_sharded_hybrid_runs_tile_the_tree():
    """ehm_run_opts.deal_depth on the multi-commutation engine: the shares of 2 ranks (run one
    after the other) tile the unsharded tree."""
    from explicit_hybrid_mpc_amd import engine
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    gp = engine.GpuProblem(mpc.compile(), eps_a, 0.2)
    full = gp.partition(np.array(roots), action='ecc')
    parts = [gp.partition(np.array(roots), action='ecc', shard=(r, 2, 0), deal_depth=3)
             for r in range(2)]
    gp.close()
    lf = by_location(full, locs)
    leaves = {name for name, k in lf.items() if full.is_leaf(k)}
    got = set()
    for part in parts:
        lp = by_location(part, locs)
        remote = [k for k in lp.values() if part.flags[k] & 4]
        assert remote and all(part.is_leaf(k) for k in remote)
        for name, k in lp.items():
            kf = lf[name]
            assert np.array_equal(part.vertices[k], full.vertices[kf]), name
            if part.flags[k] & 4:
                continue
            assert part.is_leaf(k) == full.is_leaf(kf), name
            assert (part.flags[k] & 3) == (full.flags[kf] & 3), name
            assert part.delta_idx[k] == full.delta_idx[kf], name
            if part.is_leaf(k):
                got.add(name)
    assert got == leaves
    own = [p_.info['n_closed'] - (p_.info['replicated_closed'] if r else 0)
           for r, p_ in enumerate(parts)]
    assert sum(own) == full.info['n_closed']


def test_hybrid_engine_on_the_wide_kernels():
    """
    256 commutations on LPs of 33..37 columns: the multi-commutation engine through the wide
    (one workgroup per LP, MFMA normal matrix) kernels of ehm_k3.hip -- the kernel family of
    BASELINE.json's configs[3] / configs[4].  Batched oracles and a sub-forest of a truncated
    partition against the CPU oracle.
    """
    from explicit_hybrid_mpc_amd import engine, examples
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    mpc = examples.pwa_mpc(seed=0, N=8)
    examples.THETA_SCALE.setdefault(mpc.name, 0.25)
    can = mpc.compile()
    assert can.n_delta == 256 and can.n + can.p + 1 > 32
    gp = engine.GpuProblem(can, 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    J = gp.solve_pt(0.5 * V)[0]
    assert np.isfinite(J).all()
    eps_a, eps_r = float(np.max(J)), 1.0
    gp.set_eps(eps_a, eps_r)
    orc = OracleCPU(mpc, eps_a, eps_r)
    orc.memoize = True
    # P_theta on the device = the CPU oracle's
    th = 0.5 * V[[0, 5, 10]]
    Jd, ud, dd = gp.solve_pt(th)
    for k in range(3):
        u, delta, Jc, _ = orc.P_theta(th[k])
        assert abs(Jd[k] - Jc) <= RTOL * (1 + abs(Jc))
        assert np.array_equal(can.deltas[dd[k]].astype(int), delta.astype(int))
    roots, locs = helpers.roots_of(mpc)
    flat = gp.partition(np.array(roots[:3]), action='ecc', max_depth=7, max_nodes=1 << 18)
    gp.close()
    assert flat.info['n_nodes'] > 3
    loc = flat.locations(locs[:3])
    pos = {name: k for k, name in enumerate(loc)}
    cand = [k for k in range(flat.n_nodes) if (flat.flags[k] & 2) and not flat.is_leaf(k)]
    assert cand
    k = cand[len(cand) // 2]
    root = dict(vertices=flat.vertices[k].copy(),
                commutation=flat.deltas[flat.delta_idx[k]].copy(),
                vertex_costs=flat.vertex_costs[k].copy(),
                vertex_inputs=flat.vertex_inputs[k].copy(),
                is_epsilon_suboptimal=False, leaf=True)
    cpu = PartitionCPU(orc, max_nodes=2)
    cpu.run([root], [loc[k]], 'lcss')
    decided = 0
    for name, ref in cpu.nodes.items():
        kd = pos[name]
        assert np.array_equal(flat.vertices[kd], ref['vertices']), name
        if (not ref['leaf']) or ref['is_epsilon_suboptimal']:
            assert flat.is_leaf(kd) == ref['leaf'], name
            assert bool(flat.flags[kd] & 1) == ref['is_epsilon_suboptimal'], name
            decided += 1
        if np.array_equal(flat.deltas[flat.delta_idx[kd]].astype(int),
                          ref['commutation'].astype(int)):
            assert np.allclose(flat.vertex_costs[kd], ref['vertex_costs'], rtol=RTOL, atol=RTOL)
    assert decided >= 1


def test_config5_shaped_instance_on_the_engine():
    """
    The shape of BASELINE.json's configs[4] -- n_x = 8, n_u = 3, 4 integer modes, 9-vertex
    simplices -- with a horizon short enough to enumerate its commutations (N = 4: 256, the
    engine's limit; config 5 proper needs the prefix branch-and-bound of DESIGN.md 7c):
    P_theta, V_R, bar_E_delta_R, bar_D_delta_R and a truncated partition against the CPU oracle.
    """
    from explicit_hybrid_mpc_amd import engine, examples
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    mpc = examples.pwa4_mpc()
    can = mpc.compile()
    assert (can.p, can.n_u, can.n_delta, can.delta_size) == (8, 3, 256, 4)
    gp = engine.GpuProblem(can, 1., 1.)
    half = examples.theta_box(mpc)
    V = examples.box_vertices(half)
    J = gp.solve_pt(V[::32])[0]
    assert np.isfinite(J).all()
    eps_a, eps_r = 0.25 * float(np.max(J)), 0.5
    gp.set_eps(eps_a, eps_r)
    orc = OracleCPU(mpc, eps_a, eps_r)
    orc.memoize = True
    # a2 / a3: P_theta
    th = V[[3, 77, 200]]
    Jd, ud, dd = gp.solve_pt(th)
    for k in range(len(th)):
        u, delta, Jc, _ = orc.P_theta(th[k])
        assert abs(Jd[k] - Jc) <= RTOL * (1 + abs(Jc))
        assert np.array_equal(can.deltas[dd[k]].astype(int), delta.astype(int))
    # two 9-vertex simplices: one around a point off the mode boundaries, one straddling x_1 = 0
    rng = np.random.default_rng(5)
    E = np.vstack([np.zeros(8), np.eye(8)]) - 1. / 9.
    roots = np.array([0.5 * half * np.array([1, 1, -1, 1, -1, 1, 1, -1.]) + 0.08 * half * E,
                      0.4 * half * np.array([0, 1, 1, -1, 1, -1, 1, 1.]) + 0.15 * half * E])
    # a4: V_R
    didx, vJ, vu = gp.v_r(roots)
    for k in range(2):
        delta, vx = orc.V_R(roots[k])
        if delta is None:
            assert didx[k] < 0
            continue
        assert np.array_equal(can.deltas[didx[k]].astype(int), delta.astype(int))
        assert np.allclose(vJ[k], [v[1] for v in vx], rtol=RTOL, atol=RTOL)
    # a5 / a6 on the simplex that has a commutation
    k = int(np.argmax(didx >= 0))
    assert didx[k] >= 0
    closed, tb = gp.bar_e(roots[k:k + 1], vJ[k:k + 1])
    t_all = [orc.slack(roots[k], vJ[k], d)[0] for d in range(can.n_delta)]
    assert abs(tb[0] - max(t_all)) <= 1e-7 * (1 + abs(max(t_all)))
    assert bool(closed[0]) == orc.bar_E_delta_R(roots[k], vJ[k])
    ds, ths, vJ2, vu2, vs = gp.bar_d(roots[k:k + 1], vJ[k:k + 1], can.deltas[didx[k]][None])
    dstar, theta_star, vx, var_small = orc.bar_D_delta_R(roots[k], vJ[k], can.deltas[didx[k]])
    if dstar is None:
        assert ds[0] < 0
    else:
        assert np.array_equal(can.deltas[ds[0]].astype(int), dstar.astype(int))
        assert bool(vs[0]) == bool(var_small)
    # truncated partition of the two simplices on the device engine; its top against the oracle
    flat = gp.partition(roots, action='ecc', max_depth=3, max_nodes=1 << 16)
    gp.close()
    assert flat.info['lp_solves'] > 0
    per_micp = flat.info['lp_solves'] / max(flat.info['ref_solves'], 1)
    assert 1 < per_micp < can.n_delta * 10      # LPs solved per reference-equivalent oracle call
    loc = flat.locations(['a', 'b'])
    pos = {name: i for i, name in enumerate(loc)}
    cpu = PartitionCPU(orc, max_nodes=2)
    cpu.run([roots[0], roots[1]], ['a', 'b'], 'ecc')
    checked = 0
    for name, ref in cpu.nodes.items():
        kd = pos[name]
        assert np.array_equal(flat.vertices[kd], ref['vertices']), name
        if ref['commutation'] is not None and flat.delta_idx[kd] >= 0 and np.array_equal(
                flat.deltas[flat.delta_idx[kd]].astype(int), ref['commutation'].astype(int)):
            assert np.allclose(flat.vertex_costs[kd], ref['vertex_costs'], rtol=RTOL, atol=RTOL)
            checked += 1
    assert checked >= 1


def test_cwh_z_job3_subforests_identical_to_cpu_oracle():
    """
    The reference's own law at depth: third cwh_z job (make_jobs.sh:60-66: abs_frac 0.1, rel_err
    0.1; 9 484 leaves, 81 commutations, QP / QCQP oracles).  Nodes sampled from the device tree
    are grown again by the CPU oracle (uncondensed models, oracle/qp_numpy.py; a few visits
    each); what it decides must be what the device decided.
    """
    from explicit_hybrid_mpc_amd import examples
    from oracle.oracle_cpu import OracleCPU, SolverError
    from oracle.partition_cpu import PartitionCPU
    from oracle.satellite_cpu import SatelliteZCPU
    from oracle import geometry
    full_set, part, oracle = examples.example('cwh_z', abs_frac=0.1, rel_err=0.1)
    roots, locs = geometry.delaunay_simplices(full_set)
    flat = oracle.gpu.partition(np.array(roots), action='ecc', max_nodes=1 << 20)
    eps_a, eps_r = oracle.eps_a, oracle.eps_r
    oracle.close()
    assert flat.info['n_leaves'] == 9484 and flat.info['swaps'] > 0
    total = np.prod(2 * examples.theta_box(oracle.mpc))
    assert abs(flat.info['volume_closed'] - total) <= 1e-9 * total
    loc = flat.locations(locs)
    pos = {name: k for k, name in enumerate(loc)}
    tol = 1e-6 * (1. + np.abs(flat.vertex_costs[:, 0]))
    cand = [k for k in range(flat.n_nodes) if (flat.flags[k] & 2) and not flat.is_leaf(k)
            and len(loc[k]) >= 8]
    rng = np.random.default_rng(3)
    picks = rng.choice(cand, size=20, replace=False)
    orc = OracleCPU(SatelliteZCPU(4), eps_a, eps_r)
    orc.memoize = True
    decided = 0
    for k in picks:
        root = dict(vertices=flat.vertices[k].copy(),
                    commutation=flat.deltas[flat.delta_idx[k]].copy(),
                    vertex_costs=flat.vertex_costs[k].copy(),
                    vertex_inputs=flat.vertex_inputs[k].copy(),
                    is_epsilon_suboptimal=False, leaf=True)
        cpu = PartitionCPU(orc, max_nodes=6)
        try:
            cpu.run([root], [loc[k]], 'lcss')
        except SolverError:
            continue                # the checker's own QP method gave up on this instance
        for name, ref in cpu.nodes.items():
            kd = pos[name]
            assert np.array_equal(flat.vertices[kd], ref['vertices']), name
            if ref['leaf'] and not ref['is_epsilon_suboptimal']:
                continue
            if abs(flat.tstar[kd]) < tol[kd]:
                continue            # near-threshold: two solvers may part ways here
            assert flat.is_leaf(kd) == ref['leaf'], name
            assert bool(flat.flags[kd] & 1) == ref['is_epsilon_suboptimal'], name
            assert np.array_equal(flat.deltas[flat.delta_idx[kd]].astype(int),
                                  ref['commutation'].astype(int)), name
            assert np.allclose(flat.vertex_costs[kd], ref['vertex_costs'], rtol=1e-6, atol=1e-9)
            decided += 1
    assert decided >= 50


def test_hybrid_capacity_and_infeasible_set_errors():
    """Error behaviour of the device engine: node pool exhausted -> EHM_E_CAPACITY (and the
    handle is usable again); a set with infeasible regions -> the reference's RuntimeError
    (lib/worker.py:266)."""
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd._capi import EhmError, EHM_E_CAPACITY, EHM_E_INFEASIBLE
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    gp = engine.GpuProblem(mpc.compile(), eps_a, 0.2)
    full = gp.partition(np.array(roots), action='ecc')
    with pytest.raises(EhmError) as e:
        gp.partition(np.array(roots), action='ecc', max_nodes=full.n_nodes // 3)
    assert e.value.code == EHM_E_CAPACITY
    again = gp.partition(np.array(roots), action='ecc')
    assert again.n_nodes == full.n_nodes
    from oracle import geometry
    big, _ = geometry.delaunay_simplices(30 * examples.box_vertices(examples.theta_box(mpc)))
    with pytest.raises(RuntimeError, match='Theta contains infeasible regions') as e:
        gp.partition(np.array(big), action='ecc')
    assert e.value.code == EHM_E_INFEASIBLE
    gp.close()
