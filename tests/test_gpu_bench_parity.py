"""
Parity of the HEADLINE workload at its own scale (bench.py default: config 2, abs_frac 0.02,
eps_r 1e-2 -- 1.6 M nodes, depth 27), against the CPU oracle:

* nodes sampled deep in the device tree (depth >= 15) are grown again by the CPU restatement of
  lib/worker.py:293-417 (HiGHS at 1e-10; action 'lcss' from the exported record, a bounded number
  of visits each) and the visited part of every sub-forest must be the device's subtree:
  vertices bit-identical, same closed / split verdicts, vertex costs to 1e-7;
* near-threshold routing (hard part 1): no decision with |t*| below
  1e-6 (1 + |V_0|) is taken by a shortcut -- they are counted (ehm_tree_info.near_threshold)
  and are the only places where two correct solvers may disagree; the test asserts there is NO
  disagreement outside them.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

RTOL = 1e-7
ROUTE_TOL = 1e-6            # EHM_ROUTE_TOL (csrc/ehm_k2.h)
SAMPLES, VISITS, MIN_DEPTH = 200, 24, 15


def _depths(flat):
    # parents precede children in the breadth-first export
    depth = np.zeros(flat.n_nodes, dtype=np.int32)
    for k in np.nonzero(flat.left >= 0)[0]:
        depth[flat.left[k]] = depth[flat.right[k]] = depth[k] + 1
    return depth


def _regrow_on_the_cpu(flat, orc, picks, visits, routed):
    """Every picked node of the device tree grown again by PartitionCPU from its exported record:
    the visited part of the CPU sub-forest must be the device's subtree.  Returns (decided,
    splits, routed disagreements)."""
    from oracle.partition_cpu import PartitionCPU
    left, right = flat.left, flat.right
    decided = splits = routed_disagreements = 0
    for k in picks:
        root = dict(vertices=flat.vertices[k].copy(),
                    commutation=flat.deltas[flat.delta_idx[k]].copy(),
                    vertex_costs=flat.vertex_costs[k].copy(),
                    vertex_inputs=flat.vertex_inputs[k].copy(),
                    is_epsilon_suboptimal=False, leaf=True)
        cpu = PartitionCPU(orc, max_nodes=visits)
        cpu.run([root], [''], 'lcss')
        diverged = []
        for name in sorted(cpu.nodes, key=len):
            if any(name.startswith(d) for d in diverged):
                continue                        # below a routed disagreement the trees differ
            ref = cpu.nodes[name]
            kd = int(k)
            for bit in name:
                kd = int(left[kd] if bit == '0' else right[kd])
                assert kd >= 0, 'the CPU oracle split a node the device closed'
            assert np.array_equal(flat.vertices[kd], ref['vertices']), (k, name)
            assert np.allclose(flat.vertex_costs[kd], ref['vertex_costs'], rtol=RTOL,
                               atol=RTOL), (k, name)
            if ref['leaf'] and not ref['is_epsilon_suboptimal']:
                continue                        # not visited within the budget
            same = (flat.is_leaf(kd) == ref['leaf'] and
                    bool(flat.flags[kd] & 1) == ref['is_epsilon_suboptimal'])
            if not same:
                # two correct solvers may part ways ONLY at a near-threshold node
                assert routed[kd], ('un-routed disagreement', int(k), name, flat.tstar[kd])
                routed_disagreements += 1
                diverged.append(name)
                continue
            decided += 1
            splits += 0 if ref['leaf'] else 1
    return decided, splits, routed_disagreements


def test_config4_bench_tree_subforests_identical_to_cpu_oracle():
    """bench.py --workload config4 at ITS tolerances (BASELINE.json configs[3]: n_x = 6, n_u = 3,
    N = 10; eps_r 0.25, eps_a by the rule at abs_frac 0.4; 652 Delaunay roots, 162 866 regions):
    the tree of the wide kernels (ehm_k3.hip, midpoint table answering most midpoint solves)
    against the CPU oracle on sampled sub-forests, the pattern of the headline test below.  LPs
    of 57 columns x 369 rows: 120 samples of 16 visits."""
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from oracle.oracle_cpu import OracleCPU
    mpc = helpers.make_instance('chain', 0)
    gp = engine.GpuProblem(mpc.compile(), 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    eps_a = float(np.max(gp.solve_pt(0.4 * V)[0]))
    eps_r = 0.25
    gp.set_eps(eps_a, eps_r)
    roots, _ = ehm_tools.delaunay_roots(V)
    assert len(roots) == 652
    flat = gp.partition(roots, action='ecc')
    gp.close()
    assert flat.info['n_closed'] == 162866          # the tree of profiles/r4/bench_default.json
    depth = _depths(flat)
    tol = ROUTE_TOL * (1. + np.abs(flat.vertex_costs[:, 0]))
    routed = np.abs(flat.tstar) < tol
    rng = np.random.default_rng(1)
    min_depth = 8                                   # the tree is 19 levels deep
    deep = np.nonzero(depth >= min_depth)[0]
    deep_split = np.nonzero((depth >= min_depth) & (flat.left >= 0))[0]
    picks = np.concatenate([rng.choice(deep_split, size=60, replace=False),
                            rng.choice(deep, size=60, replace=False)])
    orc = OracleCPU(mpc, eps_a, eps_r)
    orc.memoize = True
    decided, splits, routed_disagreements = _regrow_on_the_cpu(flat, orc, picks, 16, routed)
    print('\nconfig4 bench tree: %d decisions (%d splits) of %d sampled sub-forests equal, %d '
          'routed disagreements, depth %d' % (decided, splits, len(picks), routed_disagreements,
                                              depth.max()))
    assert decided >= 3 * len(picks) and splits >= 60, (decided, splits)
    assert routed_disagreements == 0


def test_bench_tree_subforests_identical_to_cpu_oracle():
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    mpc = helpers.make_instance('lin', 0)
    gp = engine.GpuProblem(mpc.compile(), 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    eps_a = float(np.max(gp.solve_pt(0.02 * V)[0]))
    eps_r = 1e-2
    gp.set_eps(eps_a, eps_r)
    roots, _ = ehm_tools.delaunay_roots(V)
    # option check_witness: the tangent-plane bound is evaluated ALSO for the 489 k nodes the
    # inherited witness proves open -- the two LP-free verdicts must never contradict each other
    # (a conflict counts as a device error and the run fails with EHM_E_NUMERIC); the tree is the
    # default run's either way
    gp.set_option('check_witness', 1)
    flat = gp.partition(roots, action='ecc', max_nodes=1 << 22)     # the bench's default engine
    gp.close()
    assert flat.info['witness_inherited'] > 400000
    assert flat.n_nodes == 1610186 and flat.info['n_closed'] == 805104
    left = flat.left
    depth = _depths(flat)
    tol = ROUTE_TOL * (1. + np.abs(flat.vertex_costs[:, 0]))
    routed = np.abs(flat.tstar) < tol
    assert flat.info['near_threshold'] == int(np.sum(routed))
    assert flat.info['near_threshold'] < 1e-4 * flat.n_nodes
    rng = np.random.default_rng(0)
    # half of the samples among the nodes the device split (sub-forests with some depth), half
    # anywhere below the depth
    deep = np.nonzero(depth >= MIN_DEPTH)[0]
    deep_split = np.nonzero((depth >= MIN_DEPTH) & (left >= 0))[0]
    picks = np.concatenate([rng.choice(deep_split, size=SAMPLES // 2, replace=False),
                            rng.choice(deep, size=SAMPLES // 2, replace=False)])
    orc = OracleCPU(mpc, eps_a, eps_r)
    orc.memoize = True
    decided, splits, routed_disagreements = _regrow_on_the_cpu(flat, orc, picks, VISITS, routed)
    assert decided >= 4 * SAMPLES and splits >= SAMPLES, (decided, splits)
    assert routed_disagreements == 0
