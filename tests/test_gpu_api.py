"""
The reference-signature ``Oracle`` (lib/oracle.py:104-443 conventions), hybrid (multi-
commutation) oracles, the hybrid partition driver, ``alg_call`` on Tree objects and the
reference-format pickle -- all through the C-ABI on the GPU, against the CPU oracle.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu
RTOL = 1e-7


@pytest.fixture(scope='module')
def pwa():
    from explicit_hybrid_mpc_amd.oracle import Oracle
    from oracle.oracle_cpu import OracleCPU
    mpc = helpers.make_instance('pwa', 0)
    eps_a, eps_r = 0.1, 0.5
    gpu = Oracle(mpc, eps_a, eps_r)
    cpu = OracleCPU(mpc, eps_a, eps_r)
    cpu.memoize = True
    yield mpc, gpu, cpu
    gpu.close()


def close(a, b):
    return abs(a - b) <= RTOL * (1 + abs(b))


def test_p_theta_and_p_theta_delta(pwa):
    from explicit_hybrid_mpc_amd import examples
    mpc, gpu, cpu = pwa
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(21)
    for k in range(6):
        theta = rng.uniform(-1, 1, 4) * half
        u, delta, J, t = gpu.P_theta(theta)
        u_r, delta_r, J_r, _ = cpu.P_theta(theta)
        assert close(J, J_r)
        assert np.array_equal(delta.astype(int), delta_r.astype(int))
        assert u.shape == (2,) and t >= 0
        u2, J2, _ = gpu.P_theta_delta(theta, delta)
        assert close(J2, J_r)
    assert gpu.P_theta(0.2 * half, check_feasibility=True) is True
    assert gpu.P_theta(40 * half, check_feasibility=True) is False
    assert gpu.P_theta(40 * half)[0] is None
    # a commutation that starts in mode 1 is infeasible well inside x_1 > overlap
    theta = np.array([0.4, 0., 0., 0.])
    assert gpu.P_theta_delta(theta, cpu.deltas[31], check_feasibility=True) == \
        cpu.P_theta_delta(theta, cpu.deltas[31], check_feasibility=True) == False  # noqa: E712
    assert gpu.P_theta_delta(theta, cpu.deltas[31])[0] is None
    with pytest.raises(Exception):
        gpu.P_theta_delta(theta, np.ones(10))       # not an admissible commutation


def test_v_r_bar_e_bar_d(pwa):
    from explicit_hybrid_mpc_amd import examples
    mpc, gpu, cpu = pwa
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(22)
    n_feas = n_none = n_better = 0
    for k in range(8):
        scale = [0.05, 0.3, 0.9][k % 3]
        ctr = rng.uniform(-1, 1, 4) * half * (1 - scale)
        R = np.clip(ctr + scale * rng.uniform(-1, 1, (5, 4)) * half, -half, half)
        d_g, vx_g = gpu.V_R(R)
        d_c, vx_c = cpu.V_R(R)
        assert (d_g is None) == (d_c is None)
        if d_g is None:
            n_none += 1
            continue
        n_feas += 1
        assert np.array_equal(d_g.astype(int), d_c.astype(int))
        V = np.array([v[1] for v in vx_g])
        assert np.allclose(V, [v[1] for v in vx_c], rtol=RTOL, atol=RTOL)
        assert gpu.bar_E_delta_R(R, V) == cpu.bar_E_delta_R(R, V)
        out_g = gpu.bar_D_delta_R(R, V, d_g)
        out_c = cpu.bar_D_delta_R(R, V, d_c)
        assert (out_g[0] is None) == (out_c[0] is None)
        if out_g[0] is not None:
            n_better += 1
            assert np.array_equal(out_g[0].astype(int), out_c[0].astype(int))
            assert np.allclose([v[1] for v in out_g[2]], [v[1] for v in out_c[2]],
                               rtol=RTOL, atol=RTOL)
            assert out_g[3] == out_c[3]
            # theta* must lie in R and give the same cost as the CPU maximiser
            Jg = cpu.P_theta_delta(out_g[1], out_g[0])[1]
            Jc = cpu.P_theta_delta(out_c[1], out_c[0])[1]
            assert abs(Jg - Jc) <= 1e-5 * (1 + abs(Jc))
    assert n_feas >= 3


def test_hybrid_partition_identical_to_cpu():
    from explicit_hybrid_mpc_amd import examples, partition
    from explicit_hybrid_mpc_amd.oracle import Oracle
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    from tests.test_gpu_partition import compare_trees
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    eps_r = 0.2
    roots, locs = helpers.roots_of(mpc)
    orc = OracleCPU(mpc, eps_a, eps_r)
    orc.memoize = True
    cpu = PartitionCPU(orc)
    cpu.run(roots, locs, 'ecc')
    gpu = Oracle(mpc, eps_a, eps_r)
    flat = partition.run_engine(gpu, np.array(roots), action='ecc')
    gpu.close()
    # nodes that never received commutation data have zero vertex costs on both sides
    for nd in cpu.nodes.values():
        if nd['vertex_costs'] is None:
            nd['vertex_costs'] = np.zeros(nd['vertices'].shape[0])
    compare_trees(flat, cpu.nodes, locs)
    loc = flat.locations(locs)
    n_delta_used = set()
    for k, name in enumerate(loc):
        ref = cpu.nodes[name]
        if ref['commutation'] is not None:
            assert np.array_equal(flat.deltas[flat.delta_idx[k]].astype(int),
                                  ref['commutation'].astype(int)), name
            n_delta_used.add(int(flat.delta_idx[k]))
    assert len(n_delta_used) >= 2
    total = np.prod(2 * examples.theta_box(mpc))
    assert abs(flat.info['volume_closed'] - total) <= 1e-9 * total


def test_alg_call_grows_tree_in_place_and_pickles(tmp_path):
    from explicit_hybrid_mpc_amd import examples, partition, tree_io, tools
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    full_set, part_tree, oracle = examples.example('linear', abs_frac=0.5, rel_err=1.0)
    mpc = oracle.mpc
    assert abs(oracle.eps_a - helpers.eps_a_rule(mpc, 0.5)) <= 1e-7 * oracle.eps_a
    # one branch through the reference's entry point
    roots, locs = tools.delaunay_roots(full_set)
    branch = Tree(NodeData(vertices=roots[3].copy()), top=False)
    assert partition.alg_call(oracle, 'ecc', branch, locs[3]) is None
    cpu = PartitionCPU(OracleCPU(mpc, oracle.eps_a, oracle.eps_r))
    cpu.run([roots[3]], [''], 'ecc')
    got = {loc: node for node, loc in branch.walk()}
    assert set(got) == set(cpu.nodes)
    for loc, node in got.items():
        ref = cpu.nodes[loc]
        assert node.is_leaf() == ref['leaf']
        assert np.array_equal(node.data.vertices, ref['vertices'])
        assert node.data.is_epsilon_suboptimal == ref['is_epsilon_suboptimal']
        assert np.allclose(node.data.vertex_costs, ref['vertex_costs'], rtol=RTOL, atol=RTOL)
    # 'lcss' on a leaf that already carries data (resume semantics, lib/scheduler.py:633-639)
    leaf = next(n for n, _ in branch.leaves())
    n_before = sum(1 for _ in branch.walk())
    Partition = partition.Partitioner(oracle)
    Partition.lcss(leaf, '')
    assert sum(1 for _ in branch.walk()) == n_before and leaf.data.is_epsilon_suboptimal
    # whole set -> right-spine tree -> reference-format pickle -> back
    root, flat = partition.partition_set(oracle, full_set)
    n_leaves = sum(1 for _ in root.leaves())
    assert n_leaves == flat.info['n_leaves']
    path = str(tmp_path / 'tree.pkl')
    tree_io.dump_reference(root, path)
    back = tree_io.load_reference(path)
    assert sum(1 for _ in back.leaves()) == n_leaves
    first = next(n for n, _ in back.leaves())
    assert hasattr(first.data, 'commutation') and first.data.vertex_inputs.shape == (5, 2)
    oracle.close()


def test_dealt_persistent_launches_cover_the_unsharded_tree():
    """
    ehm_run_opts.deal_depth: one persistent launch per rank from the roots, dealt at a tree depth
    by the nodes' path codes.  The shares of 3 ranks (run one after the other here) tile the tree.
    """
    from explicit_hybrid_mpc_amd import engine, distributed
    mpc = helpers.make_instance('lin', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    gp = engine.GpuProblem(mpc.compile(), eps_a, 0.05)
    full = gp.partition(np.array(roots))
    world = 3
    depth = distributed.deal_depth_for(len(roots), world, per_rank=32)
    parts = [gp.partition(np.array(roots), shard=(r, world, 0), deal_depth=depth)
             for r in range(world)]
    gp.close()
    full_loc = full.locations(locs)
    full_leaves = {full_loc[k] for k in range(full.n_nodes) if full.is_leaf(k)}
    got = set()
    for part in parts:
        assert part.info['decide_launches'] == 1            # no sweeps at all
        loc = part.locations(locs)
        mine = {loc[k] for k in range(part.n_nodes)
                if part.is_leaf(k) and not (part.flags[k] & 4)}
        # the replicated top's closed leaves appear in every share; everything else once
        shared = {name for name in mine & got}
        assert all(len(name) - len(locs[0]) <= depth + 40 for name in shared)
        got |= mine
        remote = [k for k in range(part.n_nodes) if part.flags[k] & 4]
        assert remote and all(part.is_leaf(k) for k in remote)
    assert got == full_leaves
    own_closed = [p_.info['n_closed'] - (p_.info['replicated_closed'] if r else 0)
                  for r, p_ in enumerate(parts)]
    assert sum(own_closed) == full.info['n_closed']
    own_nodes = [p_.info['n_nodes'] - (p_.info['replicated_nodes'] if r else 0)
                 for r, p_ in enumerate(parts)]
    assert sum(own_nodes) == full.n_nodes


def test_sharded_runs_cover_the_unsharded_tree():
    """Two ranks' shares (run one after the other on this GPU) tile the full tree."""
    from explicit_hybrid_mpc_amd import engine
    mpc = helpers.make_instance('lin', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    gp = engine.GpuProblem(mpc.compile(), eps_a, 0.05)
    full = gp.partition(np.array(roots))
    parts = [gp.partition(np.array(roots), shard=(r, 2, 64)) for r in range(2)]
    gp.close()
    full_loc = full.locations(locs)
    full_leaves = {full_loc[k] for k in range(full.n_nodes) if full.is_leaf(k)}
    got = set()
    for part in parts:
        loc = part.locations(locs)
        mine = {loc[k] for k in range(part.n_nodes)
                if part.is_leaf(k) and not (part.flags[k] & 4)}
        assert not (mine & got)
        got |= mine
        remote = [k for k in range(part.n_nodes) if part.flags[k] & 4]
        assert remote and all(part.is_leaf(k) for k in remote)
    assert got == full_leaves
    assert parts[0].info['n_closed'] + parts[1].info['n_closed'] == full.info['n_closed']
    assert abs(parts[0].info['n_closed'] - parts[1].info['n_closed']) < 0.35 * full.info['n_closed']
