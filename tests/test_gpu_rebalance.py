"""
The resumable engine (ehm_partition_begin / step / take / give / finish) on the device: two
handles on one GPU play two ranks, frontier nodes are moved between them with the plan of
distributed.balance_plan, and the merged shares must be exactly the tree one run grows.
(The torch.distributed side of the same loop is exercised with gloo in
test_distributed_gloo.py.)
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('device_blocks', [False, True])
def test_moved_frontier_nodes_give_the_same_tree(device_blocks):
    """device_blocks: the hand-over through torch tensors in DEVICE memory (take_device /
    give_device: what distributed._exchange sends peer to peer under RCCL) instead of numpy."""
    if device_blocks:
        # torch brings its own HIP runtime: it has to initialise before libehmpc's does (as in
        # bench.py, which imports torch first) or it finds no device -- a process of its own
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        code = ('import torch; torch.cuda.init(); import sys; sys.path.insert(0, %r); '
                'from tests.test_gpu_rebalance import _moved_frontier_nodes; '
                '_moved_frontier_nodes(True); print("DEVICE_BLOCKS_OK")' % root)
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=root)
        assert r.returncode == 0 and 'DEVICE_BLOCKS_OK' in r.stdout, \
            r.stdout[-3000:] + r.stderr[-3000:]
        return
    _moved_frontier_nodes(False)


def _moved_frontier_nodes(device_blocks):
    from explicit_hybrid_mpc_amd import distributed, engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    mpc = helpers.make_instance('lin', 0)
    can = mpc.compile()
    V = examples.box_vertices(examples.theta_box(mpc))
    roots, locs = ehm_tools.delaunay_roots(V)
    gps = [engine.GpuProblem(can, 1., 1.) for _ in range(3)]
    eps_a = float(np.max(gps[0].solve_pt(0.12 * V)[0]))
    for g in gps:
        g.set_eps(eps_a, 1e-2)
    ref = gps[2].partition(roots, action='ecc')
    world = 2
    runs = [gps[r].begin(roots, shard=(r, world, 64)) for r in range(world)]
    logs = [[] for _ in range(world)]
    rnd = 0
    moved = 0
    while True:
        counts = [run.step(2) for run in runs]
        if sum(counts) == 0:
            break
        # make the exchange bite: rank 0 works at half speed in this emulation
        plan = distributed.balance_plan(counts, tolerance=0.02, min_move=4)
        for donor, receiver, n in plan:
            if device_blocks:
                ids, rec, meta = runs[donor].take_device(n, 'cuda:0')
                assert rec.is_cuda and meta.is_cuda and tuple(rec.shape) == (n, runs[donor].nrec)
                first = runs[receiver].give_device(rec.clone(), meta.clone())
            else:
                ids, rec, meta = runs[donor].take(n)
                assert rec.shape == (n, runs[donor].nrec) and meta.shape == (n, 2)
                first = runs[receiver].give(rec, meta)
            logs[donor].append(dict(kind='give', round=rnd, peer=receiver, ids=ids))
            logs[receiver].append(dict(kind='recv', round=rnd, peer=donor, first=first, count=n))
            moved += n
        rnd += 1
    parts = [run.finish(export=True) for run in runs]
    for g in gps:
        g.close()
    assert moved > 0
    for part in parts:
        assert part.info['n_nodes'] == part.n_nodes
        assert int(np.sum((part.flags & 32) > 0)) == sum(
            e['count'] for e in logs[parts.index(part)] if e['kind'] == 'recv')
    received = distributed.resolve_received(parts, logs, locs)
    merged = distributed.merge_flat(parts, locs, received)
    assert merged.n_nodes == ref.n_nodes
    rloc, mloc = ref.locations(locs), merged.locations(locs)
    ridx = {n: k for k, n in enumerate(rloc)}
    assert set(rloc) == set(mloc)
    for k, name in enumerate(mloc):
        j = ridx[name]
        assert np.array_equal(merged.vertices[k], ref.vertices[j])      # bit-identical
        assert merged.is_leaf(k) == ref.is_leaf(j)
        assert (merged.flags[k] & 1) == (ref.flags[j] & 1)
        assert not (merged.flags[k] & 4)
        assert np.allclose(merged.vertex_costs[k], ref.vertex_costs[j], rtol=1e-9, atol=1e-12)
    # each closed leaf is closed by exactly one share (the replicated top has none at this size
    # beyond what rank 0 is credited with)
    own = [int(p.info['n_closed']) - int(p.info['replicated_closed']) * (r > 0)
           for r, p in enumerate(parts)]
    assert sum(own) == int(np.sum((ref.flags & 1) > 0))


def test_budgeted_persistent_rounds_from_a_single_source():
    """
    ehm_partition_advance: rounds of the PERSISTENT frontier kernel with a pop budget; what is
    left of its device queue is the frontier the ranks rebalance.  Rank 0 owns the roots, rank 1
    starts empty (shard_min_frontier < 0) and is fed by the first rounds; nothing is replicated.
    """
    from explicit_hybrid_mpc_amd import distributed, engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    mpc = helpers.make_instance('lin', 0)
    can = mpc.compile()
    V = examples.box_vertices(examples.theta_box(mpc))
    roots, locs = ehm_tools.delaunay_roots(V)
    gps = [engine.GpuProblem(can, 1., 1.) for _ in range(3)]
    eps_a = float(np.max(gps[0].solve_pt(0.12 * V)[0]))
    for g in gps:
        g.set_eps(eps_a, 1e-2)
    ref = gps[2].partition(roots, action='ecc')
    world = 2
    runs = [gps[r].begin(roots, shard=(r, world, -1)) for r in range(world)]
    assert runs[1].advance(10) == 0                     # nothing to do yet
    logs = [[] for _ in range(world)]
    rnd, moved, budget = 0, 0, 64
    while True:
        counts = [run.advance(budget) for run in runs]
        budget = min(2 * budget, 4096)
        if sum(counts) == 0:
            break
        for donor, receiver, n in distributed.balance_plan(counts, tolerance=0.02, min_move=4):
            ids, rec, meta = runs[donor].take(n)
            first = runs[receiver].give(rec, meta)
            logs[donor].append(dict(kind='give', round=rnd, peer=receiver, ids=ids))
            logs[receiver].append(dict(kind='recv', round=rnd, peer=donor, first=first, count=n))
            moved += n
        rnd += 1
    parts = [run.finish(export=True) for run in runs]
    for g in gps:
        g.close()
    assert moved > 0 and rnd >= 3
    assert parts[1].info['n_closed'] > 0.2 * ref.info['n_closed']      # rank 1 really worked
    received = distributed.resolve_received(parts, logs, locs)
    merged = distributed.merge_flat(parts, locs, received)
    assert merged.n_nodes == ref.n_nodes
    rloc, mloc = ref.locations(locs), merged.locations(locs)
    ridx = {n: k for k, n in enumerate(rloc)}
    assert set(rloc) == set(mloc)
    for k, name in enumerate(mloc):
        j = ridx[name]
        assert np.array_equal(merged.vertices[k], ref.vertices[j])
        assert merged.is_leaf(k) == ref.is_leaf(j)
        assert (merged.flags[k] & 1) == (ref.flags[j] & 1)
        assert not (merged.flags[k] & 4)
        assert np.allclose(merged.vertex_costs[k], ref.vertex_costs[j], rtol=1e-9, atol=1e-12)
    assert sum(int(p.info['n_closed']) for p in parts) == int(np.sum((ref.flags & 1) > 0))


def test_hybrid_nodes_travel_with_their_bit_rows():
    """
    take / give on the multi-commutation engine: a node moves with its feasibility rows,
    candidate set, inherited verdicts and blacklist.  Rank 0 owns the roots, rank 1 starts empty;
    the merged shares are the tree one run grows (commutations included).
    """
    from explicit_hybrid_mpc_amd import distributed, engine
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    roots = np.array(roots)
    gps = [engine.GpuProblem(mpc.compile(), eps_a, 0.2) for _ in range(3)]
    ref = gps[2].partition(roots, action='ecc')
    world = 2
    runs = [gps[r].begin(roots, shard=(r, world, -1)) for r in range(world)]
    assert runs[0].nrec > 2 * 3 + 3 + 3          # record + bit rows
    logs = [[] for _ in range(world)]
    rnd = moved = 0
    while True:
        counts = [run.step(1) for run in runs]
        if sum(counts) == 0:
            break
        mov = [run.movable() for run in runs]
        for donor, receiver, n in distributed.balance_plan(mov, tolerance=0.02, min_move=2):
            ids, rec, meta = runs[donor].take(n)
            first = runs[receiver].give(rec, meta)
            logs[donor].append(dict(kind='give', round=rnd, peer=receiver, ids=ids))
            logs[receiver].append(dict(kind='recv', round=rnd, peer=donor, first=first, count=n))
            moved += n
        rnd += 1
    parts = [run.finish(export=True) for run in runs]
    for g in gps:
        g.close()
    assert moved > 0 and parts[1].info['n_closed'] > 0
    received = distributed.resolve_received(parts, logs, locs)
    merged = distributed.merge_flat(parts, locs, received)
    assert merged.n_nodes == ref.n_nodes
    rloc, mloc = ref.locations(locs), merged.locations(locs)
    ridx = {n: k for k, n in enumerate(rloc)}
    assert set(rloc) == set(mloc)
    for k, name in enumerate(mloc):
        j = ridx[name]
        assert np.array_equal(merged.vertices[k], ref.vertices[j])
        assert merged.is_leaf(k) == ref.is_leaf(j)
        assert (merged.flags[k] & 3) == (ref.flags[j] & 3)
        assert merged.delta_idx[k] == ref.delta_idx[j]
        assert np.allclose(merged.vertex_costs[k], ref.vertex_costs[j], rtol=1e-9, atol=1e-12)
    assert sum(int(p.info['n_closed']) for p in parts) == int(np.sum((ref.flags & 1) > 0))
