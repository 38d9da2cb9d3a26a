"""
The N > 1 path on CPU: two processes under torch.distributed/gloo.  The engine itself needs
a GPU, so each rank's share of the tree is produced by a CPU emulation of the engine's
sweep order and dealing rule (frontier position k -> rank k % world, children appended in
open-list order); what is under test is the product's distributed layer: the dealing rule,
the counter collectives and the merge of the ranks' trees by location.
"""

import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import helpers


def sweep_partition_cpu(mpc, eps_a, eps_r, roots, rank, world, min_frontier):
    """Level-synchronous single-commutation partition with the engine's sharding rule."""
    from explicit_hybrid_mpc_amd.engine import FlatTree
    from explicit_hybrid_mpc_amd import distributed
    from oracle.oracle_cpu import OracleCPU
    from oracle import geometry
    orc = OracleCPU(mpc, eps_a, eps_r)
    orc.memoize = True
    d0 = orc.deltas[0]
    V, L, Rr, C, U, F, T = [], [], [], [], [], [], []

    def add(R, c, u):
        V.append(R); L.append(-1); Rr.append(-1); C.append(c); U.append(u); F.append(2); T.append(0.)
        return len(L) - 1
    frontier = []
    for R in roots:
        sol = [orc.P_theta_delta(v, d0) for v in R]
        frontier.append(add(np.array(R), np.array([s[1] for s in sol]),
                            np.array([s[0] for s in sol])))
    sharded = world == 1
    while frontier:
        if not sharded and len(frontier) >= min_frontier:
            keep = []
            for k, i in enumerate(frontier):
                if distributed.owner_of(k, world) == rank:
                    keep.append(i)
                else:
                    F[i] |= 4
            frontier, sharded = keep, True
        nxt = []
        for i in frontier:
            t, _ = orc.slack(V[i], C[i], 0)
            T[i] = t
            if not (t >= 0.):
                F[i] |= 1
                continue
            S1, S2, (a, b) = geometry.split_along_longest_edge(V[i])
            u_mid, J_mid, _ = orc.P_theta_delta(S1[a], d0)
            c1, c2, u1, u2 = C[i].copy(), C[i].copy(), U[i].copy(), U[i].copy()
            c1[a], c2[b], u1[a], u2[b] = J_mid, J_mid, u_mid, u_mid
            L[i] = add(S1, c1, u1)
            Rr[i] = add(S2, c2, u2)
            nxt += [L[i], Rr[i]]
        frontier = nxt
    info = dict(n_roots=len(roots), n_nodes=len(L), lp_solves=orc.n_solves,
                n_closed=int(np.sum(np.array(F) & 1 > 0)))
    return FlatTree(np.array(V), np.array(L, dtype=np.int32), np.array(Rr, dtype=np.int32),
                    np.zeros(len(L), dtype=np.int32), np.array(C), np.array(U),
                    np.array(F, dtype=np.uint8), np.array(T), info, np.ones((1, mpc.N)))


class CpuRun:
    """
    CPU emulation of engine.PartitionRun (begin/step/take/give/finish) with the engine's sweep
    order, dealing rule and hand-over semantics, so that distributed.run_balanced can be driven
    on a box without a GPU.
    """

    def __init__(self, mpc, eps_a, eps_r, roots, shard):
        from oracle.oracle_cpu import OracleCPU
        self.mpc = mpc
        self.orc = OracleCPU(mpc, eps_a, eps_r)
        self.orc.memoize = True
        self.d0 = self.orc.deltas[0]
        self.rank, self.world, self.min_frontier = shard if shard is not None else (0, 1, 0)
        self.sharded = self.world == 1
        self.V, self.L, self.Rr, self.C, self.U, self.F, self.T = [], [], [], [], [], [], []
        self.n_roots = len(roots)
        self.frontier = []
        for R in roots:
            sol = [self.orc.P_theta_delta(v, self.d0) for v in R]
            self.frontier.append(self._add(np.array(R), np.array([s[1] for s in sol]),
                                           np.array([s[0] for s in sol]), 2))
        p, n_u = np.array(roots[0]).shape[1], self.U[0].shape[1]
        self.p, self.n_u = p, n_u
        self.nrec = (p + 1) * p + (p + 1) + (p + 1) * n_u
        if self.world > 1 and self.min_frontier < 0:
            # dynamic balancing from a single source (ehm_run_opts.shard_min_frontier < 0):
            # rank 0 owns the roots, the others start empty
            self.sharded = True
            if self.rank > 0:
                for i in self.frontier:
                    self.F[i] |= 4
                self.frontier = []

    def _add(self, R, c, u, flag):
        self.V.append(R); self.L.append(-1); self.Rr.append(-1); self.C.append(c)
        self.U.append(u); self.F.append(flag); self.T.append(0.)
        return len(self.L) - 1

    def step(self, max_sweeps=0):
        from explicit_hybrid_mpc_amd import distributed
        from oracle import geometry
        done = 0
        while self.frontier and (max_sweeps <= 0 or done < max_sweeps):
            if not self.sharded and len(self.frontier) >= self.min_frontier:
                keep = []
                for k, i in enumerate(self.frontier):
                    if distributed.owner_of(k, self.world) == self.rank:
                        keep.append(i)
                    else:
                        self.F[i] |= 4
                self.frontier, self.sharded = keep, True
            nxt = []
            for i in self.frontier:
                t, _ = self.orc.slack(self.V[i], self.C[i], 0)
                self.T[i] = t
                if not (t >= 0.):
                    self.F[i] |= 1
                    continue
                S1, S2, (a, b) = geometry.split_along_longest_edge(self.V[i])
                u_mid, J_mid, _ = self.orc.P_theta_delta(S1[a], self.d0)
                c1, c2 = self.C[i].copy(), self.C[i].copy()
                u1, u2 = self.U[i].copy(), self.U[i].copy()
                c1[a], c2[b], u1[a], u2[b] = J_mid, J_mid, u_mid, u_mid
                self.L[i] = self._add(S1, c1, u1, 2)
                self.Rr[i] = self._add(S2, c2, u2, 2)
                nxt += [self.L[i], self.Rr[i]]
            self.frontier = nxt
            done += 1
        return len(self.frontier)

    def advance(self, max_pops=0):
        """Emulation of ehm_partition_advance: the frontier is a FIFO queue, max_pops visits."""
        from oracle import geometry
        queue, done = list(self.frontier), 0
        while queue and (max_pops <= 0 or done < max_pops):
            i = queue.pop(0)
            done += 1
            t, _ = self.orc.slack(self.V[i], self.C[i], 0)
            self.T[i] = t
            if not (t >= 0.):
                self.F[i] |= 1
                continue
            S1, S2, (a, b) = geometry.split_along_longest_edge(self.V[i])
            u_mid, J_mid, _ = self.orc.P_theta_delta(S1[a], self.d0)
            c1, c2 = self.C[i].copy(), self.C[i].copy()
            u1, u2 = self.U[i].copy(), self.U[i].copy()
            c1[a], c2[b], u1[a], u2[b] = J_mid, J_mid, u_mid, u_mid
            self.L[i] = self._add(S1, c1, u1, 2)
            self.Rr[i] = self._add(S2, c2, u2, 2)
            queue += [self.L[i], self.Rr[i]]
        self.frontier = queue
        return len(self.frontier)

    def take(self, count):
        # ehm_partition_take: every s-th entry of the frontier, s = size / count (the newest
        # `count` entries if s < 2); what stays keeps its order
        nf = len(self.frontier)
        s = nf // count if count else 0
        if s >= 2:
            taken = [self.frontier[i * s] for i in range(count)]
            keep = [self.frontier[i * s + k] for i in range(count) for k in range(1, s)] + \
                self.frontier[count * s:]
            self.frontier = keep + taken
        ids = np.array(self.frontier[len(self.frontier) - count:], dtype=np.int32)
        self.frontier = self.frontier[:len(self.frontier) - count]
        rec = np.array([np.concatenate([self.V[i].ravel(), self.C[i], self.U[i].ravel()])
                        for i in ids]).reshape(count, self.nrec)
        for i in ids:
            self.F[i] |= 4
        return ids, rec, np.zeros((count, 2), dtype=np.int32)

    def give(self, records, meta):
        p, n_u = self.p, self.n_u
        first = len(self.L)
        for r in np.asarray(records).reshape(-1, self.nrec):
            R = r[:(p + 1) * p].reshape(p + 1, p)
            c = r[(p + 1) * p:(p + 1) * p + p + 1]
            u = r[(p + 1) * p + p + 1:].reshape(p + 1, n_u)
            self.frontier.append(self._add(R.copy(), c.copy(), u.copy(), 2 | 32))
        return first

    def finish(self, export=True):
        from explicit_hybrid_mpc_amd.engine import FlatTree
        F = np.array(self.F, dtype=np.uint8)
        info = dict(n_roots=self.n_roots, n_nodes=len(self.L), lp_solves=self.orc.n_solves,
                    n_closed=int(np.sum(F & 1 > 0)))
        return FlatTree(np.array(self.V), np.array(self.L, dtype=np.int32),
                        np.array(self.Rr, dtype=np.int32), np.zeros(len(self.L), dtype=np.int32),
                        np.array(self.C), np.array(self.U), F, np.array(self.T), info,
                        np.ones((1, self.mpc.N)))


def _worker_balanced(rank, world, port, out_dir, engine='sweeps'):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import pickle
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import distributed
    distributed.init_process_group('gloo')
    mpc = helpers.make_instance('di', 0)
    roots, locs = helpers.roots_of(mpc)
    settle = engine == 'persistent-settle'
    part, log, rounds = distributed.run_balanced(
        None, roots, min_frontier=6, sweeps_per_round=1, tolerance=0.34 if settle else 0.,
        min_move=1, export=True, engine='persistent' if settle else engine, pops_per_round=3,
        pops_max=24, settle_frontier=3 if settle else 0,
        run_factory=lambda shard: CpuRun(mpc, 0.3, 0.02, roots, shard))
    with open(os.path.join(out_dir, 'bal%d.pkl' % rank), 'wb') as f:
        pickle.dump(dict(part=part, log=log, rounds=rounds), f)
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import pickle
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import distributed
    r, _, w = distributed.init_process_group('gloo')
    assert (r, w) == (rank, world)
    mpc = helpers.make_instance('di', 0)
    eps_a, eps_r = 0.3, 0.02
    roots, locs = helpers.roots_of(mpc)
    spec = distributed.shard_spec(rank, world, min_frontier=8)
    part = sweep_partition_cpu(mpc, eps_a, eps_r, roots, spec[0], spec[1], spec[2])
    mine = int(np.sum((part.flags & 1) > 0))
    tot, mx = distributed.allreduce_counters([mine, part.info['lp_solves']])
    counts = distributed.allgather_counts([mine, part.n_nodes])
    assert counts.shape == (world, 2) and counts[rank, 0] == mine
    assert tot[0] == counts[:, 0].sum() and mx[0] == counts[:, 0].max()
    with open(os.path.join(out_dir, 'part%d.pkl' % rank), 'wb') as f:
        pickle.dump(dict(part=part, tot=tot, imb=distributed.imbalance(counts[:, 0])), f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_tile_the_tree(tmp_path):
    import pickle
    from explicit_hybrid_mpc_amd import distributed
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [pickle.load(open(str(tmp_path / ('part%d.pkl' % r)), 'rb')) for r in range(2)]
    mpc = helpers.make_instance('di', 0)
    roots, locs = helpers.roots_of(mpc)
    full = sweep_partition_cpu(mpc, 0.3, 0.02, roots, 0, 1, 0)
    merged = distributed.merge_flat([o['part'] for o in outs], locs)
    assert merged.n_nodes == full.n_nodes
    floc, mloc = full.locations(locs), merged.locations(locs)
    fidx = {n: k for k, n in enumerate(floc)}
    assert set(floc) == set(mloc)
    for k, name in enumerate(mloc):
        j = fidx[name]
        assert np.array_equal(merged.vertices[k], full.vertices[j])
        assert merged.is_leaf(k) == full.is_leaf(j)
        assert (merged.flags[k] & 1) == (full.flags[j] & 1)
        assert not (merged.flags[k] & 4)
        assert np.allclose(merged.vertex_costs[k], full.vertex_costs[j], rtol=1e-9, atol=1e-12)
    assert outs[0]['tot'][0] == full.info['n_closed']
    assert 1.0 <= outs[0]['imb'] < 1.6
    # each rank really did only part of the work
    assert all(o['part'].info['n_closed'] < full.info['n_closed'] for o in outs)


def test_balance_plan_is_deterministic_and_evens_out():
    from explicit_hybrid_mpc_amd import distributed
    assert distributed.balance_plan([100, 100, 100, 100]) == []
    assert distributed.balance_plan([0, 0]) == []
    assert distributed.balance_plan([40, 3], min_move=64) == []       # too short to pay
    counts = [4000, 10, 900, 1200, 0, 3100, 50, 740]
    plan = distributed.balance_plan(counts, tolerance=0.05)
    assert plan == distributed.balance_plan(list(counts), tolerance=0.05)
    after = list(counts)
    for d, r, n in plan:
        assert n > 0 and d != r and after[d] >= n
        after[d] -= n
        after[r] += n
    assert sum(after) == sum(counts)
    assert max(after) <= 1.05 * (sum(counts) / 8.) + 16
    donors = {d for d, _, _ in plan}
    assert donors.isdisjoint({r for _, r, _ in plan})                 # nobody relays


@pytest.mark.parametrize('engine', ['sweeps', 'persistent', 'persistent-settle'])
def test_two_ranks_rebalance_and_merge(tmp_path, engine):
    """
    run_balanced over gloo: frontier nodes really move, and the merged tree is the tree.
    'sweeps': sweep rounds after a deal by position; 'persistent': budgeted rounds of the
    (emulated) persistent kernel from a single source -- rank 1 starts with nothing;
    'persistent-settle': the same until a round ends balanced with enough nodes on both ranks,
    then one unbudgeted launch each (run_balanced's settle_frontier).
    """
    import pickle
    from explicit_hybrid_mpc_amd import distributed
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_balanced, args=(2, port, str(tmp_path), engine), nprocs=2, join=True)
    outs = [pickle.load(open(str(tmp_path / ('bal%d.pkl' % r)), 'rb')) for r in range(2)]
    moved = sum(len(e['ids']) for o in outs for e in o['log'] if e['kind'] == 'give')
    got = sum(e['count'] for o in outs for e in o['log'] if e['kind'] == 'recv')
    assert moved == got and moved > 0
    settles = [[e for e in o['log'] if e['kind'] == 'settle'] for o in outs]
    if engine == 'persistent-settle':
        # both ranks took the decision in the same round on the same counts, and the launch
        # after it was the last
        assert len(settles[0]) == 1 and settles[0] == settles[1]
        assert min(settles[0][0]['counts']) >= 3
        assert outs[0]['rounds'] == outs[1]['rounds'] == settles[0][0]['round'] + 1
    else:
        assert not settles[0] and not settles[1]
    mpc = helpers.make_instance('di', 0)
    roots, locs = helpers.roots_of(mpc)
    full = sweep_partition_cpu(mpc, 0.3, 0.02, roots, 0, 1, 0)
    parts = [o['part'] for o in outs]
    received = distributed.resolve_received(parts, [o['log'] for o in outs], locs)
    assert sum(len(r) for r in received) == moved
    merged = distributed.merge_flat(parts, locs, received)
    assert merged.n_nodes == full.n_nodes
    floc, mloc = full.locations(locs), merged.locations(locs)
    fidx = {n: k for k, n in enumerate(floc)}
    assert set(floc) == set(mloc)
    for k, name in enumerate(mloc):
        j = fidx[name]
        assert np.array_equal(merged.vertices[k], full.vertices[j])
        assert merged.is_leaf(k) == full.is_leaf(j)
        assert (merged.flags[k] & 1) == (full.flags[j] & 1)
        assert not (merged.flags[k] & 4)
        assert np.allclose(merged.vertex_costs[k], full.vertex_costs[j], rtol=1e-9, atol=1e-12)


class _FailingRun(CpuRun):
    """Rank 1's engine fails in its third round (the pool-exhausted / numeric error of a device)."""

    def step(self, max_sweeps=0):
        self.calls = getattr(self, 'calls', 0) + 1
        if self.rank == 1 and self.calls == 3:
            raise RuntimeError('libehmpc error -4: node pool exhausted (emulated)')
        return super().step(max_sweeps)

    def free_nodes(self):
        return 1 << 20


def _worker_failing(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import distributed
    distributed.init_process_group('gloo')
    mpc = helpers.make_instance('di', 0)
    roots, locs = helpers.roots_of(mpc)
    try:
        distributed.run_balanced(
            None, roots, min_frontier=6, sweeps_per_round=1, tolerance=0., min_move=1,
            run_factory=lambda shard: _FailingRun(mpc, 0.3, 0.02, roots, shard))
        outcome = 'finished'
    except RuntimeError as e:
        outcome = 'raised: %s' % e
    with open(os.path.join(out_dir, 'fail%d.txt' % rank), 'w') as f:
        f.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_rank_stops_every_rank(tmp_path):
    """
    run_balanced all-gathers a status word next to the frontier size: when one rank's engine
    raises, every rank leaves the loop with an error instead of waiting in a collective for ever.
    """
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_failing, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [open(str(tmp_path / ('fail%d.txt' % r))).read() for r in range(2)]
    assert outs[1].startswith('raised: libehmpc error -4')
    assert outs[0].startswith('raised: partition run failed on rank(s) [1]')


class _FailingTake(CpuRun):
    """The donor's hand-over fails in the middle of a plan (a device error inside ehm_partition_take)."""

    def take(self, count):
        raise RuntimeError('libehmpc error -2: take failed (emulated)')

    def free_nodes(self):
        return 1 << 20


def _worker_failing_take(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import distributed
    distributed.init_process_group('gloo')
    mpc = helpers.make_instance('di', 0)
    roots, locs = helpers.roots_of(mpc)
    try:
        distributed.run_balanced(
            None, roots, min_frontier=-1, sweeps_per_round=1, tolerance=0., min_move=1,
            run_factory=lambda shard: _FailingTake(mpc, 0.3, 0.02, roots, shard))
        outcome = 'finished'
    except RuntimeError as e:
        outcome = 'raised: %s' % e
    with open(os.path.join(out_dir, 'take%d.txt' % rank), 'w') as f:
        f.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_hand_over_does_not_leave_the_receiver_waiting(tmp_path):
    """
    A donor whose take() raises mid-plan still completes its planned send (a poison block), so
    the receiver is not left in dist.recv for ever; both ranks then fail together.
    """
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_failing_take, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [open(str(tmp_path / ('take%d.txt' % r))).read() for r in range(2)]
    assert all(o.startswith('raised:') for o in outs), outs
    assert any('take failed' in o for o in outs)


def _worker_roots(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import pickle
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier, distributed
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    from oracle import prefix_bb, geometry
    distributed.init_process_group('gloo')
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    orc = bnb.PrefixOracle(mpc, eps_a, 0.2, table=prefix_bb.CpuPrefixTable(mpc))
    roots, locs = helpers.roots_of(mpc)
    trees = [Tree(NodeData(vertices=np.array(R))) for R in roots]

    def split_batch(R):
        out = [geometry.split_along_longest_edge(r) for r in R]
        return (np.array([o[0] for o in out]), np.array([o[1] for o in out]),
                np.array([o[2] for o in out], dtype=np.int32))
    trees, stats, counts = distributed.grow_roots_sharded(orc, trees, 'ecc', handoff=False,
                                                          split_batch=split_batch)
    mine = {k: [(loc, nd.is_leaf(), nd.data.is_epsilon_suboptimal) for nd, loc in t.walk(locs[k])]
            for k, t in enumerate(trees) if k % world == rank}
    with open(os.path.join(out_dir, 'roots%d.pkl' % rank), 'wb') as f:
        pickle.dump(dict(mine=mine, counts=counts, stats=stats), f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_the_roots_of_the_search_driver(tmp_path):
    """distributed.grow_roots_sharded: the roots of the set dealt over the ranks, each rank's
    roots grown by the frontier-wide search driver (CPU stand-in for the device table); the
    union is the enumerating CPU partition."""
    import pickle
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_roots, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [pickle.load(open(str(tmp_path / ('roots%d.pkl' % r)), 'rb')) for r in range(2)]
    mpc = helpers.make_instance('pwa_small', 0)
    roots, locs = helpers.roots_of(mpc)
    cpu = PartitionCPU(OracleCPU(mpc, helpers.eps_a_rule(mpc, 0.25), 0.2))
    cpu.run(roots, locs, 'ecc')
    got = {}
    for o in outs:
        for k, nodes in o['mine'].items():
            for loc, leaf, closed in nodes:
                got[loc] = (leaf, closed)
    assert set(got) == set(cpu.nodes)
    for loc, nd in cpu.nodes.items():
        assert got[loc] == (nd['leaf'], nd['is_epsilon_suboptimal'])
    counts = outs[0]['counts']
    assert counts.shape == (2, 3) and counts[:, 2].sum() == len(roots)
    assert counts[:, 1].sum() == sum(1 for nd in cpu.nodes.values() if nd['leaf'])
    assert np.array_equal(outs[0]['counts'], outs[1]['counts'])
    assert all(c > 0 for c in counts[:, 0])


def _worker_roots_dynamic(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import pickle
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import bnb, distributed, frontier
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    from oracle import prefix_bb, geometry
    distributed.init_process_group('gloo')
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    trees = [Tree(NodeData(vertices=np.array(R))) for R in roots]

    def split_batch(R):
        out = [geometry.split_along_longest_edge(r) for r in R]
        return (np.array([o[0] for o in out]), np.array([o[1] for o in out]),
                np.array([o[2] for o in out], dtype=np.int32))
    # the native driver on the CPU statement of the table (no device in this test)
    solvers = frontier.TableSolvers(prefix_bb.CpuPrefixTable(mpc, eps_a, 0.2), split_batch)
    nat = frontier.NativeFrontier(mpc, eps_a, 0.2, solvers=solvers)
    slow = bnb.PrefixOracle(mpc, eps_a, 0.2, table=prefix_bb.CpuPrefixTable(mpc, eps_a, 0.2))
    trees, stats, counts = distributed.grow_roots_sharded(
        slow, trees, 'ecc', deal='dynamic', native=nat, native_opts=dict(round_cap=16),
        handoff=False, split_batch=split_batch)
    mine = {k: [(loc, nd.is_leaf(), nd.data.is_epsilon_suboptimal)
                for nd, loc in trees[k].walk(locs[k])] for k in stats['mine']}
    with open(os.path.join(out_dir, 'dyn%d.pkl' % rank), 'wb') as f:
        pickle.dump(dict(mine=mine, counts=counts, owned=stats['mine']), f)
    nat.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_claim_the_roots_of_the_native_driver_dynamically(tmp_path):
    """deal='dynamic': the ranks claim roots from a counter in the process group's store and grow
    them with the NATIVE driver; every root is grown exactly once and the union is the enumerating
    CPU partition (= the one-rank tree)."""
    import pickle
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_roots_dynamic, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [pickle.load(open(str(tmp_path / ('dyn%d.pkl' % r)), 'rb')) for r in range(2)]
    mpc = helpers.make_instance('pwa_small', 0)
    roots, locs = helpers.roots_of(mpc)
    owned = sorted(outs[0]['owned'] + outs[1]['owned'])
    assert owned == list(range(len(roots)))              # each root claimed exactly once
    cpu = PartitionCPU(OracleCPU(mpc, helpers.eps_a_rule(mpc, 0.25), 0.2))
    cpu.run(roots, locs, 'ecc')
    got = {}
    for o in outs:
        for k, nodes in o['mine'].items():
            for loc, leaf, closed in nodes:
                got[loc] = (leaf, closed)
    assert set(got) == set(cpu.nodes)
    for loc, nd in cpu.nodes.items():
        assert got[loc] == (nd['leaf'], nd['is_epsilon_suboptimal'])
    counts = outs[0]['counts']
    assert counts[:, 2].sum() == len(roots)
    assert counts[:, 1].sum() == sum(1 for nd in cpu.nodes.values() if nd['leaf'])


def test_claims_emulated_over_measured_root_times():
    from explicit_hybrid_mpc_amd import distributed
    busy, ratio = distributed.emulate_claims([1.0] * 64, 8)
    assert abs(ratio - 1.0) < 1e-12 and len(busy) == 8
    # one slow root among many: whoever claims it keeps it, the others take the rest
    busy, ratio = distributed.emulate_claims([100.0] + [1.0] * 700, 8)
    assert max(busy) == 100.0 and ratio < 1.01


def _worker_roots_stealing(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import pickle
    import time
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import bnb, distributed, frontier
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    from oracle import prefix_bb, geometry
    distributed.init_process_group('gloo')
    mpc = helpers.make_instance('pwa_small', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    trees = [Tree(NodeData(vertices=np.array(R))) for R in roots]

    def split_batch(R):
        out = [geometry.split_along_longest_edge(r) for r in R]
        return (np.array([o[0] for o in out]), np.array([o[1] for o in out]),
                np.array([o[2] for o in out], dtype=np.int32))
    solvers = frontier.TableSolvers(prefix_bb.CpuPrefixTable(mpc, eps_a, 0.2), split_batch)
    nat = frontier.NativeFrontier(mpc, eps_a, 0.2, solvers=solvers)
    slow = bnb.PrefixOracle(mpc, eps_a, 0.2, table=prefix_bb.CpuPrefixTable(mpc, eps_a, 0.2))
    if rank == 1:
        time.sleep(1.0)         # rank 0 claims the one batch that holds every root
    trees, stats, counts = distributed.grow_roots_sharded(
        slow, trees, 'ecc', deal='dynamic', native=nat, batch=len(roots), steal=True,
        steal_slice=3, native_opts=dict(round_cap=2), handoff=False, split_batch=split_batch)
    mine = {k: [(loc, nd.is_leaf(), nd.data.is_epsilon_suboptimal,
                 getattr(nd.data, 'remote', False))
                for nd, loc in trees[k].walk(locs[k])] for k in stats['mine']}
    with open(os.path.join(out_dir, 'steal%d.pkl' % rank), 'wb') as f:
        pickle.dump(dict(mine=mine, counts=counts, owned=stats['mine'],
                         stats={k: v for k, v in stats.items() if k != 'mine'}), f)
    nat.close()
    dist.barrier()
    dist.destroy_process_group()


def test_an_idle_rank_takes_cells_from_a_busy_one(tmp_path):
    """
    deal='dynamic', steal=True (distributed.CellExchange on ehm_frontier_take / _give): rank 0
    claims EVERY root in one batch, rank 1 finds none and asks; rank 0 answers between two slices
    of its run with half of its pending cells, more than once; the sub-trees rank 1 grows return
    and are attached to rank 0's trees, which must be the enumerating CPU partition -- no leaf
    still marked remote, every leaf counted once over the two ranks.
    """
    import pickle
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_roots_stealing, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [pickle.load(open(str(tmp_path / ('steal%d.pkl' % r)), 'rb')) for r in range(2)]
    mpc = helpers.make_instance('pwa_small', 0)
    roots, locs = helpers.roots_of(mpc)
    assert outs[0]['owned'] == list(range(len(roots))) and outs[1]['owned'] == []
    s0, s1 = outs[0]['stats'], outs[1]['stats']
    assert s0['parcels_given'] >= 2 and s1['cells_adopted'] >= 2
    assert s0['subtrees_attached'] == s1['cells_adopted'] + s0['cells_adopted']
    assert s1['regions'] > 0 and s0['regions'] > 0
    cpu = PartitionCPU(OracleCPU(mpc, helpers.eps_a_rule(mpc, 0.25), 0.2))
    cpu.run(roots, locs, 'ecc')
    got = {}
    for k, nodes in outs[0]['mine'].items():
        for loc, leaf, closed, remote in nodes:
            assert not remote, loc
            got[loc] = (leaf, closed)
    assert set(got) == set(cpu.nodes)
    for loc, nd in cpu.nodes.items():
        assert got[loc] == (nd['leaf'], nd['is_epsilon_suboptimal'])
    closed = sum(1 for nd in cpu.nodes.values() if nd['leaf'] and nd['is_epsilon_suboptimal'])
    assert s0['regions'] + s1['regions'] == closed


def test_cell_exchange_protocol_over_a_store():
    """distributed.CellExchange on a HashStore, three 'ranks' as threads: rank 0 holds 40 units of
    work and serves between units; ranks 1 and 2 start idle and ask.  Every unit is done exactly
    once, takers serve each other too, and everybody leaves once all are idle with nothing on its
    way."""
    import threading
    import time
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import distributed
    store = dist.HashStore()
    world = 3
    done, errors = [], []
    lock = threading.Lock()

    def rank_main(rank):
        try:
            ex = distributed.CellExchange(store, rank, world, key='t/cells', poll=0.0005)
            work = list(range(40)) if rank == 0 else []

            def take():
                if len(work) < 2:
                    return None
                half = [work.pop() for _ in range(len(work) // 2)]
                return dict(node=np.array(half))

            while True:
                while work:
                    unit = work.pop()
                    time.sleep(0.002)
                    with lock:
                        done.append((rank, unit))
                    ex.serve(take)
                parcel = ex.wait_for_work()
                if parcel is None:
                    return
                assert parcel['id'][0] != rank
                work.extend(int(u) for u in parcel['node'])
        except BaseException as e:
            errors.append(e)
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)
    assert sorted(u for _, u in done) == list(range(40))
    assert {r for r, _ in done} == {0, 1, 2}              # everybody got some
    assert int(store.add('t/cells/idle', 0)) == world


def test_cell_exchange_a_failing_rank_releases_the_waiting_ones():
    """A rank whose driver raises calls CellExchange.fail(): the ranks that wait for cells raise too
    instead of waiting for it to become idle."""
    import threading
    import time
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import distributed
    store = dist.HashStore()
    world = 3
    outcome = {}

    def rank_main(rank):
        ex = distributed.CellExchange(store, rank, world, key='f/cells', poll=0.0005)
        try:
            if rank == 0:
                for _ in range(5):
                    time.sleep(0.01)
                    ex.serve(lambda: None)          # busy, nothing to give yet
                ex.fail()
                outcome[rank] = 'failed'
                return
            outcome[rank] = ex.wait_for_work()
        except RuntimeError as e:
            outcome[rank] = str(e)
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=30)
    assert not any(t.is_alive() for t in threads)
    assert outcome[0] == 'failed'
    assert 'another rank failed' in outcome[1] and 'another rank failed' in outcome[2]
