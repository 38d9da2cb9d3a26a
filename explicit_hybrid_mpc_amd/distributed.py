"""
Multi-GPU layer: one process per GPU under ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on the GPU node, "gloo" in the CPU tests).

The reference distributes work with an MPI master/worker star and pickled subtrees
(lib/scheduler.py:498-599, lib/worker.py:187-239).  Here the live frontier is sharded instead:
every rank grows the same top of the tree, and once a sweep's frontier is wide enough each
rank keeps the positions ``k % world == rank`` (``ehm_run_opts.shard_*``).  Subtrees are
independent, so the data path itself needs no collective.  Because subtree sizes are very
uneven, ``run_balanced`` rebalances as it goes: every few sweeps the
ranks all-gather their frontier sizes (one int64 per rank over RCCL), derive the same
donor -> receiver plan from them, and the donors send node records point to point (a pair talks
over its own xGMI link; nothing is ringed).  The other collectives carry counters (work
accounting, load-imbalance report) and, on request, the ranks' finished subtrees to rank 0,
where they are grafted by location -- the contract of ``build_tree``
(lib/scheduler.py:644-691).
"""

import os
import numpy as np

from .engine import FlatTree


def env_rank_world():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init_process_group(backend='nccl'):
    """Initialise torch.distributed from the torchrun environment (127.0.0.1 rendezvous)."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            import torch
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_spec(rank, world, min_frontier=2048):
    """Argument for ``GpuProblem.partition(shard=...)``; None on a single rank."""
    return None if world <= 1 else (int(rank), int(world), int(min_frontier))


def deal_depth_for(n_roots, world, per_rank=2048):
    """
    Tree depth at which a sharded persistent launch deals its nodes (ehm_run_opts.deal_depth):
    the first depth that can hold ``per_rank`` nodes per rank (n_roots * 2^depth >= per_rank *
    world) -- at 2048 per rank the shares of the bench tree balance to 2.4 % at 8 ranks
    (profiles/r2/shard_balance_deal_*.txt).
    """
    import math
    if world <= 1:
        return 0
    return max(1, int(math.ceil(math.log2(max(1., per_rank * world / float(n_roots))))))


def owner_of(position, world):
    """Rank that keeps frontier position ``position`` (the rule of k_shard_filter)."""
    return position % world


def allreduce_counters(values, device=None):
    """
    Sum and max over ranks of a vector of float64 counters.  Returns (sum, max) numpy arrays.
    """
    import torch
    import torch.distributed as dist
    t = torch.tensor(np.asarray(values, dtype=np.float64), dtype=torch.float64,
                     device=device if device is not None else 'cpu')
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        v = t.cpu().numpy()
        return v, v.copy()
    s, m = t.clone(), t.clone()
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return s.cpu().numpy(), m.cpu().numpy()


def allgather_counts(local_counts, device=None):
    """
    All-gather of a small int64 vector per rank (per-GPU frontier / node counts: the
    64-byte exchange of SURVEY.md section 8e).  Returns an array (world, len(local_counts)).
    """
    import torch
    import torch.distributed as dist
    t = torch.tensor(np.asarray(local_counts, dtype=np.int64), dtype=torch.int64,
                     device=device if device is not None else 'cpu')
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t.cpu().numpy()[None, :]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])


def imbalance(counts):
    """max / mean of the per-rank work counts (1.0 = perfectly balanced)."""
    counts = np.asarray(counts, dtype=np.float64)
    return float(counts.max() / max(counts.mean(), 1e-300))


def balance_plan(counts, tolerance=0.1, min_move=16):
    """
    Deterministic rebalancing plan from the all-gathered frontier sizes: a list of
    (donor, receiver, n) that brings every rank to within `tolerance` of the mean, largest
    surplus matched with largest deficit first.  Empty when the frontiers are already even,
    or too short for a transfer to pay (less than min_move nodes).
    """
    counts = [int(c) for c in counts]
    world = len(counts)
    total = sum(counts)
    if world < 2 or total == 0:
        return []
    mean = total / float(world)
    if max(counts) <= mean * (1. + tolerance) + min_move:
        return []
    target = [total // world + (1 if r < total % world else 0) for r in range(world)]
    surplus = sorted([[counts[r] - target[r], r] for r in range(world) if counts[r] > target[r]],
                     key=lambda e: (-e[0], e[1]))
    deficit = sorted([[target[r] - counts[r], r] for r in range(world) if counts[r] < target[r]],
                     key=lambda e: (-e[0], e[1]))
    plan = []
    i = j = 0
    while i < len(surplus) and j < len(deficit):
        n = min(surplus[i][0], deficit[j][0])
        if n >= min_move:
            plan.append((surplus[i][1], deficit[j][1], n))
        surplus[i][0] -= n
        deficit[j][0] -= n
        if surplus[i][0] == 0:
            i += 1
        if deficit[j][0] == 0:
            j += 1
    return plan


def _exchange(run, plan, rank, device, rnd, log):
    """
    Execute this rank's part of a plan: donors take + send, receivers recv + give.  EVERY planned
    send / recv pair is completed even after a failure on this rank -- a donor whose ``take`` raised
    sends a poison block (NaN meta words) of the planned size, a receiver whose ``give`` raised
    keeps draining its planned receives -- so no peer is left blocked in a point-to-point call; the
    first error is raised once the plan is through, and the status all-gather that follows in
    ``run_balanced`` stops every rank.  (After a failure the run's device tree is gone:
    ``PartitionRun._checked`` releases it; ``run`` cannot be continued.)
    """
    import torch
    import torch.distributed as dist
    nrec = run.nrec
    err = None
    POISON = -(2 ** 31)         # meta word of a block its donor could not fill
    for donor, receiver, n in plan:
        if device is not None:
            # device-resident blocks, peer to peer (ncclSend / ncclRecv over xGMI): the records
            # and the meta words never visit the host (ehm_partition_take / _give take device
            # pointers).  Two messages per block: float64 records, int32 (commutation, depth).
            if rank == donor:
                rec_t = meta_t = None
                if err is None:
                    try:
                        ids, rec_t, meta_t = run.take_device(n, device)
                    except Exception as e:
                        err = e
                if rec_t is None:
                    rec_t = torch.zeros((n, nrec), dtype=torch.float64, device=device)
                    meta_t = torch.full((n, 2), POISON, dtype=torch.int32, device=device)
                dist.send(rec_t, dst=receiver)
                dist.send(meta_t, dst=receiver)
                if err is None:
                    log.append(dict(kind='give', round=rnd, peer=receiver, ids=ids))
            elif rank == receiver:
                rec_t = torch.empty((n, nrec), dtype=torch.float64, device=device)
                meta_t = torch.empty((n, 2), dtype=torch.int32, device=device)
                dist.recv(rec_t, src=donor)
                dist.recv(meta_t, src=donor)
                if bool((meta_t == POISON).any().item()):
                    err = err or RuntimeError('rank %d could not hand over the %d nodes planned '
                                              'for rank %d' % (donor, n, rank))
                elif err is None:
                    try:
                        first = run.give_device(rec_t, meta_t)
                        log.append(dict(kind='recv', round=rnd, peer=donor, first=first, count=n))
                    except Exception as e:
                        err = e
            continue
        # host path (gloo: CPU tensors)
        if rank == donor:
            buf = None
            if err is None:
                try:
                    ids, rec, meta = run.take(n)
                    buf = np.concatenate([rec, meta.astype(np.float64)], axis=1)
                except Exception as e:
                    err = e
            if buf is None:
                buf = np.full((n, nrec + 2), np.nan)
            t = torch.from_numpy(np.ascontiguousarray(buf))
            dist.send(t, dst=receiver)
            if err is None:
                log.append(dict(kind='give', round=rnd, peer=receiver, ids=ids))
        elif rank == receiver:
            t = torch.empty((n, nrec + 2), dtype=torch.float64)
            dist.recv(t, src=donor)
            buf = t.numpy()
            if np.isnan(buf[:, nrec:]).any():
                err = err or RuntimeError('rank %d could not hand over the %d nodes planned for '
                                          'rank %d' % (donor, n, rank))
            elif err is None:
                try:
                    first = run.give(buf[:, :nrec], np.rint(buf[:, nrec:]).astype(np.int32))
                    log.append(dict(kind='recv', round=rnd, peer=donor, first=first, count=n))
                except Exception as e:
                    err = e
    if err is not None:
        raise err


def allgather_floats(values, device=None):
    """(world, len(values)) float64 array of every rank's values."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray([values], dtype=np.float64)
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device or 'cpu')
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy() for o in out])


_STATUS_KEYS = ('volume_filled_total', 'simplex_count_total', 'time_active_total', 'time_idle',
                'time_ecc', 'time_lcss')


def _publish(status, worker, run, live, device, rank, force=False):
    """Every rank's status record to rank 0's MainStatusPublisher (one small all-gather)."""
    worker.absorb(run.progress())
    vals = [worker.data[k] for k in _STATUS_KEYS] + [1. if worker.data['status'] == 'active'
                                                    else 0., float(live)]
    allv = allgather_floats(vals, device=device)
    if rank != 0 or status is None:
        return
    procs = []
    for r, row in enumerate(allv):
        d = dict(worker.data) if r == 0 else dict(worker.data, current_location='')
        for k, v in zip(_STATUS_KEYS, row):
            d[k] = int(v) if k == 'simplex_count_total' else float(v)
        d['simplex_count_current'] = d['simplex_count_total']
        d['status'] = 'active' if row[len(_STATUS_KEYS)] > 0 else 'idle'
        procs.append(d)
    status.update(procs, num_tasks_in_queue=int(allv[:, -1].sum()), force=force)


def run_balanced(gp, roots, action='ecc', init=None, max_nodes=0, min_frontier=1024,
                 sweeps_per_round=3, tolerance=0.1, min_move=16, device=None, export=False,
                 with_volume=False, run_factory=None, status=None, publish_status=False,
                 engine='sweeps', pops_per_round=4096, pops_max=1 << 17, max_depth=0,
                 settle_frontier=4096):
    """
    One partition over all ranks with periodic rebalancing of the live frontiers.
    Returns (FlatTree or info dict of THIS rank's share, transfer log, rounds).
    run_factory(shard) -> object with step/take/give/finish/nrec replaces the GPU engine
    (the CPU tests emulate it).
    publish_status (same value on EVERY rank) adds one all-gather of the ranks' progress
    counters per round; rank 0 passes them to ``status`` (a status.MainStatusPublisher).
    engine='persistent': the rounds are budgeted launches of the persistent frontier kernel
    (``PartitionRun.advance``: pops_per_round node visits, doubling every round up to pops_max)
    instead of sweeps; rank 0 owns the roots, the other ranks start empty and are fed by the
    first rounds -- nothing is replicated.
    settle_frontier (persistent engine, > 0): once a round ends with the frontiers balanced (no
    transfer planned at ``tolerance``) and at least this many nodes on EVERY rank, the ranks --
    which all hold the same all-gathered counts, so they decide alike -- grow what they hold to
    completion in ONE unbudgeted launch.  Budgeted launches queue both children of every split
    (what they leave must be a plain queue slice that ``take`` can cut); the unbudgeted launch
    is the single-GPU engine at full speed (kept child visited from LDS, nodes put back instead
    of waited for), so the rounds cost their overhead only while the work is being spread.
    0 = rebalance to the end (needed where sub-tree sizes under the frontier vary by more than
    the final imbalance one accepts; ``publish_status`` runs never settle: they report per
    round).
    """
    import torch.distributed as dist
    rank, _, world = env_rank_world()
    if not (dist.is_available() and dist.is_initialized()):
        world, rank = 1, 0
    persistent = engine == 'persistent'
    # multi-commutation problems are balanced from a single source too (sweep rounds)
    hybrid = gp is not None and getattr(gp.can, 'n_delta', 1) > 1
    persistent = persistent and not hybrid
    shard = shard_spec(rank, world, -1 if (persistent or hybrid) else min_frontier)
    budget = int(pops_per_round)
    if run_factory is not None:
        run = run_factory(shard)
    else:
        run = gp.begin(roots, action=action, init=init, max_nodes=max_nodes, shard=shard,
                       with_volume=with_volume, max_depth=max_depth)
    log, rnd = [], 0
    worker = None
    if publish_status:
        from . import status as _status
        worker = _status.WorkerStatus(algorithm=action)
        worker.update(active=True)

    def fail_everywhere(err, who):
        """Every rank leaves the loop together: nobody is left waiting in a collective."""
        if worker is not None:
            worker.update(active=False, failed=True)
            if status is not None and rank == 0:
                status.update([worker.data], num_tasks_in_queue=0, force=True)
        if err is not None:
            raise err
        raise RuntimeError('partition run failed on rank(s) %s' % who)

    settled = False
    while True:
        err = None
        try:
            if persistent:
                n = run.advance(budget if ((world > 1 or publish_status) and not settled) else 0)
                budget = min(2 * budget, int(pops_max))
            else:
                n = run.step(sweeps_per_round if (world > 1 or publish_status) else 0)
        except Exception as e:      # reported to the other ranks below, then re-raised
            err, n = e, -1
        if world == 1:
            if err is not None:
                fail_everywhere(err, [0])
            if publish_status:
                if n == 0:
                    worker.update(active=False)
                _publish(status, worker, run, n, device, rank, force=(n == 0))
            if n == 0:
                break
            continue
        # frontier size (-1 = this rank failed) and free pool of every rank: ONE small all-gather
        free = run.free_nodes() if (err is None and hasattr(run, 'free_nodes')) else (1 << 62)
        mov = run.movable() if (err is None and hasattr(run, 'movable')) else max(n, 0)
        info = allgather_counts([n, free, mov], device=device)
        if (info[:, 0] < 0).any():
            fail_everywhere(err, [int(r) for r in np.nonzero(info[:, 0] < 0)[0]])
        if publish_status:
            if n == 0:
                worker.update(active=False)
            _publish(status, worker, run, n, device, rank)
        counts = info[:, 0]
        if counts.sum() == 0:
            break
        # a receiver is never sent more than its pool can take (every rank derives the same plan)
        room = [int(v) for v in info[:, 1]]
        plan = []
        # planned on what can move (multi-commutation runs: nodes that carry data)
        wanted = balance_plan(info[:, 2], tolerance, min_move)
        for donor, receiver, k in wanted:
            k = min(k, max(0, room[receiver] - 2 * int(counts[receiver]) - 64))
            if k >= min_move:
                plan.append((donor, receiver, k))
                room[receiver] -= k
        # settling is decided on the plan BEFORE the receivers' room clips it: a round whose
        # transfers were all dropped for lack of room is not a balanced one
        if (persistent and settle_frontier > 0 and not publish_status and not wanted and
                int(counts.min()) >= int(settle_frontier)):
            settled = True          # every rank sees the same counts: the next launch is the last
            log.append(dict(kind='settle', round=rnd, counts=[int(c) for c in counts]))
            rnd += 1
            continue
        try:
            _exchange(run, plan, rank, device, rnd, log)
        except Exception as e:
            err = e
        bad = allgather_counts([0 if err is None else 1], device=device)[:, 0]
        if bad.any():
            fail_everywhere(err, [int(r) for r in np.nonzero(bad)[0]])
        rnd += 1
    if publish_status and world > 1:
        _publish(status, worker, run, 0, device, rank, force=True)
    return run.finish(export), log, rnd


def resolve_received(parts, logs, root_locations=None):
    """
    Locations of the nodes every rank received: a received root sits where its donor's node
    sat.  parts[r] / logs[r] = FlatTree and transfer log of rank r.  Returns a list of
    {node id: location} per rank (transfers are matched by round, in plan order).
    """
    world = len(parts)
    received = [dict() for _ in range(world)]
    gives = {}      # (round, donor, receiver) -> list of id arrays, in order
    for r in range(world):
        for e in logs[r]:
            if e['kind'] == 'give':
                gives.setdefault((e['round'], r, e['peer']), []).append(e['ids'])
    pending = []
    for r in range(world):
        seen = {}
        for e in logs[r]:
            if e['kind'] == 'recv':
                key = (e['round'], e['peer'], r)
                k = seen.get(key, 0)
                seen[key] = k + 1
                pending.append((r, e['first'], gives[key][k], e['peer']))
    while pending:
        locs = [parts[r].locations(root_locations, received[r]) for r in range(world)]
        rest = []
        for r, first, ids, donor in pending:
            names = [locs[donor][i] for i in ids]
            if any(nm is None for nm in names):
                rest.append((r, first, ids, donor))
                continue
            for k, nm in enumerate(names):
                received[r][first + k] = nm
        if len(rest) == len(pending):
            raise RuntimeError('transfer log cannot be resolved')
        pending = rest
    return received


def merge_flat(parts, root_locations=None, received=None):
    """
    Graft the ranks' trees into one FlatTree.  Every part holds the replicated top of the
    tree plus its own subtrees; leaves flagged bit2 are placeholders for subtrees another
    rank owns.  Nodes are matched by location string.
    """
    owner = {}
    for pi, part in enumerate(parts):
        loc = part.locations(root_locations, received[pi] if received else None)
        for k, name in enumerate(loc):
            remote = bool(part.flags[k] & 4)
            if name not in owner or (owner[name][2] and not remote):
                owner[name] = (pi, k, remote)
    n_roots = parts[0].info['n_roots']
    first_loc = parts[0].locations(root_locations, received[0] if received else None)
    order = [first_loc[r] for r in range(n_roots)]
    index = {name: i for i, name in enumerate(order)}
    head = 0
    left, right = [], []
    while head < len(order):                      # breadth first from the roots
        name = order[head]
        pi, k, _ = owner[name]
        part = parts[pi]
        if part.left[k] >= 0:
            for ch in ('0', '1'):
                index[name + ch] = len(order)
                order.append(name + ch)
            left.append(index[name + '0'])
            right.append(index[name + '1'])
        else:
            left.append(-1)
            right.append(-1)
        head += 1
    src = [owner[name][:2] for name in order]

    def take(attr):
        return np.array([getattr(parts[pi], attr)[k] for pi, k in src])
    flags = take('flags')
    info = dict(parts[0].info)
    info['n_nodes'] = len(order)
    info['n_leaves'] = int(np.sum(np.array(left) < 0))
    info['n_closed'] = int(np.sum(flags & 1 > 0))
    info['lp_solves'] = sum(p.info['lp_solves'] for p in parts)
    info['volume_closed'] = None
    return FlatTree(take('vertices'), np.array(left, dtype=np.int32),
                    np.array(right, dtype=np.int32), take('delta_idx'), take('vertex_costs'),
                    take('vertex_inputs'), flags, take('tstar'), info, parts[0].deltas)


def claim_roots(n_roots, batch=1, key='ehm_root_ticket', store=None):
    """
    Dynamic deal of independent roots over the ranks (the reference's task queue,
    lib/scheduler.py:498-599, without a scheduler process): a shared counter in the
    ``torch.distributed`` store of the process group -- ``store.add`` is atomic -- hands out the
    next ``batch`` root indices to whoever asks.  Generator of lists of indices; exhausted when the
    counter has passed ``n_roots``.  No process group: this process takes every root, in order.
    ``key`` must be fresh per deal (the counter is never reset).
    """
    import torch.distributed as dist
    if store is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        store = dist.distributed_c10d._get_default_store()
    if store is None:
        for k0 in range(0, n_roots, batch):
            yield list(range(k0, min(n_roots, k0 + batch)))
        return
    while True:
        hi = int(store.add(key, int(batch)))
        lo = hi - int(batch)
        if lo >= n_roots:
            return
        yield list(range(lo, min(hi, n_roots)))


def emulate_claims(seconds, world):
    """Makespan of ``claim_roots`` over roots that take ``seconds[k]`` each (every rank asks for
    the next root the moment it is free): (per-rank busy seconds, max / mean)."""
    import heapq
    free = [(0.0, r) for r in range(world)]
    heapq.heapify(free)
    busy = [0.0] * world
    for t in seconds:
        at, r = heapq.heappop(free)
        busy[r] = at + float(t)
        heapq.heappush(free, (busy[r], r))
    mean = sum(busy) / max(world, 1)
    return busy, (max(busy) / mean if mean > 0 else 1.0)


class CellExchange:
    """
    Take / give of pending cells between the ranks, through the process group's store (the
    reference's scheduler hands any leaf to any idle worker, lib/scheduler.py:498-599, 633-639;
    here there is no scheduler process: a rank that ran out of roots ASKS the others in turn, a
    rank that still holds work answers between two slices of its run with the shallowest half of
    its pending cells -- ``NativeFrontier.take`` -- and everybody stops once all ranks are idle
    with nothing on its way).

    Keys under ``key`` (fresh per deal -- counters are never reset):
      idle             counter: ranks that are idle AND have no parcel on its way to them
      req/<v>          counter: requests addressed to rank v;  req/<v>/<t> = b"<requester>:<n>"
      box/<r>/<n>      the answer to rank r's n-th request: a pickled parcel, or None
      failed           counter: ranks whose driver raised (``fail``); a waiting rank raises too
    A giver lowers ``idle`` on the taker's behalf BEFORE it posts a non-empty parcel, so
    ``idle == world`` can only be seen when no work exists anywhere.
    """

    def __init__(self, store, rank, world, key='ehm_cells', poll=0.002):
        self.store, self.rank, self.world = store, int(rank), int(world)
        self.key, self.poll = key, float(poll)
        self.answered = 0           # requests addressed to this rank that it has answered
        self.asked = 0              # requests this rank has made
        self.given = 0              # parcels this rank gave away
        self._victim = self.rank

    def _k(self, *parts):
        return '/'.join([self.key] + [str(x) for x in parts])

    def serve(self, take):
        """Answers every request addressed to this rank: ``take()`` -> a parcel (dict of arrays,
        ``NativeFrontier.take``) or None.  Returns the parcels given away, each stamped with
        ``parcel['id']`` = (this rank, running number)."""
        import pickle
        out = []
        pending = int(self.store.add(self._k('req', self.rank), 0))
        while self.answered < pending:
            self.answered += 1
            who, n = self.store.get(self._k('req', self.rank, self.answered)).decode().split(':')
            parcel = take()
            if parcel is not None and len(parcel['node']):
                self.given += 1
                parcel['id'] = (self.rank, self.given)
                self.store.add(self._k('idle'), -1)         # it is on its way: `who` is busy again
                out.append(parcel)
            else:
                parcel = None
            self.store.set(self._k('box', who, n), pickle.dumps(parcel))
        return out

    def fail(self):
        """This rank gives up (its driver raised): the others must not wait for it."""
        self.store.add(self._k('failed'), 1)
        self.serve(lambda: None)

    def _ask(self):
        self._victim = (self._victim + 1) % self.world
        if self._victim == self.rank:
            self._victim = (self._victim + 1) % self.world
        self.asked += 1
        t = int(self.store.add(self._k('req', self._victim), 1))
        self.store.set(self._k('req', self._victim, t), '%d:%d' % (self.rank, self.asked))
        return self._k('box', self.rank, self.asked)

    def wait_for_work(self):
        """This rank is idle: asks the others in turn (answering their requests with None
        meanwhile) until one gives it a parcel -- returned -- or every rank is idle: None."""
        import pickle
        import time
        self.store.add(self._k('idle'), 1)
        box, empty = None, 0
        while True:
            self.serve(lambda: None)
            if box is None:
                box = self._ask()
            if self.store.check([box]):
                parcel = pickle.loads(self.store.get(box))
                box = None
                if parcel is not None:
                    return parcel
                empty += 1
                if empty % (self.world - 1):
                    continue            # the next rank may still hold work: ask at once
            if int(self.store.add(self._k('failed'), 0)) > 0:
                raise RuntimeError('another rank failed while this one was waiting for cells')
            if int(self.store.add(self._k('idle'), 0)) >= self.world:
                return None
            time.sleep(self.poll)


def grow_roots_sharded(oracle, roots, action='ecc', device=None, deal='static', native=None,
                       claim_key='ehm_root_ticket', batch=1, native_opts=None, steal=False,
                       steal_slice=2000, **kw):
    """
    The search-oracle driver (problems whose mode sequences cannot be enumerated) over the ranks;
    the subtrees of the roots -- the Delaunay roots of the set -- are independent, so the data path
    has no collective (weak scaling over the roots) and the counts are all-gathered at the end.

    deal='static':  root ``k`` belongs to rank ``k % world``; a rank's roots share their rounds.
    deal='dynamic': the ranks CLAIM roots from a shared counter (``claim_roots``), ``batch`` at a
                    time, each group grown to completion before the next claim -- the roots of
                    configs[4] cost anything between 2 s and 8 min each, and a static deal leaves
                    most ranks idle behind the slow ones.
    ``steal`` (with deal='dynamic' and ``native``): a rank that finds no root left asks the others
                    for pending CELLS (``CellExchange``: take / give between driver handles), so one
                    expensive root is shared; the sub-trees return to the rank that owns the root
                    (all-gather of the adopted trees at the end) and are attached there.  A busy
                    rank looks for requests every ``steal_slice`` cell visits.
    ``native``: a ``frontier.NativeFrontier`` -- the roots are then grown by the native driver
    (``frontier.grow_cells``, ``oracle`` only finishes cells it hands back open; ``native_opts``:
    its keyword arguments), otherwise by ``bnb_frontier.grow_frontier(oracle, ...)``.
    Returns (the list of roots with THIS rank's roots grown in place, stats of this rank with
    ``stats['mine']`` = the indices it grew, (world, 3) array of every rank's [host visits,
    leaves, roots]).  ``device``: where the final all-gather's tensors live ('cuda:k' under the
    nccl = RCCL backend, which cannot move host tensors; default: the oracle table's device there,
    the host under gloo).
    """
    import torch.distributed as dist
    from . import bnb_frontier
    rank, _, world = env_rank_world()
    if not (dist.is_available() and dist.is_initialized()):
        world, rank = 1, 0

    exchange = None
    if steal and deal == 'dynamic' and native is not None and world > 1:
        exchange = CellExchange(dist.distributed_c10d._get_default_store(), rank, world,
                                key=claim_key + '/cells')
    given = []              # [(parcel id, [leaf per cell])]: cells of this rank's runs grown elsewhere

    def serve(nat, st):     # between two slices of a run: requests of idle ranks
        return exchange.serve(lambda: nat.take(nat.pending() // 2) if nat.pending() >= 2 else None)

    def grow(part, cells=None):
        if native is not None:
            from . import frontier
            opts = dict(native_opts or {})
            if exchange is not None:
                opts.update(between_slices=serve, slice_visits=int(steal_slice))
            st = frontier.grow_cells(native, part, slow_oracle=lambda: oracle,
                                     slow_opts=kw or None, cells=cells, **opts)
            out = dict(host_visits=st['visits'] + st['slow_path_visits'], rounds=st['rounds'],
                       truncated=bool(st['truncated']), regions=st['regions'])
            out['given'] = [(pc['id'], leaves) for pc, leaves in st['given_away']]
            return out
        return bnb_frontier.grow_frontier(oracle, part, action, **kw)

    def tally(st):
        stats['host_visits'] += st['host_visits']
        stats['rounds'] += st['rounds']
        stats['truncated'] = stats['truncated'] or bool(st['truncated'])
        stats['regions'] += int(st.get('regions', 0))

    stats = dict(host_visits=0, rounds=0, truncated=False, regions=0)
    if deal == 'dynamic':
        mine = []
        adopted = []
        try:
            for ks in claim_roots(len(roots), batch=batch, key=claim_key):
                st = grow([roots[k] for k in ks])
                mine += ks
                tally(st)
                given += st.get('given', [])
            while exchange is not None:
                from .tree import NodeData, Tree
                parcel = exchange.wait_for_work()
                if parcel is None:
                    break
                trees = [Tree(NodeData(vertices=V.copy())) for V in parcel['vertices']]
                st = grow(trees, cells=parcel)
                tally(st)
                adopted.append(dict(id=parcel['id'], trees=trees, given=st['given']))
        except BaseException:
            # the ranks that wait for cells (now or later) leave with an error of their own
            # instead of waiting for this rank to become idle
            if exchange is not None:
                exchange.fail()
            raise
        if exchange is not None:
            stats['cells_adopted'] = sum(len(a['trees']) for a in adopted)
            stats['parcels_given'] = exchange.given
            everybody = [None] * world
            dist.all_gather_object(everybody, adopted)
            by_id = {a['id']: a for part in everybody for a in part}
            from . import frontier
            stats['subtrees_attached'] = frontier.attach_adopted(given, by_id)
    else:
        mine = [k for k in range(len(roots)) if k % world == rank]
        if mine:
            stats = dict(grow([roots[k] for k in mine]))
    stats['mine'] = mine
    leaves = sum(1 for k in mine for _ in roots[k].leaves())
    if device is None and world > 1 and dist.get_backend() == 'nccl':
        device = 'cuda:%d' % int(getattr(getattr(oracle, 'table', None), 'device', 0))
    counts = allgather_counts([stats['host_visits'], leaves, len(mine)], device=device)
    return roots, stats, counts
