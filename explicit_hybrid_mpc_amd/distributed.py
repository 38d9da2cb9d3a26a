"""
Multi-GPU layer: one process per GPU under ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on the GPU node, "gloo" in the CPU tests).

The reference distributes work with an MPI master/worker star and pickled subtrees
(lib/scheduler.py:498-599, lib/worker.py:187-239).  Here the live frontier is sharded instead:
every rank grows the same top of the tree, and once a sweep's frontier is wide enough each
rank keeps the positions ``k % world == rank`` (``ehm_run_opts.shard_*``).  Subtrees are
independent, so the data path needs no collective; the collectives below carry only counters
(work accounting, load-imbalance report) and, on request, the ranks' finished subtrees to
rank 0, where they are grafted by location -- the contract of ``build_tree``
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


def merge_flat(parts, root_locations=None):
    """
    Graft the ranks' trees into one FlatTree.  Every part holds the replicated top of the
    tree plus its own subtrees; leaves flagged bit2 are placeholders for subtrees another
    rank owns.  Nodes are matched by location string.
    """
    owner = {}
    for pi, part in enumerate(parts):
        loc = part.locations(root_locations)
        for k, name in enumerate(loc):
            remote = bool(part.flags[k] & 4)
            if name not in owner or (owner[name][2] and not remote):
                owner[name] = (pi, k, remote)
    n_roots = parts[0].info['n_roots']
    first_loc = parts[0].locations(root_locations)
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
