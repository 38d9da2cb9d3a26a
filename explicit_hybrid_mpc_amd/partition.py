"""
Partition algorithms with the reference's entry points (lib/worker.py:180-185, 241-417):

    alg_call(which_alg, branch, location)   ->  grows ``branch`` (a ``tree.Tree``) in place

``ecc`` builds the feasible-commutation partition and continues with ``lcss``;
``lcss`` refines until every leaf is epsilon-suboptimal.  The recursion, MPI off-loading
and per-node Python oracle calls of the reference are replaced by frontier sweeps on the
GPU inside ``ehm_partition_run``: single-commutation problems in the persistent frontier
kernel (or level-synchronous sweeps), multi-commutation (hybrid) problems in the device
engine of csrc/ehm_hybrid.h.

The per-node semantics are the reference's; the grown tree does
not depend on the visiting order because a node's fate depends only on its own record.
"""

import numpy as np

from . import engine
from .engine import FlatTree
from .tree import Tree, NodeData


def grow_hybrid(gp, roots, action='ecc', init=None, max_nodes=1 << 20, **kw):
    """
    ``ecc`` / ``lcss`` for any number of commutations: one call into ``ehm_partition_run`` --
    the device engine of csrc/ehm_hybrid.h (frontier sweeps whose work lists, decisions and
    child records never leave the GPU).  Returns a FlatTree.
    """
    return gp.partition(roots, action=action, init=init, max_nodes=max_nodes, **kw)


# ---------------------------------------------------------------------------------------
# FlatTree <-> reference tree objects
# ---------------------------------------------------------------------------------------
def node_data_from_flat(flat, k):
    """NodeData of flat node k, attribute-absence semantics of lib/tree.py:34-39."""
    has = bool(flat.flags[k] & 2)
    nd = NodeData(vertices=flat.vertices[k].copy(),
                  commutation=flat.deltas[flat.delta_idx[k]].copy() if has else None,
                  vertex_costs=flat.vertex_costs[k].copy() if has else None,
                  vertex_inputs=flat.vertex_inputs[k].copy() if has else None)
    nd.is_epsilon_suboptimal = bool(flat.flags[k] & 1)
    return nd


def graft_flat(flat, targets):
    """
    Write the subtree below flat root r into ``targets[r]`` (existing ``Tree`` nodes, grown
    in place), iteratively.  One bulk copy of every array (the export's arrays are page-locked
    and recycled: the tree must not hold on to them), then the nodes are ROW VIEWS of those copies
    and their payloads are filled through ``__dict__`` with the cyclic collector paused.  What is
    left is the interpreter allocating ~8 objects per node (about 4 microseconds: 1.6 M nodes take
    seconds either way; the flat arrays are what ``explicit.ExplicitMPC`` and ``tree_io`` read, the
    object tree is for consumers written against lib/tree.py).
    """
    import gc
    import time
    n = flat.n_nodes
    # (millions of small objects: the cyclic collector would rescan the growing tree again and
    # again -- it is paused for the loop, there is nothing cyclic to find in it)
    was_on = gc.isenabled()
    gc.disable()
    try:
        return _graft_flat(flat, targets, n, time.time())
    finally:
        if was_on:
            gc.enable()


def _graft_flat(flat, targets, n, stamp):
    V = list(np.array(flat.vertices))          # list of (p+1, p) views into ONE private copy
    C = list(np.array(flat.vertex_costs))
    U = list(np.array(flat.vertex_inputs))
    left, right = flat.left.tolist(), flat.right.tolist()
    flags = flat.flags.tolist()
    didx = flat.delta_idx.tolist()
    # one array per distinct commutation, shared by the nodes that hold it -- as the reference's
    # children share their parent's delta_star object (lib/worker.py:356-365)
    deltas = [np.array(d) for d in flat.deltas]
    new_data, new_tree = NodeData.__new__, Tree.__new__
    nodes = [None] * n
    for r, t in enumerate(targets):
        nodes[r] = t
    for k in range(n):                          # parents precede children
        node = nodes[k]
        f = flags[k]
        data = new_data(NodeData)
        if f & 2:                               # attribute-absence semantics of lib/tree.py:34-39
            data.__dict__ = {'timestamp': stamp, 'vertices': V[k],
                             'is_epsilon_suboptimal': bool(f & 1),
                             'commutation': deltas[didx[k]], 'vertex_costs': C[k],
                             'vertex_inputs': U[k]}
        else:
            data.__dict__ = {'timestamp': stamp, 'vertices': V[k],
                             'is_epsilon_suboptimal': bool(f & 1)}
        node.data = data
        lk = left[k]
        if lk >= 0:
            a, b = new_tree(Tree), new_tree(Tree)
            a.__dict__ = {'data': None, 'top': False}
            b.__dict__ = {'data': None, 'top': False}
            node.left, node.right = a, b
            nodes[lk] = a
            nodes[right[k]] = b
    return targets


def run_engine(oracle, roots, action='ecc', init=None, **kw):
    """Grow ``roots`` on the GPU and return the FlatTree."""
    return oracle.gpu.partition(roots, action=action, init=init, **kw)


def alg_call(oracle, which_alg, branch, location=''):
    """
    Drop-in for the reference's ``alg_call(which_alg, branch, location)``
    (lib/worker.py:180-185): grows ``branch`` in place, returns None.  ``location`` is
    accepted for signature compatibility (the result does not depend on it).
    """
    data = branch.data
    R = np.asarray(data.vertices, dtype=np.float64)[None]
    init = None
    if which_alg != 'ecc':
        init = dict(delta=np.asarray(data.commutation)[None],
                    vertex_costs=np.asarray(data.vertex_costs)[None],
                    vertex_inputs=np.asarray(data.vertex_inputs)[None])
    flat = run_engine(oracle, R, action='ecc' if which_alg == 'ecc' else 'lcss', init=init)
    graft_flat(flat, [branch])
    return None


class Partitioner:
    """Object form with the reference's method names (``Worker.ecc`` / ``Worker.lcss``)."""

    def __init__(self, oracle):
        self.oracle = oracle

    def ecc(self, node, location=''):
        return alg_call(self.oracle, 'ecc', node, location)

    def lcss(self, node, location=''):
        return alg_call(self.oracle, 'lcss', node, location)


def partition_set(oracle, set_vertices, **kw):
    """
    Whole pipeline of scheduler.setup + workers (lib/scheduler.py:373-391, 620-642): Delaunay
    roots of the set, every root grown to epsilon-suboptimality in ONE engine run, result
    grafted into the reference's right-spine tree.  Returns (root Tree, FlatTree).
    """
    from . import tools
    root, Nsx, vol = tools.delaunay(set_vertices)
    roots, locs = tools.delaunay_roots(set_vertices)
    flat = run_engine(oracle, roots, action='ecc', **kw)
    targets = [root.descend(loc) for loc in locs]
    graft_flat(flat, targets)
    return root, flat
