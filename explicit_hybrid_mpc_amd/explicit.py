"""
Explicit MPC evaluation with the reference's interface (lib/mpc_library.py:662-792) on the GPU:

    ExplicitMPC(tree, oracle)(x)  ->  (u, t)            one state, like the reference
    ExplicitMPC(tree, oracle).evaluate(X)  ->  u [n, n_u]   a batch in one kernel launch

``tree`` is the partition as the reference stores it (nested ``tree.Tree`` with the Delaunay
right spine on top, e.g. loaded from a reference ``tree.pkl``) or the engine's ``FlatTree``.
Point location follows the reference exactly: at an internal node go left iff the state lies
in the left child's simplex (barycentric weights in [-eps, 1+eps]), else right; at the leaf
interpolate the vertex inputs.  The walk runs in ``ehm_explicit_eval_batch`` (one thread per
state); ``inv([v1-v0 .. vp-v0])`` of every simplex is computed on the device at set-up
(``compute_simplex_basis_inverse``, lib/mpc_library.py:685-712).  No CPU fallback.
"""

import ctypes
import time

import numpy as np

from . import _capi
from ._capi import f64, ptr
from .engine import FlatTree


def _check(rc):
    if rc != _capi.EHM_OK:
        raise _capi.EhmError(rc, _capi.load().ehm_explicit_last_error().decode('utf-8', 'replace'))


def flatten_tree(root):
    """
    Nested ``tree.Tree`` (reference layout) -> flat arrays with node 0 = root, children after
    their parents: (vertices, vertex_inputs, left, right, nodes).  Data-less spine nodes and
    nodes without inputs get placeholders; they are never tested (only left children and final
    leaves are, lib/mpc_library.py:703-711) or never returned.
    """
    first, stack = None, [root]
    while stack:                        # any node with vertex inputs tells the shapes
        nd = stack.pop()
        if nd.data is not None and hasattr(nd.data, 'vertex_inputs'):
            first = nd
            break
        if not nd.is_leaf():
            stack += [nd.right, nd.left]
    if first is None:
        raise ValueError('the tree has no node with vertex inputs')
    p = np.asarray(first.data.vertices).shape[1]
    n_u = np.asarray(first.data.vertex_inputs).shape[1]
    unit = np.vstack([np.zeros(p), np.eye(p)])
    nodes, left, right = [root], [], []
    head = 0
    while head < len(nodes):
        nd = nodes[head]
        if nd.is_leaf():
            left.append(-1)
            right.append(-1)
        else:
            left.append(len(nodes))
            right.append(len(nodes) + 1)
            nodes += [nd.left, nd.right]
        head += 1
    K = len(nodes)
    vertices = np.empty((K, p + 1, p))
    vinput = np.zeros((K, p + 1, n_u))
    for k, nd in enumerate(nodes):
        d = nd.data
        ok = d is not None and np.asarray(d.vertices).shape == (p + 1, p)
        vertices[k] = np.asarray(d.vertices, dtype=np.float64) if ok else unit
        if d is not None and hasattr(d, 'vertex_inputs'):
            vinput[k] = np.asarray(d.vertex_inputs, dtype=np.float64)
    return vertices, vinput, np.array(left, dtype=np.int32), np.array(right, dtype=np.int32), nodes


class ImplicitMPC:
    """
    The implicit law of lib/mpc_library.py:626-660 (same constructor and call signature): one
    mixed-integer oracle solve ``P_theta(x)`` per call, i.e. on the device one batched launch
    over the commutations.  ``evaluate`` is the batched form.
    """

    def __init__(self, oracle):
        mpc = oracle.mpc
        self.plant = getattr(mpc, 'plant', None)
        self.T_s = getattr(mpc, 'T_s', None)
        if hasattr(mpc, 'specs'):
            self.specs = mpc.specs
        self.__oracle = oracle

    def __call__(self, x):
        """(u, t): epsilon-suboptimal (here: optimal) input and evaluation time."""
        u, _, _, t = self.__oracle.P_theta(x)
        return u, t

    def evaluate(self, X):
        """Inputs for a batch of states (n, p) -> (n, n_u); NaN rows where infeasible."""
        J, u0, didx = self.__oracle.gpu.solve_pt(np.asarray(X, dtype=np.float64))
        u0 = u0.copy()
        u0[didx < 0] = np.nan
        return u0


class ExplicitMPC:
    """GPU counterpart of lib/mpc_library.py:662-792 (same constructor and call signature)."""

    def __init__(self, tree, oracle=None, device=0):
        mpc = getattr(oracle, 'mpc', None)
        self.plant = getattr(mpc, 'plant', None)
        self.T_s = getattr(mpc, 'T_s', None)
        if hasattr(mpc, 'specs'):
            self.specs = mpc.specs
        self.tree = tree
        self.device = int(device)
        self._lib = _capi.load()
        self._handle = ctypes.c_void_p()
        self.setup()

    def setup(self):
        """Readies the evaluator: flat arrays + simplex basis inverses on the device."""
        if isinstance(self.tree, FlatTree):
            t = self.tree
            vertices, vinput = f64(t.vertices), f64(t.vertex_inputs)
            left = np.ascontiguousarray(t.left, dtype=np.int32)
            right = np.ascontiguousarray(t.right, dtype=np.int32)
            n_roots = int(t.info['n_roots'])
            self.nodes = None
        else:
            vertices, vinput, left, right, self.nodes = flatten_tree(self.tree)
            n_roots = 1
        self.n_nodes, self.p, self.n_u = vertices.shape[0], vertices.shape[2], vinput.shape[2]
        self.eps = np.finfo(np.float64).eps
        _check(self._lib.ehm_explicit_create(self.device, self.n_nodes, n_roots, self.p, self.n_u,
                                             ptr(left), ptr(right), ptr(vertices), ptr(vinput),
                                             ctypes.byref(self._handle)))

    def close(self):
        if getattr(self, '_handle', None):
            self._lib.ehm_explicit_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def evaluate(self, X, return_info=False):
        """u [n, n_u] for the states X [n, p]; with return_info also (leaf ids, tests, seconds)."""
        X = f64(np.atleast_2d(X))
        n = X.shape[0]
        u = np.empty((n, self.n_u))
        leaf = np.empty(n, dtype=np.int32)
        visited = np.empty(n, dtype=np.int32)
        secs = ctypes.c_double(0.)
        _check(self._lib.ehm_explicit_eval_batch(self._handle, n, ptr(X), ptr(u), ptr(leaf),
                                                 ptr(visited), ctypes.addressof(secs)))
        if return_info:
            return u, leaf, visited, secs.value
        return u

    def get_containing_cell(self, x):
        """NodeData of the leaf that contains x (nested trees), or its node index (FlatTree)."""
        _, leaf, _, _ = self.evaluate(np.asarray(x, dtype=np.float64)[None], return_info=True)
        k = int(leaf[0])
        return self.nodes[k].data if self.nodes is not None else k

    def __call__(self, x):
        """Epsilon-suboptimal input for state x and the evaluation time (lib/mpc_library.py:769)."""
        tic = time.time()
        u = self.evaluate(np.asarray(x, dtype=np.float64)[None])[0]
        return u, time.time() - tic
