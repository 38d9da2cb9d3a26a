"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

CPU restatement of the reference's explicit-MPC evaluation, function by function:
``compute_simplex_basis_inverse`` (lib/mpc_library.py:685-712), ``check_containment``
(:714-735), ``get_containing_cell`` (:737-767) and ``__call__`` (:769-792), on the nested
``Tree`` the reference stores (recursion replaced by loops: the spine of a p = 6 set is 652
levels deep).  The reference class itself cannot be imported here (lib/mpc_library.py imports
cvxpy at module level), and the reference ships no fixtures for it: parity unpinned, but the
arithmetic is 20 lines of numpy restated verbatim (la.inv, Minv.dot(x - c), the eps test).
"""

import numpy as np
import numpy.linalg as la


class ExplicitCPU:
    def __init__(self, tree):
        self.tree = tree
        self.eps = np.finfo(np.float64).eps          # lib/mpc_library.py:683
        self.compute_simplex_basis_inverse()

    @staticmethod
    def _minv(v):
        return la.inv(np.column_stack([_v - v[0] for _v in v[1:]]))     # :705

    def compute_simplex_basis_inverse(self):
        stack = [self.tree]
        while stack:
            cursor = stack.pop()
            if cursor.is_leaf():                                         # :706-707
                cursor.data.Minv = self._minv(cursor.data.vertices)
            else:                                                        # :708-711
                cursor.left.data.Minv = self._minv(cursor.left.data.vertices)
                stack += [cursor.right, cursor.left]

    def check_containment(self, x, cell):
        c = cell.vertices[0]
        alpha = list(cell.Minv.dot(x - c))                               # :731
        alpha.append(1 - sum(alpha))
        return bool(np.all([a >= -self.eps and a <= 1 + self.eps for a in alpha]))

    def get_containing_cell(self, x):
        cursor = self.tree
        while not cursor.is_leaf():                                      # :760-766
            cursor = cursor.left if self.check_containment(x, cursor.left.data) else cursor.right
        return cursor.data

    def __call__(self, x):
        R = self.get_containing_cell(x)
        alpha = R.Minv.dot(x - R.vertices[0])                            # :786
        alpha0 = 1 - sum(alpha)
        return alpha0 * R.vertex_inputs[0] + R.vertex_inputs[1:].T.dot(alpha)   # :788


class ExplicitFlatCPU:
    """
    The same walk (lib/mpc_library.py:714-792) over the FLAT arrays the partition engine exports
    (nodes 0 .. n_roots-1 = the Delaunay roots in spine order, children behind their parents):
    ``la.inv`` of a simplex the first time the walk tests it, ``Minv.dot(x - c)``, the eps test.
    Used as the checker of the device evaluation on flat trees and as ``bench.py --workload
    explicit``'s cpu_baseline.
    """

    def __init__(self, vertices, vertex_inputs, left, right, n_roots):
        self.V = np.asarray(vertices, dtype=np.float64)
        self.U = np.asarray(vertex_inputs, dtype=np.float64)
        self.left = np.asarray(left)
        self.right = np.asarray(right)
        self.n_roots = int(n_roots)
        self.eps = np.finfo(np.float64).eps
        self._minv = {}

    def minv(self, k):
        M = self._minv.get(k)
        if M is None:
            v = self.V[k]
            M = self._minv[k] = la.inv(np.column_stack([_v - v[0] for _v in v[1:]]))
        return M

    def check_containment(self, x, k):
        alpha = list(self.minv(k).dot(x - self.V[k][0]))
        alpha.append(1 - sum(alpha))
        return bool(np.all([a >= -self.eps and a <= 1 + self.eps for a in alpha]))

    def get_containing_cell(self, x):
        k = self.n_roots - 1                     # the spine: first root that holds x, else the last
        for r in range(self.n_roots - 1):
            if self.check_containment(x, r):
                k = r
                break
        while self.left[k] >= 0:
            k = self.left[k] if self.check_containment(x, self.left[k]) else self.right[k]
        return int(k)

    def __call__(self, x):
        k = self.get_containing_cell(x)
        alpha = self.minv(k).dot(x - self.V[k][0])
        alpha0 = 1 - sum(alpha)
        return alpha0 * self.U[k][0] + self.U[k][1:].T.dot(alpha), k
