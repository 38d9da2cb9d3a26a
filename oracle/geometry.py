"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

CPU restatement of the geometry helpers of the partition loop
(reference lib/tools.py:134-257).  PINNED against fixtures produced by the unmodified
reference functions (tests/golden/geometry_golden.npz, made by
tests/golden/make_geometry_golden.py).
"""

import ctypes
import itertools
import math
import numpy as np
import scipy.spatial

from . import build as _build

_lib = None


def _geom():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        _lib.ehm_ref_longest_edge.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_int),
                                              ctypes.POINTER(ctypes.c_int)]
        _lib.ehm_ref_longest_edge.restype = None
        _lib.ehm_ref_edge_length.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib.ehm_ref_edge_length.restype = ctypes.c_double
    return _lib


def longest_edge(R):
    """First longest edge (i, j), i < j -- lib/tools.py:244-249."""
    R = np.ascontiguousarray(R, dtype=np.float64)
    i, j = ctypes.c_int(), ctypes.c_int()
    _geom().ehm_ref_longest_edge(R.ctypes.data, R.shape[0], R.shape[1],
                                 ctypes.byref(i), ctypes.byref(j))
    return i.value, j.value


def longest_edge_numpy(R):
    """Same rule through numpy's own norm/argmax (host BLAS decides the rounding)."""
    combos = list(itertools.combinations(range(R.shape[0]), 2))
    k = int(np.argmax([np.linalg.norm(R[a] - R[b]) for a, b in combos]))
    return combos[k]


def split_along_longest_edge(R):
    """lib/tools.py:224-257: returns (S_1, S_2, (i, j))."""
    R = np.asarray(R, dtype=np.float64)
    i, j = longest_edge(R)
    v_mid = (R[i] + R[j]) / 2.
    S_1, S_2 = R.copy(), R.copy()
    S_1[i] = v_mid
    S_2[j] = v_mid
    return S_1, S_2, (i, j)


def simplex_volume(R):
    """lib/tools.py:134-150: |det([v_i - v_0])| / p!."""
    R = np.asarray(R, dtype=np.float64)
    M = np.column_stack([v - R[0] for v in R[1:]])
    return abs(np.linalg.det(M)) / math.factorial(R.shape[0] - 1)


def delaunay_simplices(V):
    """
    Root simplices of the set with vertices V (rows), in the order the reference's
    ``delaunay`` walks them (lib/tools.py:171-188: Qhull order).  Returns a list of
    ((p+1, p) arrays) and the right-spine location strings '1'*i+'0' (last: '1'*(n-1)).
    """
    V = np.asarray(V, dtype=np.float64)
    tri = scipy.spatial.Delaunay(V)
    simplices = [V[idx].copy() for idx in tri.simplices]
    n = len(simplices)
    if n == 1:
        return simplices, ['']
    locations = ['1' * i + '0' for i in range(n - 1)] + ['1' * (n - 1)]
    return simplices, locations
