"""
Geometry helpers of the partition loop with the reference's signatures
(lib/tools.py:134-257).  The arithmetic that decides the shape of the tree
(``split_along_longest_edge``) and the volume accounting run on the GPU through the
C-ABI; the one-off Delaunay triangulation of the set to partition stays on Qhull
(``scipy.spatial``), exactly as in the reference (lib/tools.py:173).
"""

import numpy as np
import scipy.spatial

from . import engine
from .tree import Tree, NodeData

DEVICE = 0


def split_along_longest_edge(R):
    """
    Split simplex R (rows = vertices) at the midpoint of its longest edge.
    Returns (S_1, S_2, (i, j)) like lib/tools.py:224-257: row i of S_1 and row j of S_2
    are the midpoint; the first longest edge in ``itertools.combinations`` order is taken.
    """
    S1, S2, ij = engine.split_batch(np.asarray(R, dtype=np.float64)[None], device=DEVICE)
    return S1[0], S2[0], (int(ij[0, 0]), int(ij[0, 1]))


def simplex_volume(R):
    """|det([v_i - v_0])| / p!  (lib/tools.py:134-150)."""
    return float(engine.volume_batch(np.asarray(R, dtype=np.float64)[None], device=DEVICE)[0])


def delaunay_roots(V):
    """
    Root simplices of the polytope with vertices V (rows), in Qhull order, and their
    location strings on the reference's right-spine tree ('1'*i+'0', last '1'*(n-1);
    lib/tools.py:176-188).  Returns (roots (n, p+1, p), locations).
    """
    V = np.asarray(V, dtype=np.float64)
    tri = scipy.spatial.Delaunay(V)
    roots = np.ascontiguousarray(V[tri.simplices])
    n = roots.shape[0]
    if n == 1:
        return roots, ['']
    return roots, ['1' * i + '0' for i in range(n - 1)] + ['1' * (n - 1)]


def delaunay(R):
    """
    Partition polytope R (rows = vertices) into simplices arranged as the reference's
    right-spine binary tree (lib/tools.py:152-189).  Returns (root, Nsx, vol).
    """
    R = np.asarray(R, dtype=np.float64)
    roots, _ = delaunay_roots(R)
    Nsx = roots.shape[0]
    vol = float(np.sum(engine.volume_batch(roots, device=DEVICE)))
    root = Tree(NodeData(vertices=R))
    cursor = root
    for i in range(Nsx - 1):
        right = NodeData(vertices=roots[i + 1]) if i == Nsx - 2 else None
        cursor.grow(NodeData(vertices=roots[i]), right)
        cursor = cursor.right
    return root, Nsx, vol


def join_triangulation(cursor, new_tree):
    """
    Append ``new_tree`` at the rightmost leaf of ``cursor`` (lib/tools.py:191-222):
    that leaf becomes a data-less parent of (its old payload, new_tree).
    """
    while not cursor.is_leaf():
        cursor = cursor.right
    cursor.grow(cursor.data, None)
    cursor.right = new_tree
    new_tree.top = False
    cursor.data = None
