"""
Generates tests/golden/geometry_golden.npz by running the UNMODIFIED reference functions
``tools.split_along_longest_edge``, ``tools.simplex_volume``, ``tools.delaunay``
(/root/reference/lib/tools.py:134-257) and ``tree.Tree`` / ``tree.NodeData``
(/root/reference/lib/tree.py).  Runs only in the build container (the reference tree is
not present on the GPU box); the .npz it writes is committed.

``tools.py`` imports mpi4py at module scope (lib/tools.py:19-22) only for its MPI
wrapper, so a stub module is injected; nothing of the stub is exercised by the geometry
functions.
"""

import os
import sys
import types
import numpy as np

sys.dont_write_bytecode = True
REF = '/root/reference/lib'


def import_reference_tools():
    stub = types.ModuleType('mpi4py')
    stub.rc = types.SimpleNamespace()
    mpi = types.ModuleType('mpi4py.MPI')

    class _Comm:
        def Get_rank(self):
            return 0

        def Get_size(self):
            return 1
    mpi.COMM_WORLD = _Comm()
    stub.MPI = mpi
    sys.modules['mpi4py'] = stub
    sys.modules['mpi4py.MPI'] = mpi
    sys.path.insert(0, REF)
    import tools  # noqa: E402  (the reference's)
    import tree   # noqa: E402
    return tools, tree


def main():
    tools, tree = import_reference_tools()
    rng = np.random.default_rng(20260921)
    out = {}
    # 1) random simplices
    for p in (2, 3, 4, 6, 8):
        n = 150
        R = rng.standard_normal((n, p + 1, p))
        S1 = np.empty_like(R)
        S2 = np.empty_like(R)
        ij = np.empty((n, 2), dtype=np.int32)
        vol = np.empty(n)
        for k in range(n):
            a, b, c = tools.split_along_longest_edge(R[k])
            S1[k], S2[k], ij[k] = a, b, c
            vol[k] = tools.simplex_volume(R[k])
        out['rand_p%d_R' % p] = R
        out['rand_p%d_S1' % p] = S1
        out['rand_p%d_S2' % p] = S2
        out['rand_p%d_ij' % p] = ij
        out['rand_p%d_vol' % p] = vol
    # 2) tie-heavy: Delaunay of boxes with non-dyadic half-widths, then repeated
    #    longest-edge bisection (breadth first) -- exact and near ties everywhere
    for p in (2, 3, 4):
        half = 0.3 + 0.4 * rng.random(p)
        signs = np.array(np.meshgrid(*[[-1., 1.]] * p, indexing='ij')).reshape(p, -1).T
        V = signs * half
        root, Nsx, volume = tools.delaunay(V)
        # collect leaves of the right spine in order
        leaves = []
        cursor = root
        while not cursor.is_leaf():
            leaves.append(cursor.left.data.vertices)
            if cursor.right.data is not None and cursor.right.is_leaf():
                leaves.append(cursor.right.data.vertices)
                break
            cursor = cursor.right
        if Nsx == 1:
            leaves = [root.data.vertices]
        out['box_p%d_V' % p] = V
        out['box_p%d_roots' % p] = np.array(leaves)
        out['box_p%d_Nsx' % p] = np.array(Nsx)
        out['box_p%d_vol' % p] = np.array(volume)
        frontier = list(leaves)
        Rs, S1s, S2s, ijs, vols = [], [], [], [], []
        depth = {2: 9, 3: 7, 4: 5}[p]
        for _ in range(depth):
            nxt = []
            for R in frontier:
                a, b, c = tools.split_along_longest_edge(R)
                Rs.append(R)
                S1s.append(a)
                S2s.append(b)
                ijs.append(c)
                vols.append(tools.simplex_volume(R))
                nxt += [a, b]
            frontier = nxt
        out['tie_p%d_R' % p] = np.array(Rs)
        out['tie_p%d_S1' % p] = np.array(S1s)
        out['tie_p%d_S2' % p] = np.array(S2s)
        out['tie_p%d_ij' % p] = np.array(ijs, dtype=np.int32)
        out['tie_p%d_vol' % p] = np.array(vols)
    # 3) tree semantics (lib/tree.py:31-39, 72-95)
    nd = tree.NodeData(np.zeros((3, 2)))
    t = tree.Tree(nd)
    flags = [hasattr(nd, 'commutation'), hasattr(nd, 'vertex_costs'),
             hasattr(nd, 'vertex_inputs'), nd.is_epsilon_suboptimal, t.top, t.is_leaf()]
    t.grow(tree.NodeData(np.ones((3, 2))), None)
    flags += [t.is_leaf(), t.left.top, t.right.data is None, t.left.is_leaf()]
    out['tree_flags'] = np.array(flags, dtype=np.int32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'geometry_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items() if k.endswith('_R')})


if __name__ == '__main__':
    main()
