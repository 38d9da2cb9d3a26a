"""
The oracle's geometry restatement against fixtures produced by the UNMODIFIED reference
functions (tests/golden/make_geometry_golden.py -> lib/tools.py:134-257, lib/tree.py).
This is what pins the geometry half of the oracle.
"""

import numpy as np

from oracle import geometry


def _cases(z):
    return [k[:-2] for k in z.files if k.endswith('_R')]


def test_split_bit_exact_against_reference(golden_geometry):
    z = golden_geometry
    n = 0
    for pre in _cases(z):
        R, S1, S2, ij = z[pre + '_R'], z[pre + '_S1'], z[pre + '_S2'], z[pre + '_ij']
        for q in range(R.shape[0]):
            a, b, c = geometry.split_along_longest_edge(R[q])
            assert c == tuple(ij[q]), (pre, q)
            assert np.array_equal(a, S1[q]) and np.array_equal(b, S2[q]), (pre, q)
            n += 1
    assert n > 3000


def test_tie_cases_really_contain_ties(golden_geometry):
    """The box-derived fixtures must exercise the first-max rule (exact ties present)."""
    z = golden_geometry
    ties = 0
    for p in (2, 3, 4):
        R = z['tie_p%d_R' % p]
        for q in range(R.shape[0]):
            lens = sorted((np.linalg.norm(R[q][a] - R[q][b])
                           for a in range(p + 1) for b in range(a + 1, p + 1)), reverse=True)
            ties += lens[0] == lens[1]
    assert ties > 100


def test_host_blas_agrees_with_fma_chain(golden_geometry):
    """numpy's own norm/argmax on this host picks the same edges as the C restatement."""
    z = golden_geometry
    R = z['tie_p4_R']
    for q in range(0, R.shape[0], 7):
        assert geometry.longest_edge_numpy(R[q]) == geometry.longest_edge(R[q])


def test_volume_and_delaunay(golden_geometry):
    z = golden_geometry
    for pre in _cases(z):
        R, vol = z[pre + '_R'], z[pre + '_vol']
        for q in range(0, R.shape[0], 5):
            assert abs(geometry.simplex_volume(R[q]) - vol[q]) <= 1e-13 * abs(vol[q])
    for p in (2, 3, 4):
        V = z['box_p%d_V' % p]
        roots, locs = geometry.delaunay_simplices(V)
        assert len(roots) == int(z['box_p%d_Nsx' % p])
        assert np.array_equal(np.array(roots), z['box_p%d_roots' % p])
        total = sum(geometry.simplex_volume(r) for r in roots)
        assert abs(total - float(z['box_p%d_vol' % p])) <= 1e-12 * total
        assert locs[0] == '0' and locs[-1] == '1' * (len(roots) - 1)
