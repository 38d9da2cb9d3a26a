/*
 * TEST INFRASTRUCTURE (see oracle/__init__.py) -- not linked into the product library.
 *
 * Restates the floating-point arithmetic of tools.split_along_longest_edge
 * (reference lib/tools.py:224-257).  There the edge length is la.norm(R[i]-R[j]), which
 * numpy evaluates as sqrt(x.dot(x)); for float64 vectors that dot product is OpenBLAS
 * ddot, whose scalar tail loop compiles to a fused multiply-add chain
 *     s = fma(x[k], x[k], s),  k = 0 .. p-1,  s0 = 0
 * (verified bit-for-bit against numpy 2.2.6 / OpenBLAS 0.3.29 for p <= 9 by
 * tests/golden/make_geometry_golden.py).  np.argmax keeps the FIRST maximal edge in
 * itertools.combinations order.  Compiled with -ffp-contract=off so that only the fma()
 * calls written here fuse.
 */
#include <math.h>

double ehm_ref_edge_length(const double* a, const double* b, int p)
{
    double s = 0.0;
    for (int k = 0; k < p; ++k) {
        double d = a[k] - b[k];
        s = fma(d, d, s);
    }
    return sqrt(s);
}

/* R: nv rows of p doubles, row-major.  Returns the edge as (i,j), i<j. */
void ehm_ref_longest_edge(const double* R, int nv, int p, int* out_i, int* out_j)
{
    double best = -1.0;
    int bi = 0, bj = 1;
    for (int i = 0; i < nv; ++i)
        for (int j = i + 1; j < nv; ++j) {
            double len = ehm_ref_edge_length(R + (long)i * p, R + (long)j * p, p);
            if (len > best) { best = len; bi = i; bj = j; }
        }
    *out_i = bi;
    *out_j = bj;
}
