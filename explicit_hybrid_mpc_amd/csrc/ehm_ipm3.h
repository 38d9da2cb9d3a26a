// Workgroup-cooperative primal-dual interior-point LP solver for gfx950 (CDNA4): the WIDE
// instances of the path (33 .. 64 LP columns, up to 256*EHM3_RS rows -- BASELINE.json configs
// 4 and 5), where the constant block [G | -S | -1] of a commutation (167 KB at config 4) no
// longer fits in LDS and the normal matrix is a real contraction for the matrix cores.
//
// One workgroup of 256 threads (4 wavefronts, one per SIMD) owns one LP
//        min c^T x   s.t.  A x <= b,
// same algorithm, tolerances and acceptance rules as the wave-local solver (ehm_ipm2.h,
// oracle/ipm_numpy.py): Mehrotra predictor-corrector, normal equations, dependent-pivot
// guard, best-iterate tracking, sign-only stop for the suboptimality test.  What changes is
// where the data lives and who does what:
//   * the constant block is read from global memory (L2 resident, shared by every
//     workgroup): a column-major image for the row products (thread = row, coalesced over
//     the rows) and a row-major image padded to 64 columns for the column products (lane =
//     column, 512-byte rows) and for the matrix cores;
//   * M = W^T diag(d) W is formed by v_mfma_f64_16x16x4_f64: the four wavefronts split the
//     rows (K), each keeps the 10 (or 6) lower-triangular 16x16 tiles in registers, no LDS
//     traffic for the operands at all; the partial tiles meet in LDS in a fixed order
//     (bit-reproducible);
//   * the LP variables keep the layout of the block: [ z (n) | beta (p) | t or tau ].  On the
//     MPC rows the weights act through psi = E beta (E = edge matrix of the simplex), so the
//     tiles are transformed M <- T^T M T, T = blockdiag(I, E, 1), and columns a kind does not
//     use are masked to identity; the few extra rows (simplex facets, suboptimality rows,
//     phase-one bound) are dense rank-one terms;
//   * factorisation and triangular solves run in wavefront 0 with lane j = row j in
//     registers (64 = the wavefront width), pivot rows broadcast through LDS, as in
//     ehm_ipm2.h; the m-vectors live in registers, row i = thread + 256 * slot.
// Reference call sites this arithmetic replaces: lib/oracle.py:131,134,166,169,203,276,305,350.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ehm_dev.h"

#ifndef EHM3_RS
#define EHM3_RS 2           // row slots per thread: LP rows <= 256 * EHM3_RS
#endif
#define EHM3_THREADS 256
#ifndef EHM3_LU_CHUNK
#define EHM3_LU_CHUNK 32
#endif
#ifndef EHM3_WAVES_PER_EU
#define EHM3_WAVES_PER_EU 2     // two workgroups per CU: at most 256 registers per lane
#endif
#define EHM3_KERNEL __global__ __launch_bounds__(EHM3_THREADS) \
    __attribute__((amdgpu_waves_per_eu(EHM3_WAVES_PER_EU, EHM3_WAVES_PER_EU)))

#define EHM3_TOL_RES      1e-10
#define EHM3_TOL_GAP      1e-10
#define EHM3_MAX_ITER     40
#define EHM3_STEP_FRAC    0.999
#define EHM3_STEP_FRAC_SAFE 0.99
#define EHM3_STEP_FRAC_LAST 0.9
#define EHM3_ATTEMPTS     3
#define EHM3_PIVOT_REL    1e-13
#define EHM3_PIVOT_BIG    1e128
#define EHM3_STALL_ZONE   1e4
#define EHM3_ACCEPT_MERIT 1e3
#define EHM3_SIGN_RES      1e-7
#define EHM3_SIGN_GAP      0.5
#define EHM3_SIGN_RES_REL  1e-3

#define EHM3_CAT2(a, b) a##b
#define EHM3_CAT(a, b) EHM3_CAT2(a, b)
#define EHM3_NS EHM3_CAT(ehm3_rs, EHM3_RS)

namespace EHM3_NS {

using namespace ehm;

constexpr int NW = 64;              // column capacity = wavefront width
constexpr int NT = EHM3_THREADS;
constexpr int RS = EHM3_RS;
constexpr int LDM = NW + 1;         // odd: rows and columns of the square matrix conflict free
constexpr int MAXE = 16;            // extra rows (p + 3 <= 11)
constexpr int MROWS = NT * RS;

typedef double double2v __attribute__((ext_vector_type(2)));
typedef double double4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// wave / workgroup helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// Keeps everything derived from x inside the current phase: without it LLVM hoists dozens of
// loop-invariant per-thread addresses and comparison masks out of the IPM iteration (and out
// of the item loop of the kernels) and spills them.
__device__ __forceinline__ int pin(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
// 1/x for positive finite x well inside the normal range: v_rcp_f64 + two Newton steps
__device__ __forceinline__ double frcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return r;
}

// Optional phase timing (-DEHM3_PROFILE, experimental builds only): wall-clock ticks of thread 0
// per solver phase, accumulated in g_prof3[] and read by ehm_debug_profile().
#ifdef EHM3_PROFILE
__device__ unsigned long long g_prof3[32];
#define EHM3_TICK(slot)                                                       \
    do {                                                                      \
        const unsigned long long now_ = wall_clock64();                       \
        if (B.tid == 0) atomicAdd(&g_prof3[slot], now_ - B.pt);               \
        B.pt = now_;                                                          \
    } while (0)
#define EHM3_TICK_INIT() B.pt = wall_clock64()
#else
#define EHM3_TICK(slot) do { } while (0)
#define EHM3_TICK_INIT() do { } while (0)
#endif

struct Block {
    int tid, lane, wave;
    double* red;    // [2][4][8] reduction scratch, double buffered
    int flip;
#ifdef EHM3_PROFILE
    mutable unsigned long long pt;
#endif
};

// mx[] -> maxima, sm[] -> sums over the workgroup; every thread receives the same values
// (wave butterflies, then the four wave results combined in a fixed order).
__device__ __forceinline__ void block_reduce(Block& B, double (&mx)[2], double (&sm)[4]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) mx[k] = wave_max(mx[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[k] = wave_sum(sm[k]);
    double* r = B.red + B.flip * 32;
    if (B.lane == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) r[B.wave * 8 + k] = mx[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[B.wave * 8 + 2 + k] = sm[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) mx[k] = fmax(fmax(r[k], r[8 + k]), fmax(r[16 + k], r[24 + k]));
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[k] = ((r[2 + k] + r[10 + k]) + r[18 + k]) + r[26 + k];
    B.flip ^= 1;
}

// ---------------------------------------------------------------------------------------
// LP workspace
// ---------------------------------------------------------------------------------------
struct Lp {
    // constants of the commutation (global memory)
    const double* Wcm;  // column-major [n+p+2][ldc]:  G | -S | -1 | 0
    const double* Wrm;  // row-major    [m_pad][64]  :  G | -S | -1 | 0 ...   (rows >= m zero)
    const double* wv;   // [m]
    const double* cv;   // [n]
    int n, m, p, ldc, m_pad, ntile;
    double sign_floor;  // a sign-only stop must establish |optimum| >= this (EHM_ROUTE_TOL)
    // LDS
    double* M;      // NW x LDM normal matrix, then the packed factor U
    double* dvec;   // nvec   d = lam / s
    double* u0;     // nvec   row-vector inputs of the column products
    double* u1;     // nvec
    double* X;      // [MAXE][NW] extra rows (LP columns)
    double* E;      // p x p edge matrix (row-major), E[r][q] = R[q+1][r] - R[0][r]
    double* c;      // NW objective
    double* x;      // NW iterate
    double* xb;     // NW best iterate
    double* t;      // NW step (LP columns)
    double* xw;     // NW its image in the columns of the block (psi = E beta)
    double* g0;     // NW column products, LP columns
    double* g1;
    double* gw0;    // NW column products in the columns of the block
    double* gw1;
    double* part;   // [4][2][NW] per-wavefront partial column products
    double* db;     // NW original diagonal, then reciprocal pivots
    double* ub;     // NW forward-solve results
    double* dext;   // MAXE  d of the extra rows
    int nvec;       // m_pad + MAXE
    // set by the assembly
    int ne, m_lp;
    int has_beta;   // columns n .. n+p-1 are barycentric weights
    int spec_mpc;   // column n+p enters the MPC rows with coefficient -1 (tau); else only extras
    unsigned long long act;     // bit j: LP column j exists
};

__host__ __device__ inline int m_pad_of(int m) { return (m + 63) & ~63; }
__host__ __device__ inline size_t lp_doubles(int m) {
    const size_t nvec = (size_t)m_pad_of(m) + MAXE;
    return (size_t)NW * LDM + 1 + 3 * nvec + (size_t)MAXE * NW + 64 + 11 * (size_t)NW +
           8 * (size_t)NW + MAXE;
}
__device__ inline void carve_lp(Lp& L, double* base, const DevProblem& P, int d) {
    L.n = P.n; L.m = P.m; L.p = P.p; L.ldc = P.lda2; L.m_pad = P.mpad3;
    L.ntile = (P.n + P.p + 1 + 15) >> 4;
    L.Wcm = P.Wc2 + (size_t)d * P.ncw2 * P.lda2;
    L.Wrm = P.Wr3 + (size_t)d * P.mpad3 * NW;
    L.wv = P.w + (size_t)d * P.m;
    L.cv = P.c;
    L.nvec = L.m_pad + MAXE;
    L.M = base;      base += NW * LDM + 1;
    L.dvec = base;   base += L.nvec;
    L.u0 = base;     base += L.nvec;
    L.u1 = base;     base += L.nvec;
    L.X = base;      base += MAXE * NW;
    L.E = base;      base += 64;
    L.c = base;      base += NW;
    L.x = base;      base += NW;
    L.xb = base;     base += NW;
    L.t = base;      base += NW;
    L.xw = base;     base += NW;
    L.g0 = base;     base += NW;
    L.g1 = base;     base += NW;
    L.gw0 = base;    base += NW;
    L.gw1 = base;    base += NW;
    L.db = base;     base += NW;
    L.ub = base;     base += NW;
    L.part = base;   base += 8 * NW;
    L.dext = base;
    L.ne = 0; L.m_lp = P.m; L.has_beta = 0; L.spec_mpc = 0; L.act = 0;
}

__device__ __forceinline__ double step_fraction(int attempt) {
    return attempt == 0 ? EHM3_STEP_FRAC : (attempt == 1 ? EHM3_STEP_FRAC_SAFE : EHM3_STEP_FRAC_LAST);
}

struct IpmResult {
    double obj;
    double merit;
    double margin;   // lower bound of |optimum| when the solve stopped on its sign, else |obj|
    int iters;
    int status;      // 0 optimal / accepted, 1 stalled
};

// ---------------------------------------------------------------------------------------
// products with the constraint matrix
// ---------------------------------------------------------------------------------------
// L.t (LP columns) -> L.xw (columns of the block); one wavefront, lanes = columns
__device__ __forceinline__ void to_block_columns(const Lp& L, int lane) {
    const int n = L.n, p = L.p;
    double v = 0.0;
    if (lane < n) {
        v = L.t[lane];
    } else if (lane < n + p) {
        if (L.has_beta) {
            const int r = lane - n;
            for (int q = 0; q < p; ++q) v = fma(L.E[r * p + q], L.t[n + q], v);
        }
    } else if (lane == n + p) {
        v = L.spec_mpc ? L.t[lane] : 0.0;
    }
    L.xw[lane] = v;
}

// out[s] = (A t)_i for the rows of this thread; L.t / L.xw hold the vector.  The loads of
// all row slots of a column step are issued together (latency bound: the block comes from L2).
__device__ __forceinline__ void rows_times(const Lp& L, int tid, int wave, double (&out)[RS]) {
    const int ncw = L.n + L.p + 1;
    const size_t ldc = (size_t)L.ldc;
    const double* col[RS];
    bool mpc[RS];
    double a0[RS], a1[RS];
#pragma unroll
    for (int sl = 0; sl < RS; ++sl) {
        const int i = tid + NT * sl;
        mpc[sl] = i < L.m;
        col[sl] = L.Wcm + (mpc[sl] ? i : 0);
        a0[sl] = 0.0;
        a1[sl] = 0.0;
    }
    // wave-uniform skip of slots without MPC rows (rows are dealt in order)
    const int nsl = (L.m - 64 * wave + NT - 1) / NT;       // slots with MPC rows in this wave
    int j = 0;
    // 8 columns per trip, the loads of every row slot issued before the first product
    for (; j + 8 <= ncw; j += 8) {
        double w[RS][8];
#pragma unroll
        for (int sl = 0; sl < RS; ++sl)
            if (sl < nsl) {
#pragma unroll
                for (int u = 0; u < 8; ++u) w[sl][u] = col[sl][(size_t)(j + u) * ldc];
            }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            const double x0 = L.xw[j + u], x1 = L.xw[j + u + 1];
#pragma unroll
            for (int sl = 0; sl < RS; ++sl)
                if (sl < nsl) {
                    a0[sl] = fma(w[sl][u], x0, a0[sl]);
                    a1[sl] = fma(w[sl][u + 1], x1, a1[sl]);
                }
        }
    }
    for (; j < ncw; ++j) {
        const double x0 = L.xw[j];
#pragma unroll
        for (int sl = 0; sl < RS; ++sl)
            if (sl < nsl) a0[sl] = fma(col[sl][(size_t)j * ldc], x0, a0[sl]);
    }
#pragma unroll
    for (int sl = 0; sl < RS; ++sl) {
        const int i = tid + NT * sl;
        double acc = a0[sl] + a1[sl];
        if (!mpc[sl]) {
            acc = 0.0;
            if (i < L.m_lp) {
                const double* xr = L.X + (i - L.m) * NW;
                for (int jj = 0; jj < ncw; ++jj) acc = fma(xr[jj], L.t[jj], acc);
            }
        }
        out[sl] = acc;
    }
}

// g0 = A^T u0 (and g1 = A^T u1) in LP columns.  Entry and exit are workgroup barriers.
template <bool TWO>
__device__ __forceinline__ void cols_times(const Lp& L, const Block& B, const double* u0,
                                           const double* u1) {
    const int lane = pin(B.lane), wave = B.wave;
    {
        const int chunk = L.m_pad >> 2;
        const int r0 = wave * chunk;
        const double* wr = L.Wrm + (size_t)r0 * NW + lane;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        // 16 rows per trip: all 16 loads are issued before the first product (latency bound;
        // the inner loops have constant trip counts so that they really unroll)
        for (int k = 0; k < chunk; k += 16) {
            double w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = wr[(size_t)(k + u) * NW];
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                a0 = fma(w[u], u0[r0 + k + u], a0);
                b0 = fma(w[u + 1], u0[r0 + k + u + 1], b0);
                if (TWO) {
                    a1 = fma(w[u], u1[r0 + k + u], a1);
                    b1 = fma(w[u + 1], u1[r0 + k + u + 1], b1);
                }
            }
        }
        L.part[(wave * 2 + 0) * NW + lane] = a0 + b0;
        if (TWO) L.part[(wave * 2 + 1) * NW + lane] = a1 + b1;
    }
    __syncthreads();
    if (wave == 0) {
        const int n = L.n, p = L.p;
        L.gw0[lane] = ((L.part[lane] + L.part[2 * NW + lane]) + L.part[4 * NW + lane]) +
                      L.part[6 * NW + lane];
        if (TWO)
            L.gw1[lane] = ((L.part[NW + lane] + L.part[3 * NW + lane]) + L.part[5 * NW + lane]) +
                          L.part[7 * NW + lane];
        wsync();
        double r0 = 0.0, r1 = 0.0;
        if (lane < n) {
            r0 = L.gw0[lane];
            if (TWO) r1 = L.gw1[lane];
        } else if (lane < n + p) {
            if (L.has_beta) {
                const int q = lane - n;
                for (int r = 0; r < p; ++r) {
                    const double e = L.E[r * p + q];
                    r0 = fma(e, L.gw0[n + r], r0);
                    if (TWO) r1 = fma(e, L.gw1[n + r], r1);
                }
            }
        } else if (lane == n + p && L.spec_mpc) {
            r0 = L.gw0[lane];
            if (TWO) r1 = L.gw1[lane];
        }
        for (int e = 0; e < L.ne; ++e) {
            const double xe = L.X[e * NW + lane];
            r0 = fma(xe, u0[L.m + e], r0);
            if (TWO) r1 = fma(xe, u1[L.m + e], r1);
        }
        L.g0[lane] = r0;
        if (TWO) L.g1[lane] = r1;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// normal matrix on the matrix cores
// ---------------------------------------------------------------------------------------
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// D element r of lane l is D[i = (l >> 4) + 4 r][j = l & 15]   (cdna_hip_programming.md).
// Tile (I, J), I >= J, of W^T D W:  A = (W[rows, 16 I ..])^T d,  B = W[rows, 16 J ..]: both
// operands are "row 4 ks + (l >> 4), column 16 T + (l & 15)" of the row-major image -- four
// 128-byte segments per load.
// One wavefront's share of the lower-triangular tiles: up to three tiles (I0,J0) (I1,J1)
// (I2,J2) (I < 0: none) over ALL rows.  The tiles are split over the four wavefronts, not the
// rows: no partial sums to combine (bit-reproducible by construction), 24 accumulator
// registers instead of 80.  Operands of chunk c+1 (4 k-steps = 16 rows) are in flight while
// chunk c runs on the matrix cores.
template <int I0, int J0, int I1, int J1, int I2, int J2>
__device__ __forceinline__ void form_tile_set(const Lp& L, int lane) {
    constexpr bool use[4] = {I0 == 0 || J0 == 0 || I1 == 0 || J1 == 0 || I2 == 0 || J2 == 0,
                             I0 == 1 || J0 == 1 || I1 == 1 || J1 == 1 || I2 == 1 || J2 == 1,
                             I0 == 2 || J0 == 2 || I1 == 2 || J1 == 2 || I2 == 2 || J2 == 2,
                             I0 == 3 || J0 == 3 || I1 == 3 || J1 == 3 || I2 == 3 || J2 == 3};
    const int li = lane & 15, lk = lane >> 4;
    double4v C0 = {0.0, 0.0, 0.0, 0.0}, C1 = C0, C2 = C0;
    const int chunks = L.m_pad >> 4;                // 16 rows (4 k-steps) per chunk
    const double* wr = L.Wrm + (size_t)lk * NW + li;
    const double* dv = L.dvec + lk;
    double wc[4][4], wn[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int T = 0; T < 4; ++T) wn[u][T] = use[T] ? wr[(size_t)(4 * u) * NW + 16 * T] : 0.0;
    for (int c = 0; c < chunks; ++c) {
        double d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            d[u] = dv[16 * c + 4 * u];
#pragma unroll
            for (int T = 0; T < 4; ++T) wc[u][T] = wn[u][T];
        }
        if (c + 1 < chunks) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int T = 0; T < 4; ++T)
                    if (use[T]) wn[u][T] = wr[(size_t)(16 * (c + 1) + 4 * u) * NW + 16 * T];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[u][I0] * d[u], wc[u][J0], C0, 0, 0, 0);
            if (I1 >= 0)
                C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[u][I1 < 0 ? 0 : I1] * d[u],
                                                          wc[u][J1 < 0 ? 0 : J1], C1, 0, 0, 0);
            if (I2 >= 0)
                C2 = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[u][I2 < 0 ? 0 : I2] * d[u],
                                                          wc[u][J2 < 0 ? 0 : J2], C2, 0, 0, 0);
        }
    }
    // D element r of this lane is (row (lane >> 4) + 4 r, column lane & 15) of the tile
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rr = lk + 4 * r;
        {
            const double val = C0[r];
            L.M[(16 * I0 + rr) * LDM + 16 * J0 + li] = val;
            if (I0 != J0) L.M[(16 * J0 + li) * LDM + 16 * I0 + rr] = val;
        }
        if (I1 >= 0) {
            const double val = C1[r];
            L.M[(16 * I1 + rr) * LDM + 16 * J1 + li] = val;
            if (I1 != J1) L.M[(16 * J1 + li) * LDM + 16 * I1 + rr] = val;
        }
        if (I2 >= 0) {
            const double val = C2[r];
            L.M[(16 * I2 + rr) * LDM + 16 * J2 + li] = val;
            if (I2 != J2) L.M[(16 * J2 + li) * LDM + 16 * I2 + rr] = val;
        }
    }
}

// W^T diag(d) W for the first 16 NTILE columns of the block; ends with a workgroup barrier.
template <int NTILE>
__device__ __forceinline__ void form_tiles(const Lp& L, const Block& B) {
    const int lane = pin(B.lane), wave = B.wave;
#ifdef EHM3_PROFILE
    const unsigned long long tw0 = clock64();
#endif
    if (NTILE == 4) {           // 10 tiles: 3 + 3 + 2 + 2
        if (wave == 0) form_tile_set<0, 0, 1, 0, 1, 1>(L, lane);
        else if (wave == 1) form_tile_set<2, 0, 2, 1, 2, 2>(L, lane);
        else if (wave == 2) form_tile_set<3, 0, 3, 1, -1, -1>(L, lane);
        else form_tile_set<3, 2, 3, 3, -1, -1>(L, lane);
    } else {                    // 6 tiles: 2 + 2 + 1 + 1
        if (wave == 0) form_tile_set<0, 0, 1, 0, -1, -1>(L, lane);
        else if (wave == 1) form_tile_set<2, 0, 2, 1, -1, -1>(L, lane);
        else if (wave == 2) form_tile_set<1, 1, -1, -1, -1, -1>(L, lane);
        else form_tile_set<2, 2, -1, -1, -1, -1>(L, lane);
    }
#ifdef EHM3_PROFILE
    if (lane == 0) {        // per-wavefront cycles of the tile loop, and where the wavefront runs
        atomicAdd(&g_prof3[20 + wave], (unsigned long long)clock64() - tw0);
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID
        atomicOr(&g_prof3[24 + wave], 1ULL << ((hw >> 4) & 3));
        atomicAdd(&g_prof3[28], 1ULL);
    }
#endif
    EHM3_TICK(13);
    __syncthreads();
}

// M = A^T diag(d) A in LP columns, identity on the columns the LP does not have.
// Entry: L.dvec / L.dext visible.  Exit: workgroup barrier passed.
__device__ __forceinline__ void form_normal_matrix(const Lp& L, const Block& B) {
    int tid = pin(B.tid);
    if (L.ntile >= 4) form_tiles<4>(L, B);
    else form_tiles<3>(L, B);
    EHM3_TICK(14);
    const int n = L.n, p = L.p;
    if (L.has_beta) {
        // psi -> beta on the weight block:  M <- T^T M T,  T = blockdiag(I, E, 1)
        double tmp[8];
        if (tid < NW) {            // columns: row `tid` times E
            double* mr = L.M + tid * LDM + n;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double a = 0.0;
                if (q < p)
                    for (int r = 0; r < p; ++r) a = fma(mr[r], L.E[r * p + q], a);
                tmp[q] = a;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < p) mr[q] = tmp[q];
        }
        __syncthreads();
        if (tid < NW) {            // rows: E^T times column `tid`
            double* mc = L.M + n * LDM + tid;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double a = 0.0;
                if (q < p)
                    for (int r = 0; r < p; ++r) a = fma(L.E[r * p + q], mc[r * LDM], a);
                tmp[q] = a;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < p) mc[q * LDM] = tmp[q];
        }
        __syncthreads();
    }
    EHM3_TICK(15);
    // mask, extra rows, identity: thread -> row tid >> 2, columns 16 (tid & 3) .. +15
    {
        tid = pin(B.tid);
        const int r = tid >> 2, c0 = (tid & 3) * 16;
        unsigned long long mact = L.act;                       // columns with MPC entries
        if (!L.spec_mpc) mact &= ~(1ULL << (n + p));
        const bool mr = (mact >> r) & 1ULL;
        double val[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = c0 + k;
            const bool on = mr && ((mact >> c) & 1ULL);
            const double mv = L.M[r * LDM + c];
            val[k] = on ? mv : 0.0;
        }
        for (int e = 0; e < L.ne; ++e) {
            const double xr = L.X[e * NW + r];
            if (xr != 0.0) {
                const double dx = L.dext[e] * xr;
                const double* xe = L.X + e * NW + c0;
#pragma unroll
                for (int k = 0; k < 16; ++k) val[k] = fma(dx, xe[k], val[k]);
            }
        }
        const bool ar = (L.act >> r) & 1ULL;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = c0 + k;
            L.M[r * LDM + c] = (!ar && c == r) ? 1.0 : val[k];
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// factorisation and triangular solves (wavefront 0, lane j = row j)
// ---------------------------------------------------------------------------------------
// Packed upper-triangular factor: row k keeps its columns (k & ~1) .. NW-1 (an even start keeps
// every row 16-byte aligned); element (k, q) lives at U[uoff(k) + q].
__host__ __device__ constexpr int u_row_start(int k) {
    return (k & 1) ? (k * NW - 2 * (k / 2) * (k / 2)) : (k * NW - 2 * (k / 2) * (k / 2 - 1));
}
__host__ __device__ constexpr int uoff(int k) { return u_row_start(k) - (k & ~1); }

// Step k publishes column k of the current Schur complement (register k of every lane; by
// symmetry it is row k) as row k of the packed factor, which overwrites the square matrix
// (dead: every lane holds its row).  No per-step lane predicates in the arithmetic: finished
// lanes keep computing values nobody reads.  The multipliers L[j][k], j > k, are parked
// column-wise behind U (lcol), so the row registers die with this function.
// Dependent-pivot guard (LIPSOL/PCx) against the original diagonal; rinv_out = 1 / U[lane][lane]
// (guarded).
constexpr int U_SIZE = NW * (NW + 2) / 2;
__host__ __device__ constexpr int lcol(int k) { return U_SIZE + k * (NW - 1) - k * (k - 1) / 2 - k - 1; }
static_assert(U_SIZE + NW * (NW - 1) / 2 <= NW * LDM + 1, "L and U share the matrix region");

__device__ __forceinline__ void lu_factor(const Lp& L, int lane, double& rinv_out) {
    double* U = L.M;
    double row[NW];
    double diag0;
    {
        const double* mrow = L.M + lane * LDM;
#pragma unroll
        for (int q = 0; q < NW; ++q) row[q] = mrow[q];
        diag0 = mrow[lane];
    }
    wsync();
    double rinv_own = 0.0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int kk = k & ~1;
        if (lane >= kk) U[uoff(k) + lane] = row[k];
        // pivot and its original value travel through SGPRs: the reciprocal is ready while the
        // column is still on its way through LDS
        double piv = readlane_d(row[k], k);
        const double orig = readlane_d(diag0, k);
        const bool bad = !(piv > EHM3_PIVOT_REL * orig) || !(piv > 0.0);
        piv = bad ? EHM3_PIVOT_BIG : piv;
        const double rinv = frcp(piv);
        rinv_own = (lane == k) ? rinv : rinv_own;
        const double l = row[k] * rinv;
        if (lane > k) U[lcol(k) + lane] = l;        // element (lane, k) of L
        wsync();
        if (((k + 1) & 1) && k + 1 < NW) {
            const double ukq = U[uoff(k) + k + 1];
            row[k + 1] = fma(-l, ukq, row[k + 1]);
        }
#pragma unroll
        for (int q = (k + 2) & ~1; q < NW; q += 2) {
            const double2v u = *reinterpret_cast<const double2v*>(U + uoff(k) + q);
            row[q] = fma(-l, u.x, row[q]);
            row[q + 1] = fma(-l, u.y, row[q + 1]);
            // at most EHM3_LU_CHUNK broadcast values in flight (register budget)
            if (((q - ((k + 2) & ~1)) / 2) % (EHM3_LU_CHUNK / 2) == EHM3_LU_CHUNK / 2 - 1)
                __builtin_amdgcn_sched_barrier(0);
        }
        // keeps the trailing update of step k in step k (otherwise every FMA chain is sunk to
        // where row[q] is next read and n^2/2 broadcast values stay alive)
#pragma unroll
        for (int q = k + 1; q < NW; ++q) asm volatile("" : "+v"(row[q]));
    }
    rinv_out = rinv_own;
    wsync();
}

// Solve (LU) x = rhs; lane j passes rhs_j and 1/U[j][j]; L.t[j] receives x_j.
// Lanes <= k read the multiplier column k outside its range (in-bounds garbage): their
// running right-hand side is dead by then, y_k / x_k are taken from the broadcast.
__device__ __forceinline__ void lu_solve(const Lp& L, double rinv, double rhs, int lane) {
    double bv = rhs;
    const double* lc = L.M + lane;
    // no LDS stores inside the chains: lane k keeps y_k / x_k itself, so the multiplier and
    // pivot-row loads can all be issued ahead of the serial v_readlane / FMA chain
    double own = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < NW; k0 += 16) {
        double lv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) lv[u] = lc[lcol(k0 + u)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double yk = readlane_d(bv, k0 + u);
            own = (lane == k0 + u) ? yk : own;
            bv = fma(-lv[u], yk, bv);
        }
    }
    bv = own;
    const double* urow = L.M + uoff(lane);
#pragma unroll
    for (int k0 = NW - 16; k0 >= 0; k0 -= 16) {
        double uv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) uv[u] = urow[k0 + u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 15; u >= 0; --u) {
            const double xk = readlane_d(bv * rinv, k0 + u);
            own = (lane == k0 + u) ? xk : own;
            bv = fma(-uv[u], xk, bv);
        }
    }
    L.t[lane] = own;
    wsync();
}

// ---------------------------------------------------------------------------------------
// The solver.  On entry: L.X, L.c, L.E and the flags are set, b in registers (row i =
// tid + 256 * slot; MPC rows first, extras at m ..).  On exit L.xb holds the best iterate.
// Every thread returns the same result.
// ---------------------------------------------------------------------------------------
// gout (optional, LDS, L.p doubles; point problems): gradient of the optimal value with respect
// to the parameter, -S^T lambda = sum_i lambda_i Wcm[n+q][i]; NaN unless the solve converged to
// the tolerances (as in ehm_ipm2.h).
__device__ __forceinline__ IpmResult ipm_solve(const Lp& L, Block& B, const double (&b)[RS],
                                      bool sign_only, double step_frac,
                                      double* gout = nullptr) {
    int tid = pin(B.tid), lane = tid & 63;
    const int wave = B.wave;
    const int m = L.m, m_lp = L.m_lp;
    bool valid[RS];
    double s[RS], lam[RS], v[RS];
    double bmax = 0.0;
#pragma unroll
    for (int sl = 0; sl < RS; ++sl) {
        const int i = tid + NT * sl;
        valid[sl] = i < m_lp;
        v[sl] = valid[sl] ? b[sl] : 0.0;                 // x0 = 0  =>  b - A x0 = b
        s[sl] = valid[sl] ? fmax(b[sl], 1.0) : 1.0;
        lam[sl] = valid[sl] ? 1.0 : 0.0;
        bmax = fmax(bmax, fabs(v[sl]));
    }
    for (int k = tid; k < L.nvec; k += NT) {            // pad rows read as zero
        L.dvec[k] = 0.0;
        L.u0[k] = 0.0;
        L.u1[k] = 0.0;
    }
    double cj = 0.0;
    if (tid < NW) {
        cj = L.c[tid];
        L.x[tid] = 0.0;
        L.xb[tid] = 0.0;
    }
    double mx[2] = {bmax, fabs(cj)}, sm[4] = {0.0, 0.0, 0.0, 0.0};
    block_reduce(B, mx, sm);
    const double bnorm = 1.0 + mx[0];
    const double cnorm = 1.0 + mx[1];
    const double inv_m = 1.0 / (double)m_lp;

    IpmResult res;
    res.obj = 0.0;
    res.merit = 1e300;
    res.margin = 0.0;
    res.iters = 0;
    res.status = 1;
    int stall = 0;
    double rinv_l = 0.0;    // wavefront 0: 1 / U[lane][lane]

    EHM3_TICK_INIT();
#ifdef EHM3_PROFILE
    const unsigned long long prof_c0 = clock64(), prof_w0 = wall_clock64();
#endif
    for (int it = 0; it <= EHM3_MAX_ITER; ++it) {
        // ---- residuals -----------------------------------------------------------------
        tid = pin(B.tid);       // per-thread addresses are re-derived every phase (see pin)
        lane = tid & 63;
        double r_p[RS], rs[RS];
        double rpmax = 0.0, sl_sum = 0.0, vl_sum = 0.0;
#pragma unroll
        for (int sl = 0; sl < RS; ++sl) {
            const int i = tid + NT * sl;
            r_p[sl] = valid[sl] ? (s[sl] - v[sl]) : 0.0;        // A x + s - b
            rpmax = fmax(rpmax, fabs(r_p[sl]));
            sl_sum = fma(s[sl], lam[sl], sl_sum);
            vl_sum = fma(v[sl], lam[sl], vl_sum);
            rs[sl] = frcp(s[sl]);
            if (valid[sl]) {
                L.u0[i] = lam[sl];
                L.u1[i] = lam[sl] * rs[sl] * r_p[sl];
            }
        }
        __syncthreads();
        EHM3_TICK(0);
        cols_times<true>(L, B, L.u0, L.u1);
        EHM3_TICK(1);
        double atl = 0.0, atdr = 0.0, xj = 0.0;
        if (tid < NW) {
            atl = L.g0[tid];
            atdr = L.g1[tid];
            xj = L.x[tid];
        }
        const double r_d = atl + cj;
        mx[0] = fmax(rpmax / bnorm, fabs(r_d) / cnorm);
        mx[1] = 0.0;
        sm[0] = sl_sum;
        sm[1] = vl_sum + xj * atl;         // b^T lam = v^T lam + x^T (A^T lam)
        sm[2] = cj * xj;
        sm[3] = 0.0;
        block_reduce(B, mx, sm);
        const double emax = mx[0];
        const double mu = sm[0] * inv_m;
        const double dobj = -sm[1];
        const double pobj = sm[2];
        const double e_g = fabs(pobj - dobj) / (1.0 + fabs(pobj));
        const double merit = fmax(emax / EHM3_TOL_RES, e_g / EHM3_TOL_GAP);
        // fmax drops NaNs: a non-finite input would pass as "converged" (see ehm_ipm2.h)
        if (!(mu == mu) || !(pobj == pobj) || fabs(pobj) > 1e300) {
            res.merit = 1e300;
            res.obj = pobj;
            res.status = 1;
            res.iters = it;
            break;
        }
        if (merit < res.merit) {
            res.merit = merit;
            res.obj = pobj;
            stall = 0;
            if (tid < NW) L.xb[tid] = xj;
        } else if (res.merit < EHM3_STALL_ZONE) {
            ++stall;
        }
        res.iters = it;
        if (merit <= 1.0) {
            res.status = 0;
            break;
        }
        if (sign_only && emax <= EHM3_SIGN_RES && pobj * dobj > 0.0) {
            const double lo = fmin(fabs(pobj), fabs(dobj));
            if (lo >= L.sign_floor && fabs(pobj - dobj) <= EHM3_SIGN_GAP * lo &&
                emax * (1.0 + fabs(pobj)) <= EHM3_SIGN_RES_REL * lo) {
                res.obj = pobj;
                res.merit = merit;
                res.margin = lo;
                res.status = 0;
                if (tid < NW) L.xb[tid] = xj;
                __syncthreads();
                return res;
            }
        }
        if (stall >= 3 || it == EHM3_MAX_ITER || !(merit == merit)) break;

        // ---- normal matrix and its factorisation ----------------------------------------
        tid = pin(B.tid);
        lane = tid & 63;
#pragma unroll
        for (int sl = 0; sl < RS; ++sl) {
            const int i = tid + NT * sl;
            if (valid[sl]) {
                const double d = lam[sl] * rs[sl];
                L.dvec[i] = d;
                if (i >= m) L.dext[i - m] = d;
            }
        }
        __syncthreads();
        EHM3_TICK(2);
        form_normal_matrix(L, B);
        EHM3_TICK(3);
        const double rhs_aff = -cj - atdr;      // 0 on columns the LP does not have
        tid = pin(B.tid);
        lane = tid & 63;
        if (wave == 0) {
            lu_factor(L, lane, rinv_l);
            EHM3_TICK(4);
            // ---- predictor --------------------------------------------------------------
            lu_solve(L, rinv_l, rhs_aff, lane);
            to_block_columns(L, lane);
            EHM3_TICK(5);
        }
        __syncthreads();
        EHM3_TICK(6);
        tid = pin(B.tid);
        lane = tid & 63;
        double adx[RS];
        rows_times(L, tid, wave, adx);
        EHM3_TICK(7);
        double ds_a[RS], dl_a[RS];
        double rho_p = 0.0, rho_d = 0.0;
#pragma unroll
        for (int sl = 0; sl < RS; ++sl) {
            ds_a[sl] = valid[sl] ? (-r_p[sl] - adx[sl]) : 0.0;
            // dl = -(s lam + lam ds)/s = -lam - (lam/s) ds ;  -dl/lam = 1 + ds/s
            dl_a[sl] = valid[sl] ? (-lam[sl] - lam[sl] * rs[sl] * ds_a[sl]) : 0.0;
            rho_p = fmax(rho_p, -ds_a[sl] * rs[sl]);
            rho_d = fmax(rho_d, valid[sl] ? fma(ds_a[sl], rs[sl], 1.0) : 0.0);
        }
        mx[0] = rho_p;
        mx[1] = rho_d;
        sm[0] = sm[1] = sm[2] = sm[3] = 0.0;
        block_reduce(B, mx, sm);
        double ap = (mx[0] > 1.0) ? 1.0 / mx[0] : 1.0;
        double ad = (mx[1] > 1.0) ? 1.0 / mx[1] : 1.0;
        double mu_aff = 0.0;
#pragma unroll
        for (int sl = 0; sl < RS; ++sl)
            if (valid[sl])
                mu_aff = fma(s[sl] + ap * ds_a[sl], lam[sl] + ad * dl_a[sl], mu_aff);
        mx[0] = mx[1] = 0.0;
        sm[0] = mu_aff;
        block_reduce(B, mx, sm);
        mu_aff = sm[0] * inv_m;
        const double ratio = mu_aff / mu;
        const double sigma = ratio * ratio * ratio;
        const double smu = sigma * mu;

        // ---- corrector ------------------------------------------------------------------
        tid = pin(B.tid);
        lane = tid & 63;
        double corr[RS];
#pragma unroll
        for (int sl = 0; sl < RS; ++sl) {
            corr[sl] = valid[sl] ? (ds_a[sl] * dl_a[sl] - smu) * rs[sl] : 0.0;
            if (valid[sl]) L.u1[tid + NT * sl] = corr[sl];
        }
        __syncthreads();
        EHM3_TICK(8);
        cols_times<false>(L, B, L.u1, L.u1);
        EHM3_TICK(9);
        tid = pin(B.tid);
        lane = tid & 63;
        if (wave == 0) {
            const double rhs = rhs_aff + L.g0[lane];
            lu_solve(L, rinv_l, rhs, lane);
            to_block_columns(L, lane);
        }
        __syncthreads();
        EHM3_TICK(10);
        tid = pin(B.tid);
        lane = tid & 63;
        rows_times(L, tid, wave, adx);
        EHM3_TICK(11);
        double ds[RS], dl[RS];
        rho_p = 0.0;
        rho_d = 0.0;
#pragma unroll
        for (int sl = 0; sl < RS; ++sl) {
            const double rl = frcp(valid[sl] ? lam[sl] : 1.0);
            ds[sl] = valid[sl] ? (-r_p[sl] - adx[sl]) : 0.0;
            // dl = -(s lam + corr_num + lam ds)/s = -lam - corr - (lam/s) ds
            dl[sl] = valid[sl] ? (-lam[sl] - corr[sl] - lam[sl] * rs[sl] * ds[sl]) : 0.0;
            rho_p = fmax(rho_p, -ds[sl] * rs[sl]);
            rho_d = fmax(rho_d, -dl[sl] * rl);
        }
        mx[0] = rho_p;
        mx[1] = rho_d;
        sm[0] = 0.0;
        block_reduce(B, mx, sm);
        ap = (mx[0] > step_frac) ? step_frac / mx[0] : 1.0;
        ad = (mx[1] > step_frac) ? step_frac / mx[1] : 1.0;
        if (tid < NW) L.x[tid] = fma(ap, L.t[tid], xj);
#pragma unroll
        for (int sl = 0; sl < RS; ++sl) {
            if (valid[sl]) {
                s[sl] = fma(ap, ds[sl], s[sl]);
                lam[sl] = fma(ad, dl[sl], lam[sl]);
                v[sl] = fma(-ap, adx[sl], v[sl]);
            }
        }
        __syncthreads();
        EHM3_TICK(12);
    }
    __syncthreads();
#ifdef EHM3_PROFILE
    if (tid == 0) {     // shader clock vs the 100 MHz wall clock
        atomicAdd(&g_prof3[16], (unsigned long long)clock64() - prof_c0);
        atomicAdd(&g_prof3[17], (unsigned long long)wall_clock64() - prof_w0);
    }
#endif
    if (gout) {
        // lam is the multiplier of the LAST iterate = the returned one when the loop left
        // through the convergence test
        const bool conv = (res.status == 0) && (res.merit <= 1.0);
        for (int q0 = 0; q0 < L.p; q0 += 4) {
            double mx2[2] = {0.0, 0.0}, sm4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (q0 + c < L.p) {
                    const double* col = L.Wcm + (size_t)(L.n + q0 + c) * L.ldc;
                    double a = 0.0;
#pragma unroll
                    for (int sl = 0; sl < RS; ++sl) {
                        const int i = tid + NT * sl;
                        if (i < m) a = fma(col[i], lam[sl], a);
                    }
                    sm4[c] = a;
                }
            }
            block_reduce(B, mx2, sm4);
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (q0 + c < L.p) gout[q0 + c] = conv ? sm4[c] : __builtin_nan("");
            }
        }
        __syncthreads();
    }
    if (res.status != 0 && res.merit <= EHM3_ACCEPT_MERIT) res.status = 0;
    res.margin = fabs(res.obj);
    return res;
}

}  // namespace EHM3_NS
