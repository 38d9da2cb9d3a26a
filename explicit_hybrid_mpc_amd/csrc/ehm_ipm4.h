// Workgroup-cooperative interior-point LP solver with the constant block RESIDENT IN LDS, for
// gfx950 (CDNA4): the wide instances of the path (BASELINE.json configs 4 and 5) once the
// epigraph columns are eliminated.
//
// ehm_ipm3.h (round 2) streams the two global images of a commutation's block through every
// phase of every iteration -- 1.3 MB per iteration and workgroup; with thousands of distinct
// prefix blocks per launch (configs[4]) that is 1-2 GB of HBM traffic per launch and the wide
// kernels run at a third of the HBM roofline instead of on the FP64 pipes.  Here:
//   * the z-columns that ehm_ipm2.h eliminates (the trailing range of which every MPC row holds
//     at most one: epigraph variables of an infinity-norm cost, lib/mpc_library.py:530-560) are
//     eliminated for the wide LPs too (oracle/schur_numpy.py is the numpy statement; same
//     iterates to rounding).  57 -> 37 factorised columns at config 4, 39/44/49 -> 27/30/33 for
//     the horizon-6/7/8 tables of configs[4];
//   * what is left of the block, [G_D | -S] (nd0 + p columns), is copied ONCE per workgroup and
//     commutation into LDS (column-major, column stride = 4 mod 8 doubles: the three access
//     patterns below are all bank-conflict free) together with the tables of the eliminated block;
//     every product of every iteration reads LDS.  One workgroup of 512 threads (8 wavefronts,
//     two per SIMD) per CU; thread i owns LP row i;
//   * S0 = (A0' D0 A0)_DD - G Delta^-1 G' on v_mfma_f64_16x16x4_f64: the lower-triangular 16x16
//     tiles in two K-slices over the eight wavefronts (three tile-tasks per SIMD), operands
//     straight from the LDS block, the Schur update as ceil(nE/4) more K-steps;
//   * the factorisation is BLOCKED: 16-column panels eliminated by wavefront 0 with the panel in
//     registers (pivot rows broadcast by v_readlane: no LDS round trip in the dependent chain),
//     trailing updates  S_IJ -= L_Ik D_k L_Jk'  as matrix-core tiles spread over the wavefronts;
//     S = L D L' with the dependent-pivot guard of ehm_ipm2.h; the two Newton solves run on L in
//     place (forward by rows, backward by columns: both conflict free at the odd stride);
//   * the dense extra rows (suboptimality rows, phase-one bound) enter through the reduction of
//     ehm_ipm2.h (dense_prep / solve_full), the simplex rows analytically; nothing is stored for
//     the phase-one column of -1's (synthesised as an operand value).
// Algorithm, tolerances, attempts, sign-only stop, acceptance rules: those of ehm_ipm2.h /
// ehm_ipm3.h / oracle/ipm_numpy.py.
// Reference call sites this arithmetic replaces: lib/oracle.py:131,134,166,169,203,276,305,350.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ehm_dev.h"

#define EHM4_THREADS 512
#define EHM4_KERNEL __global__ __launch_bounds__(EHM4_THREADS)

#define EHM4_TOL_RES      1e-10
#define EHM4_TOL_GAP      1e-10
#define EHM4_MAX_ITER     40
#define EHM4_STEP_FRAC    0.999
#define EHM4_STEP_FRAC_SAFE 0.99
#define EHM4_STEP_FRAC_LAST 0.9
#define EHM4_ATTEMPTS     3
#define EHM4_PIVOT_REL    1e-13
#define EHM4_PIVOT_BIG    1e128
#define EHM4_STALL_ZONE   1e4
#define EHM4_ACCEPT_MERIT 1e3
#define EHM4_SIGN_RES      1e-7
#define EHM4_SIGN_GAP      0.5
#define EHM4_SIGN_RES_REL  1e-3

namespace ehm4 {

using namespace ehm;

constexpr int NT = EHM4_THREADS;
constexpr int NWV = NT / 64;        // wavefronts per workgroup
constexpr int NF = 48;              // capacity of the factorised columns: nd0 + p + 1 <= NF
constexpr int SQ = NF + 1;          // row stride of the square matrix (odd: rows and columns conflict free)
constexpr int GS = NF + 1;          // row stride of the eliminated block's rows
constexpr int MAXNE = 32;           // eliminated columns (elim_detect: nE p <= 256)
constexpr int NC = NF + MAXNE;      // LP columns, internal order [factorised | eliminated]
constexpr int XPAD = 16;            // extra rows behind the MPC rows (p + 3 <= 11)

typedef double double2v __attribute__((ext_vector_type(2)));
typedef double double4v __attribute__((ext_vector_type(4)));

// The arrays of FIXED size sit at compile-time offsets from the start of the workgroup's dynamic
// LDS, in front of everything whose size depends on the problem: their addresses are immediates
// of the DS instructions (as struct members they were 2 SGPRs each, and the first version of this
// solver spilled 700 SGPRs per kernel).
extern __shared__ __attribute__((aligned(16))) char k4_smem[];
constexpr int O_REC = 0;                    // node record (rec_doubles <= 160)
constexpr int O_TH = O_REC + 160;           // parameter / midpoint (8)
constexpr int O_RED = O_TH + 8;             // [2][NWV][8] workgroup reductions
constexpr int O_G = O_RED + 2 * NWV * 8;    // gradient of a point solve (8)
constexpr int O_STASH = O_G + 8;            // first input of a midpoint solve kept across the
                                            // suboptimality-test LP (k4_persist, 8)
constexpr int O_X = O_STASH + 8;            // [2][NC] dense extra rows, internal column order
constexpr int O_XH = O_X + 2 * NC;          // [2][NF] the same, reduced
constexpr int O_HX = O_XH + 2 * NF;         // [2][MAXNE]  X_E Delta^-1
constexpr int O_ID = O_HX + 2 * MAXNE;      // [MAXNE] 1 / Delta_e
constexpr int O_QE = O_ID + MAXNE;          // [MAXNE] r_E / Delta
constexpr int O_C = O_QE + MAXNE;           // [NC] objective
constexpr int O_XV = O_C + NC;              // [NC] iterate
constexpr int O_XB = O_XV + NC;             // [NC] best iterate
constexpr int O_T = O_XB + NC;              // [NC] solution of the last Newton system
constexpr int O_XW = O_T + NC;              // [NF] its psi-form over the image columns
constexpr int O_G0 = O_XW + NF;             // [NC] column products
constexpr int O_G1 = O_G0 + NC;
constexpr int O_GW = O_G1 + NC;             // [2][NF] psi-form column products
constexpr int O_PART = O_GW + 2 * NF;       // [NWV][NF] partial column products
constexpr int O_E = O_PART + NWV * NF;      // [64] edge matrix
constexpr int O_DB = O_E + 64;              // [NF] original diagonal
constexpr int O_PV = O_DB + NF;             // [NF] pivots
constexpr int O_RINV = O_PV + NF;           // [NF] reciprocal pivots
constexpr int O_DEXT = O_RINV + NF;         // [XPAD] d of the extra rows
constexpr int O_DN = O_DEXT + XPAD;         // [8]
constexpr int O_VAR = O_DN + 8;             // the problem-sized part starts here
__device__ __forceinline__ double* lds_at(int off) {
    return reinterpret_cast<double*>(k4_smem) + off;
}

// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double identity, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(identity), __double2loint(v), CTRL,
                                               ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), __double2hiint(v), CTRL,
                                               ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// wave totals by DPP row shifts / row broadcasts (the reduction of ehm_ipm2.h); every lane
// receives the total
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0x111, 0xf>(0.0, v);
    v += dpp_move<0x112, 0xf>(0.0, v);
    v += dpp_move<0x114, 0xf>(0.0, v);
    v += dpp_move<0x118, 0xf>(0.0, v);
    v += dpp_move<0x142, 0xa>(0.0, v);
    v += dpp_move<0x143, 0xc>(0.0, v);
    return lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {
    const double ninf = -__builtin_huge_val();
    v = fmax(v, dpp_move<0x111, 0xf>(ninf, v));
    v = fmax(v, dpp_move<0x112, 0xf>(ninf, v));
    v = fmax(v, dpp_move<0x114, 0xf>(ninf, v));
    v = fmax(v, dpp_move<0x118, 0xf>(ninf, v));
    v = fmax(v, dpp_move<0x142, 0xa>(ninf, v));
    v = fmax(v, dpp_move<0x143, 0xc>(ninf, v));
    return lane63(v);
}
// sum over the four lanes of a quad, returned to all four (quad_perm [1,0,3,2], [2,3,0,1])
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_move<0xB1, 0xf>(0.0, v);
    v += dpp_move<0x4E, 0xf>(0.0, v);
    return v;
}
__device__ __forceinline__ double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
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

// Optional phase timing (-DEHM4_PROFILE, experimental builds only): shader-clock cycles of thread 0
// per solver phase -- wavefront 0's time line, barrier waits included, i.e. the critical path --
// accumulated in g_prof4[] and read by ehm_k4_profile() (tools/k4_phases.py prints the table).
#ifdef EHM4_PROFILE
__device__ unsigned long long g_prof4[40];
#define EHM4_TICK(slot)                                                       \
    do {                                                                      \
        const unsigned long long now_ = clock64();                            \
        if (B.tid == 0) atomicAdd(&g_prof4[slot], now_ - B.pt);               \
        B.pt = now_;                                                          \
    } while (0)
#define EHM4_TICK_INIT() B.pt = clock64()
#else
#define EHM4_TICK(slot) do { } while (0)
#define EHM4_TICK_INIT() do { } while (0)
#endif

struct Ctx {
    int tid, lane, wave;
    int flip;       // which half of the double-buffered reduction scratch (O_RED) is next
#ifdef EHM4_PROFILE
    mutable unsigned long long pt;
#endif
};

// mx[] -> maxima, sm[] -> sums over the workgroup (the first NMX / NSM entries); every thread
// receives the same values (wave reductions, then the eight wave results combined in a fixed order)
template <int NMX, int NSM>
__device__ __forceinline__ void block_reduce(Ctx& B, double (&mx)[2], double (&sm)[4]) {
#pragma unroll
    for (int k = 0; k < NMX; ++k) mx[k] = wave_max(mx[k]);
#pragma unroll
    for (int k = 0; k < NSM; ++k) sm[k] = wave_sum(sm[k]);
    double* r = lds_at(O_RED) + B.flip * (NWV * 8);
    if (B.lane == 0) {
#pragma unroll
        for (int k = 0; k < NMX; ++k) r[B.wave * 8 + k] = mx[k];
#pragma unroll
        for (int k = 0; k < NSM; ++k) r[B.wave * 8 + 2 + k] = sm[k];
    }
    __syncthreads();
    // lane w < 8 fetches wave w's partial; three DPP row shifts leave the total in lane 7 (a fixed
    // order: bit-reproducible), v_readlane hands it to everybody -- 1 LDS load per quantity instead
    // of 8 (a wavefront issues one ds_read_b64 per ~19 ticks: tools/micro/pipe_rates.hip)
    const int wl = B.lane & 7;
#pragma unroll
    for (int k = 0; k < NMX; ++k) {
        double v = r[wl * 8 + k];
        v = fmax(v, dpp_move<0x111, 0xf>(-__builtin_huge_val(), v));
        v = fmax(v, dpp_move<0x112, 0xf>(-__builtin_huge_val(), v));
        v = fmax(v, dpp_move<0x114, 0xf>(-__builtin_huge_val(), v));
        mx[k] = readlane_d(v, 7);
    }
#pragma unroll
    for (int k = 0; k < NSM; ++k) {
        double v = r[wl * 8 + 2 + k];
        v += dpp_move<0x111, 0xf>(0.0, v);
        v += dpp_move<0x112, 0xf>(0.0, v);
        v += dpp_move<0x114, 0xf>(0.0, v);
        sm[k] = readlane_d(v, 7);
    }
    B.flip ^= 1;
}

// ---------------------------------------------------------------------------------------
// LDS layout
// ---------------------------------------------------------------------------------------
// What a workgroup keeps of one commutation (shared by every LP it solves against it).
struct Blk {
    double* Wb;     // [ncb][ld]  G_D | -S, column-major, rows m .. ld-1 zero
    double* aE;     // [m4]  entry of row i in ITS eliminated column (0: none)
    double* eval;   // [nE][LE]  entries of eliminated column e (0-padded)
    int* erow;      // [nE][LE]  their rows (0-padded)
    int* eidx;      // [m4]  eliminated column of row i (0 where aE = 0)
    const double* wv;   // [m]  right-hand side of the commutation (global)
    const double* cv;   // [n]  cost of the z-columns, ORIGINAL order (global)
    int n, m, p, nd0, nE, LE;
    int ncb;        // nd0 + p image columns
    int nrf;        // nd0 + p + 1 factorised LP columns [z_D | beta | t or tau]
    int ld;         // column stride (doubles), = 4 mod 8, >= m4
    int m4;         // m rounded up to a multiple of 4; the extra rows sit at m4 ..
    int srows;      // rows of the square matrix that exist (nrf rounded up to 2)
};
#ifndef EHM4_LDMODE
#define EHM4_LDMODE 0
#endif
__host__ __device__ inline int ld_of(int m) {
#if EHM4_LDMODE == 1
    int ld = (m + 1) & ~1;          // experiment: = 2 mod 4 (conflict free for the tile operands)
    while ((ld & 3) != 2) ld += 2;
    return ld;
#else
    int ld = (m + 3) & ~3;
    while ((ld & 7) != 4) ld += 4;
    return ld;
#endif
}
// One ds_read_b64 that hipcc may not fuse with its neighbours (ds_read2_b64 is served at half the
// bytes per clock: MI355X_MICROARCH.md, LDS table)
typedef __attribute__((address_space(3))) const volatile double lds_cvdouble;
__device__ __forceinline__ double lds1(const double* p) {
    return *(lds_cvdouble*)p;      // (measured on config 4: tile phase 21.1k -> 17.8k ticks per iteration)
}
__host__ __device__ inline size_t blk_doubles(int m, int p, int nd0, int nE, int LE) {
    const size_t m4 = ((size_t)m + 3) & ~(size_t)3;
    const size_t tab = (size_t)nE * LE + ((size_t)nE * LE + m4 + 1) / 2;
    return (((size_t)(nd0 + p) * ld_of(m) + m4 + tab) + 1) & ~(size_t)1;
}
__device__ inline void carve_blk(Blk& S, double* base, const DevProblem& P) {
    S.n = P.n; S.m = P.m; S.p = P.p; S.nd0 = P.nd0; S.nE = P.n - P.nd0; S.LE = P.LE4;
    S.ncb = P.nd0 + P.p;
    S.nrf = S.ncb + 1;
    S.ld = ld_of(P.m);
    S.m4 = (P.m + 3) & ~3;
    S.srows = (S.nrf + 1) & ~1;
    S.Wb = base;
    S.aE = base + (size_t)S.ncb * S.ld;
    S.eval = S.aE + S.m4;
    S.erow = reinterpret_cast<int*>(S.eval + (size_t)S.nE * S.LE);
    S.eidx = S.erow + (size_t)S.nE * S.LE;
    S.wv = P.w;
    S.cv = P.c;
}
// all threads of the workgroup; the caller brackets it with __syncthreads()
__device__ inline void load_blk(const Blk& S, const DevProblem& P, int d, int tid) {
    const double* src = P.Wc4 + (size_t)d * P.tot4;
    const int lda = P.lda4, ld = S.ld, m = S.m;
    for (int j = 0; j < S.ncb; ++j) {
        const double* col = src + (size_t)j * lda;
        double* dst = S.Wb + (size_t)j * ld;
        for (int i = tid; i < ld; i += NT) dst[i] = (i < m) ? col[i] : 0.0;
    }
    if (S.nE > 0) {
        // layout of DevProblem::Wc4: [G_D | -S | -1 | 0 | aE] then eval, erow, eidx
        const double* aE = src + (size_t)(P.nd0 + P.p + 2) * lda;
        const double* ev = src + (size_t)P.ncw4 * lda;
        const int* er = reinterpret_cast<const int*>(ev + (size_t)S.nE * S.LE);
        const int* ei = er + (size_t)S.nE * S.LE;
        for (int i = tid; i < S.m4; i += NT) {
            S.aE[i] = (i < m) ? aE[i] : 0.0;
            S.eidx[i] = (i < m) ? ei[i] : 0;
        }
        for (int k = tid; k < S.nE * S.LE; k += NT) {
            S.eval[k] = ev[k];
            S.erow[k] = er[k];
        }
    }
}

// Private to the LP in flight.  The arrays of FIXED size sit at compile-time offsets from the start
// of the workgroup's dynamic LDS (in front of everything whose size depends on the problem), so
// their addresses are immediates of the DS instructions: as struct members they were 2 SGPRs
// each, and the first version of this solver spilled 700 SGPRs per kernel.
struct Lp {
    double* M;      // [srows][SQ] square matrix, then L (strictly lower) in place
    double* gE;     // [nE][GS]  (A0' D0 A0)_DE, beta-form after the transform
    double* u0;     // [m4 + XPAD] row vectors (extras at m4 ..)
    double* u1;
    __device__ __forceinline__ double* X() const { return lds_at(O_X); }
    __device__ __forceinline__ double* xh() const { return lds_at(O_XH); }
    __device__ __forceinline__ double* hX() const { return lds_at(O_HX); }
    __device__ __forceinline__ double* iD() const { return lds_at(O_ID); }
    __device__ __forceinline__ double* qE() const { return lds_at(O_QE); }
    __device__ __forceinline__ double* c() const { return lds_at(O_C); }
    __device__ __forceinline__ double* x() const { return lds_at(O_XV); }
    __device__ __forceinline__ double* xb() const { return lds_at(O_XB); }
    __device__ __forceinline__ double* t() const { return lds_at(O_T); }
    __device__ __forceinline__ double* xw() const { return lds_at(O_XW); }
    __device__ __forceinline__ double* g0() const { return lds_at(O_G0); }
    __device__ __forceinline__ double* g1() const { return lds_at(O_G1); }
    __device__ __forceinline__ double* gw() const { return lds_at(O_GW); }
    __device__ __forceinline__ double* part() const { return lds_at(O_PART); }
    __device__ __forceinline__ double* E() const { return lds_at(O_E); }
    __device__ __forceinline__ double* db() const { return lds_at(O_DB); }
    __device__ __forceinline__ double* pv() const { return lds_at(O_PV); }
    __device__ __forceinline__ double* rinv() const { return lds_at(O_RINV); }
    __device__ __forceinline__ double* dext() const { return lds_at(O_DEXT); }
    __device__ __forceinline__ double* dn() const { return lds_at(O_DN); }
    int nsx;        // extra rows 0 .. nsx-1 are the simplex rows (p + 1 or 0), analytic
    int kd;         // dense extra rows (0 .. 2)
    int has_beta;   // columns nd0 .. nd0+p-1 are barycentric weights
    int spec_mpc;   // column nd0+p enters every MPC row with coefficient -1 (tau)
    unsigned long long act;     // bit j: factorised LP column j exists
    double sign_floor;
};
// doubles of the square-matrix region: the matrix, and never less than the partial column
// products that borrow it (cols_times<true>)
__host__ __device__ inline size_t m_doubles(int srows) {
    const size_t a = (size_t)srows * SQ, b = (size_t)NWV * NF;
    return ((a > b ? a : b) + 1) & ~(size_t)1;
}
// the problem-sized part of the LP workspace (behind O_VAR)
__host__ __device__ inline size_t lp_doubles(int m, int p, int nd0, int nE) {
    const size_t m4 = ((size_t)m + 3) & ~(size_t)3;
    const int srows = ((nd0 + p + 1) + 1) & ~1;
    const size_t gE = ((size_t)nE * GS + 1) & ~(size_t)1;
    return m_doubles(srows) + gE + 2 * (m4 + XPAD);
}
__device__ inline void carve_lp(Lp& L, double* base, const Blk& S) {
    L.M = base;     base += m_doubles(S.srows);
    L.gE = base;    base += ((size_t)S.nE * GS + 1) & ~(size_t)1;
    L.u0 = base;    base += S.m4 + XPAD;
    L.u1 = base;
    L.nsx = 0; L.kd = 0; L.has_beta = 0; L.spec_mpc = 0; L.act = 0; L.sign_floor = 0.0;
}
// internal index of the z-column with ORIGINAL index j
__device__ __forceinline__ int zcol(const Blk& S, int j) {
    return (j < S.nd0) ? j : (S.nrf + j - S.nd0);
}

__device__ __forceinline__ double step_fraction(int attempt) {
    return attempt == 0 ? EHM4_STEP_FRAC : (attempt == 1 ? EHM4_STEP_FRAC_SAFE : EHM4_STEP_FRAC_LAST);
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
// psi-form of L.t() over the image columns: one wavefront, lanes = columns (wsync at the end)
__device__ __forceinline__ void to_block_columns(const Blk& S, const Lp& L, int lane) {
    const int nd0 = S.nd0, p = S.p;
    if (lane < NF) {
        double v = 0.0;
        if (lane < nd0) {
            v = L.t()[lane];
        } else if (lane < nd0 + p) {
            if (L.has_beta) {
                const int r = lane - nd0;
                for (int q = 0; q < p; ++q) v = fma(L.E()[r * p + q], L.t()[nd0 + q], v);
            }
        } else if (lane == nd0 + p) {
            v = L.spec_mpc ? L.t()[lane] : 0.0;
        }
        L.xw()[lane] = v;
    }
    wsync();
}

// (A t)_i for the row of this thread; L.xw / L.t / L.dn[4..5] hold the vector (see solve_full).
__device__ __forceinline__ double rows_times(const Blk& S, const Lp& L, int tid) {
    const int m = S.m, ld = S.ld, ncb = S.ncb;
    const int xb = S.m4;
    double acc = 0.0;
    if (tid < m) {
        const double* col = S.Wb + tid;
        double a1 = 0.0;
        int j = 0;
        for (; j + 8 <= ncb; j += 8) {      // eight columns per trip, their loads issued together
            double w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = lds1(col + (size_t)(j + u) * ld);
            const double2v xa = *reinterpret_cast<const double2v*>(L.xw() + j);
            const double2v xc = *reinterpret_cast<const double2v*>(L.xw() + j + 2);
            const double2v xe = *reinterpret_cast<const double2v*>(L.xw() + j + 4);
            const double2v xg = *reinterpret_cast<const double2v*>(L.xw() + j + 6);
            acc = fma(w[0], xa.x, acc);
            a1 = fma(w[1], xa.y, a1);
            acc = fma(w[2], xc.x, acc);
            a1 = fma(w[3], xc.y, a1);
            acc = fma(w[4], xe.x, acc);
            a1 = fma(w[5], xe.y, a1);
            acc = fma(w[6], xg.x, acc);
            a1 = fma(w[7], xg.y, a1);
        }
        for (; j < ncb; ++j) acc = fma(col[(size_t)j * ld], L.xw()[j], acc);
        acc += a1;
        acc -= L.xw()[ncb];                         // the column of -1's (0 unless spec_mpc)
        if (S.nE > 0) acc = fma(S.aE[tid], L.t()[S.nrf + S.eidx[tid]], acc);
    } else if (tid >= xb && tid < xb + L.nsx) {
        const int e = tid - xb;
        if (e < S.p) {
            acc = -L.t()[S.nd0 + e];
        } else {
            for (int q = 0; q < S.p; ++q) acc += L.t()[S.nd0 + q];
        }
    } else if (tid >= xb + L.nsx && tid < xb + L.nsx + L.kd) {
        acc = L.dn()[4 + tid - xb - L.nsx];
    }
    return acc;
}

// g0 = A^T u0 (and g1 = A^T u1), internal column order.  Entry: u0 / u1 visible (a workgroup
// barrier passed); exit: a workgroup barrier passed.  p1 = where the partials of the second
// vector go (NWV x NF doubles; the square matrix is free when TWO is asked for).
// No lane-divergent control flow: a lane whose column is not in the image reads column 0 and
// multiplies by a 0 mask; the column of -1's of the phase-one kinds is a constant added to that.
template <bool TWO>
__device__ __forceinline__ void cols_times(const Blk& S, const Lp& L, const Ctx& B,
                                           const double* u0, const double* u1, double* p1) {
    const int lane = pin(B.lane), wave = B.wave;
    {
        // K-slices over the wavefronts; lanes = (column of a group of 16, row offset of 4).  The
        // lane's entries of the row vectors are fetched once for all column groups (a wavefront
        // issues one ds_read_b64 per ~19 ticks: the loads are what this loop costs).
        const int lr = lane & 3, lc = lane >> 2;
        const int kc = (((S.m4 >> 2) + NWV - 1) / NWV) << 2;        // rows per wavefront (<= 64)
        const int r0 = wave * kc + lr;
        const int r1 = (wave * kc + kc < S.m4) ? wave * kc + kc : S.m4;
        const int ncol = S.ncb + (L.spec_mpc ? 1 : 0);
        double x[16], y[16];
        int ro[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = r0 + 4 * u;
            const bool in = r < r1;
            ro[u] = in ? r : 0;
            x[u] = u0[ro[u]];
            y[u] = TWO ? u1[ro[u]] : 0.0;
            x[u] = in ? x[u] : 0.0;
            y[u] = in ? y[u] : 0.0;
        }
        const int nu = (kc + 3) >> 2;       // row steps of a lane (wave-uniform)
#pragma unroll
        for (int cg = 0; cg < NF / 16; ++cg) {
            const int col = 16 * cg + lc;
            double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;
            if (16 * cg + 16 <= S.ncb) {        // (wave-uniform) every column of the group is in the image
                const double* wp = S.Wb + (size_t)col * S.ld;
#pragma unroll
                for (int u0_ = 0; u0_ < 16; u0_ += 4) {
                    if (u0_ < nu) {             // (wave-uniform)
                        double w[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) w[u] = lds1(wp + ro[u0_ + u]);
#pragma unroll
                        for (int u = 0; u < 4; u += 2) {
                            a0 = fma(w[u], x[u0_ + u], a0);
                            c0 = fma(w[u + 1], x[u0_ + u + 1], c0);
                            if (TWO) {
                                a1 = fma(w[u], y[u0_ + u], a1);
                                c1 = fma(w[u + 1], y[u0_ + u + 1], c1);
                            }
                        }
                    }
                }
                a0 = quad_sum(a0 + c0);
                if (TWO) a1 = quad_sum(a1 + c1);
            } else if (16 * cg < ncol) {        // (wave-uniform) the group that holds the last columns
                const double mk = (col < S.ncb) ? 1.0 : 0.0;
                const double un = (col == S.ncb && L.spec_mpc) ? -1.0 : 0.0;
                const double* wp = S.Wb + (size_t)((col < S.ncb) ? col : 0) * S.ld;
#pragma unroll
                for (int u0_ = 0; u0_ < 16; u0_ += 4) {
                    if (u0_ < nu) {             // (wave-uniform)
                        double w[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) w[u] = lds1(wp + ro[u0_ + u]);
#pragma unroll
                        for (int u = 0; u < 4; u += 2) {
                            const double f0 = fma(w[u], mk, un), f1 = fma(w[u + 1], mk, un);
                            a0 = fma(f0, x[u0_ + u], a0);
                            c0 = fma(f1, x[u0_ + u + 1], c0);
                            if (TWO) {
                                a1 = fma(f0, y[u0_ + u], a1);
                                c1 = fma(f1, y[u0_ + u + 1], c1);
                            }
                        }
                    }
                }
                a0 = quad_sum(a0 + c0);
                if (TWO) a1 = quad_sum(a1 + c1);
            }
            if (lr == 0) {
                L.part()[wave * NF + col] = a0;
                if (TWO) p1[wave * NF + col] = a1;
            }
        }
    }
    if (wave == 1) {
        // the eliminated columns: lane e gathers the (few) rows of column e, four at a time
        const int nrf = S.nrf, xb = S.m4;
        const bool el = lane < S.nE;
        const int eb = (el ? lane : 0) * S.LE;
        double e0 = 0.0, e1 = 0.0;
        for (int k = 0; k < S.LE; k += 4) {
            int i[4];
            double a[4], w0[4], w1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                i[u] = S.erow[eb + k + u];
                a[u] = S.eval[eb + k + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w0[u] = u0[i[u]];
                w1[u] = TWO ? u1[i[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                e0 = fma(a[u], w0[u], e0);
                if (TWO) e1 = fma(a[u], w1[u], e1);
            }
        }
        for (int r = 0; r < L.kd; ++r) {
            const double xe = L.X()[r * NC + nrf + (el ? lane : 0)];
            e0 = fma(xe, u0[xb + L.nsx + r], e0);
            if (TWO) e1 = fma(xe, u1[xb + L.nsx + r], e1);
        }
        if (el) {
            L.g0()[nrf + lane] = e0;
            if (TWO) L.g1()[nrf + lane] = e1;
        }
    }
    EHM4_TICK(20);
    __syncthreads();
    if (wave == 0) {
        const int nd0 = S.nd0, p = S.p, nrf = S.nrf, xb = S.m4;
        if (lane < NF) {
            double q0[NWV], q1[NWV];
#pragma unroll
            for (int w = 0; w < NWV; ++w) {
                q0[w] = L.part()[w * NF + lane];
                q1[w] = TWO ? p1[w * NF + lane] : 0.0;
            }
            double s0 = q0[0], s1 = q1[0];
#pragma unroll
            for (int w = 1; w < NWV; ++w) {
                s0 += q0[w];
                s1 += q1[w];
            }
            L.gw()[lane] = s0;
            if (TWO) L.gw()[NF + lane] = s1;
        }
        wsync();
        double r0 = 0.0, r1 = 0.0;
        {
            const int jl = (lane < NF) ? lane : (NF - 1);
            const int q = lane - nd0;
            const bool isb = q >= 0 && q < p;
            const double gl0 = L.gw()[jl], gl1 = TWO ? L.gw()[NF + jl] : 0.0;
            // psi -> beta on the weight columns:  g_beta = E' g_psi
            double b0 = 0.0, b1 = 0.0;
            const int qc = isb ? q : 0;
            for (int r = 0; r < p; ++r) {
                const double e = L.E()[r * p + qc];
                b0 = fma(e, L.gw()[nd0 + r], b0);
                if (TWO) b1 = fma(e, L.gw()[NF + nd0 + r], b1);
            }
            if (L.nsx > 0) {        // the simplex rows: -1 in row q, +1 in the row of the sum
                b0 += u0[xb + p] - u0[xb + qc];
                if (TWO) b1 += u1[xb + p] - u1[xb + qc];
            }
            const bool lin = lane < nd0 || (lane == nd0 + p && L.spec_mpc);
            r0 = lin ? gl0 : ((isb && L.has_beta) ? b0 : 0.0);
            r1 = lin ? gl1 : ((isb && L.has_beta) ? b1 : 0.0);
        }
        for (int r = 0; r < L.kd; ++r) {
            const double ud0 = u0[xb + L.nsx + r], ud1 = TWO ? u1[xb + L.nsx + r] : 0.0;
            const double xd = L.X()[r * NC + ((lane < NF) ? lane : 0)];
            r0 = fma(xd, ud0, r0);
            if (TWO) r1 = fma(xd, ud1, r1);
        }
        if (lane < nrf) {
            const bool on = (L.act >> lane) & 1ULL;
            L.g0()[lane] = on ? r0 : 0.0;
            if (TWO) L.g1()[lane] = on ? r1 : 0.0;
        }
    }
    __syncthreads();
    EHM4_TICK(21);
}

// ---------------------------------------------------------------------------------------
// normal matrix
// ---------------------------------------------------------------------------------------
// G = (A0' D0 A0)_DE and Delta of the eliminated block, psi-form:
//     g_re = sum_{i in rows(e)} d_i a_ir a_ie ,   Delta_e = sum_{i in rows(e)} d_i a_ie^2 .
// Task (e, r) per thread, r over the image columns and the synthesised one (mask / constant as in
// cols_times: no divergent loads).
__device__ __forceinline__ void form_eliminated(const Blk& S, const Lp& L, const double* dvec,
                                                int tid) {
    const int ncp = S.ncb + 1;
    const int ntask = S.nE * ncp;
    for (int t0 = 0; t0 < ntask; t0 += NT) {
        const int t = t0 + tid;
        const bool in = t < ntask;
        const int tc = in ? t : 0;
        const int e = tc / ncp, r = tc - e * ncp;
        const int eb = e * S.LE;
        const bool img = r < S.ncb;
        const double* col = S.Wb + (size_t)(img ? r : 0) * S.ld;
        const double mk = img ? 1.0 : 0.0;
        const double un = img ? 0.0 : (L.spec_mpc ? -1.0 : 0.0);
        double g = 0.0, dl = 0.0;
        for (int k = 0; k < S.LE; k += 4) {
            int i[4];
            double ev[4], dv[4], av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                i[u] = S.erow[eb + k + u];
                ev[u] = S.eval[eb + k + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dv[u] = dvec[i[u]];
                av[u] = col[i[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double de = dv[u] * ev[u];
                g = fma(de, fma(av[u], mk, un), g);
                dl = fma(de, ev[u], dl);
            }
        }
        if (in) {
            L.gE[e * GS + r] = g;
            if (r == 0) L.iD()[e] = frcp(dl);   // Delta_e > 0: d > 0, the column is not empty
        }
    }
}

// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// D element r of lane l is D[i = (l >> 4) + 4 r][j = l & 15]   (cdna_hip_programming.md).
// Tile (I, J), I >= J, of W' D W:  A = (W[rows, 16 I ..])' d,  B = W[rows, 16 J ..]: both operands
// are "row 4 ks + (l >> 4), column 16 T + (l & 15)" of the LDS block -- address
// (l & 15) ld + (l >> 4) + const, conflict free at ld = 4 mod 8.
// One wavefront's task: up to two tiles (I0,J0) (I1,J1) (I1 < 0: one) over the K-slice
// `slice` of two (rows, then the Schur update  -G Delta^-1 G'  as ceil(nE/4) more K-steps in slice
// 1).  Slice 0 stores, slice 1 adds after the barrier the caller places between them.
template <int I0, int J0, int I1, int J1>
__device__ __forceinline__ void form_tile_task(const Blk& S, const Lp& L, const double* dvec,
                                               int lane, int slice, double4v& C0, double4v& C1) {
    constexpr bool use[3] = {I0 == 0 || J0 == 0 || I1 == 0 || J1 == 0,
                             I0 == 1 || J0 == 1 || I1 == 1 || J1 == 1,
                             I0 == 2 || J0 == 2 || I1 == 2 || J1 == 2};
    const int li = lane & 15, lk = lane >> 4;
    const int ksteps = S.m4 >> 2;
    const int kh = (ksteps + 1) >> 1;
    const int k0 = slice ? kh : 0, k1 = slice ? ksteps : kh;
    // a lane whose column is not in the image reads the last image column instead: rows and
    // columns >= ncb of the result are rewritten afterwards (fix_last_columns), so the loop is
    // loads, one multiply per A operand and the matrix instructions -- nothing else shares the
    // double-precision pipe with them
    const double* wp[3];
#pragma unroll
    for (int T = 0; T < 3; ++T) {
        const int col = 16 * T + li;
        wp[T] = S.Wb + (size_t)((col < S.ncb) ? col : (S.ncb - 1)) * S.ld + lk;
    }
    const double* dv = dvec + lk;
    C0 = (double4v){0.0, 0.0, 0.0, 0.0};
    C1 = C0;
    double4v D0 = C0, D1 = C0;      // second accumulators: consecutive instructions are independent
    // four K-steps (16 rows) per trip; the operands of the next trip are in flight while this
    // one runs on the matrix cores
    double wn[4][3], dn_[4];
    const int ntrip = (k1 - k0) >> 2;
    int k = k0;
    if (ntrip > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            dn_[u] = lds1(dv + 4 * (k + u));
#pragma unroll
            for (int T = 0; T < 3; ++T)
                if (use[T]) wn[u][T] = lds1(wp[T] + 4 * (k + u));
        }
    }
    for (int trip = 0; trip < ntrip; ++trip, k += 4) {
        double wc[4][3], dc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            dc[u] = dn_[u];
#pragma unroll
            for (int T = 0; T < 3; ++T)
                if (use[T]) wc[u][T] = wn[u][T];
        }
        if (trip + 1 < ntrip) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dn_[u] = lds1(dv + 4 * (k + 4 + u));
#pragma unroll
                for (int T = 0; T < 3; ++T)
                    if (use[T]) wn[u][T] = lds1(wp[T] + 4 * (k + 4 + u));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[u][I0] * dc[u], wc[u][J0], C0, 0, 0, 0);
            if (I1 >= 0)
                C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[u][I1 < 0 ? 0 : I1] * dc[u],
                                                          wc[u][J1 < 0 ? 0 : J1], C1, 0, 0, 0);
            D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[u + 1][I0] * dc[u + 1], wc[u + 1][J0], D0,
                                                      0, 0, 0);
            if (I1 >= 0)
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[u + 1][I1 < 0 ? 0 : I1] * dc[u + 1],
                                                          wc[u + 1][J1 < 0 ? 0 : J1], D1, 0, 0, 0);
        }
    }
    for (; k < k1; ++k) {
        const double d = dv[4 * k];
        double w[3];
#pragma unroll
        for (int T = 0; T < 3; ++T)
            if (use[T]) w[T] = wp[T][4 * k];
        C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[I0] * d, w[J0], C0, 0, 0, 0);
        if (I1 >= 0)
            C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[I1 < 0 ? 0 : I1] * d, w[J1 < 0 ? 0 : J1],
                                                      C1, 0, 0, 0);
    }
    if (slice == 1) {
        // Schur update of the eliminated block:  - G Delta^-1 G'  (K = 4 columns e per instruction)
        for (int e0 = 0; e0 < S.nE; e0 += 4) {
            const int e = e0 + lk;
            const bool in = e < S.nE;
            const int ec = in ? e : 0;
            const double id = in ? -L.iD()[ec] : 0.0;
            double g[3];
#pragma unroll
            for (int T = 0; T < 3; ++T) {
                const int col = 16 * T + li;
                if (use[T]) g[T] = L.gE[ec * GS + col];     // (col <= 47 < GS)
                if (use[T]) g[T] = (col < S.ncb) ? g[T] : 0.0;      // rows / columns >= ncb: fix-up
            }
            D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(g[I0] * id, g[J0], D0, 0, 0, 0);
            if (I1 >= 0)
                D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(g[I1 < 0 ? 0 : I1] * id,
                                                          g[J1 < 0 ? 0 : J1], D1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        C0[r] += D0[r];
        C1[r] += D1[r];
    }
}
// D element r of this lane is (row (lane >> 4) + 4 r, column lane & 15) of the tile
template <int I, int J, bool ADD>
__device__ __forceinline__ void store_tile(const Blk& S, const Lp& L, int lane, const double4v& C) {
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + lk + 4 * r, col = 16 * J + li;
        // (diagonal tiles: the lower triangle is mirrored, so the matrix is EXACTLY symmetric)
        if (row < S.srows && col < S.srows && (I != J || row >= col)) {
            double* a = L.M + row * SQ + col;
            const double val = ADD ? (*a + C[r]) : C[r];
            *a = val;
            L.M[col * SQ + row] = val;
        }
    }
}
// Tasks of the eight wavefronts (wavefronts w and w + 4 share a SIMD: three tile-slices each):
//   three column tiles : w0,w1: (0,0)+(1,0)   w2,w3: (2,0)+(2,1)   w4,w5: (1,1)   w6,w7: (2,2)
//   two column tiles   : w0,w1: (0,0)   w2,w3: (1,0)   w4,w5: (1,1)   w6,w7: idle
// even wavefronts take K-slice 0, odd ones slice 1.  Ends with a workgroup barrier.
template <int NTILE>
__device__ __forceinline__ void form_tiles(const Blk& S, const Lp& L, const Ctx& B,
                                           const double* dvec) {
    const int lane = pin(B.lane), wave = B.wave;
    const int slice = wave & 1, task = wave >> 1;
    double4v C0, C1;
    if (NTILE == 3) {
        if (task == 0) form_tile_task<0, 0, 1, 0>(S, L, dvec, lane, slice, C0, C1);
        else if (task == 1) form_tile_task<2, 0, 2, 1>(S, L, dvec, lane, slice, C0, C1);
        else if (task == 2) form_tile_task<1, 1, -1, -1>(S, L, dvec, lane, slice, C0, C1);
        else form_tile_task<2, 2, -1, -1>(S, L, dvec, lane, slice, C0, C1);
    } else {
        if (task == 0) form_tile_task<0, 0, -1, -1>(S, L, dvec, lane, slice, C0, C1);
        else if (task == 1) form_tile_task<1, 0, -1, -1>(S, L, dvec, lane, slice, C0, C1);
        else if (task == 2) form_tile_task<1, 1, -1, -1>(S, L, dvec, lane, slice, C0, C1);
    }
#define EHM4_STORE(ADD)                                                                   \
    if (NTILE == 3) {                                                                     \
        if (task == 0) { store_tile<0, 0, ADD>(S, L, lane, C0); store_tile<1, 0, ADD>(S, L, lane, C1); } \
        else if (task == 1) { store_tile<2, 0, ADD>(S, L, lane, C0); store_tile<2, 1, ADD>(S, L, lane, C1); } \
        else if (task == 2) store_tile<1, 1, ADD>(S, L, lane, C0);                        \
        else store_tile<2, 2, ADD>(S, L, lane, C0);                                       \
    } else {                                                                              \
        if (task == 0) store_tile<0, 0, ADD>(S, L, lane, C0);                             \
        else if (task == 1) store_tile<1, 0, ADD>(S, L, lane, C0);                        \
        else if (task == 2) store_tile<1, 1, ADD>(S, L, lane, C0);                        \
    }
    if (slice == 0) { EHM4_STORE(false) }
    __syncthreads();
    if (slice == 1) { EHM4_STORE(true) }
#undef EHM4_STORE
    __syncthreads();
}

// Rows and columns ncb .. srows-1 of the psi-form matrix after form_tiles: zero, except -- for the
// phase-one kinds -- row / column ncb = the column of -1's (tau):
//     M[tau][j] = - sum_i d_i W[i][j] - sum_e g_e,tau g_e,j / Delta_e ,   M[tau][tau] = sum_i d_i - ...
// with  sum_i d_i W[i][j] = gw[j]  and  - sum_i d_i = gw[ncb]  from a column product with d
// (cols_times<false>(d), called before the tiles).  Thread t < srows owns row / column t.
__device__ __forceinline__ void fix_last_columns(const Blk& S, const Lp& L, int tid) {
    const int ncb = S.ncb;
    if (tid < S.srows) {
        double val = 0.0;
        if (L.spec_mpc && tid <= ncb) {
            double sch = 0.0;
            for (int e = 0; e < S.nE; ++e)
                sch = fma(L.gE[e * GS + ncb] * L.iD()[e], L.gE[e * GS + tid], sch);
            val = -L.gw()[tid] - sch;
        }
        for (int c = ncb; c < S.srows; ++c) {
            const double v = (c == ncb) ? val : 0.0;
            L.M[c * SQ + tid] = v;
            L.M[tid * SQ + c] = v;
        }
    }
}

// psi-form -> beta-form of the weight block of the square matrix and of the eliminated block's
// rows:  M <- T' M T,  G <- G T,  T = blockdiag(I, E, 1), then the terms of the simplex rows
// -beta_q <= 0, sum beta <= 1:  diag(d_q) + d_sum 1 1'  on the weight block.  One task per entry
// that changes: (row, q) for M T, then (q, column) for T' (M T) -- wavefronts 0..3 --, (e, q) for
// G T -- wavefronts 4..7; everybody reads before anybody writes (workgroup barriers).
__device__ __forceinline__ void to_beta_form(const Blk& S, const Lp& L, int tid) {
    const int p = S.p, p0 = S.nd0, nrf = S.nrf;
    const double* Em = L.E();
    // pass 1: tasks [0, nrf p): (row, q) of M T;  [nrf p, nrf p + nE p): (e, q) of G T
    // (at most two per thread: (48 + 32) 8 = 640 <= 2 NT)
    const int n1 = nrf * p, ntask = n1 + S.nE * p;
    double val[2] = {0.0, 0.0};
    double* dst[2] = {nullptr, nullptr};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int t = tid + k * NT;
        if (t < ntask) {
            const bool ism = t < n1;
            const int tt = ism ? t : t - n1;
            const int row = tt / p, q = tt - row * p;
            double* src = ism ? (L.M + row * SQ + p0) : (L.gE + row * GS + p0);
            double v[8], ev[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rc = (r < p) ? r : 0;
                v[r] = src[rc];
                ev[r] = Em[rc * p + q];
            }
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < 8; ++r) a = (r < p) ? fma(v[r], ev[r], a) : a;
            val[k] = a;
            dst[k] = src + q;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (dst[k]) *dst[k] = val[k];
    __syncthreads();
    // pass 2: task t < nrf p -> (q, column) of T' (M T)
    double v2 = 0.0;
    const bool in2 = tid < n1;
    const int col = in2 ? tid / p : 0, q2 = in2 ? tid - col * p : 0;
    {
        double v[8], ev[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int rc = (r < p) ? r : 0;
            v[r] = L.M[(p0 + rc) * SQ + col];
            ev[r] = Em[rc * p + q2];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v2 = (r < p) ? fma(ev[r], v[r], v2) : v2;
    }
    __syncthreads();
    if (in2) L.M[(p0 + q2) * SQ + col] = v2;
    __syncthreads();
    if (L.nsx > 0 && tid < p * p) {
        const int a = tid / p, c = tid - a * p;
        L.M[(p0 + a) * SQ + p0 + c] += L.dext()[p] + ((a == c) ? L.dext()[a] : 0.0);
    }
}

// The dense extra rows in the reduced system (ehm_ipm2.h dense_prep, oracle/schur_numpy.py):
//     hX = X_E Delta^-1 ,  Xh = X_D - hX G' ,  Gh = diag(1 / d_r) + hX X_E' = L diag(delta) L' ,
//     xh = L^-1 Xh ,  dn = (l, 1/delta_1, 1/delta_2).   Wavefront 0.
__device__ __forceinline__ void dense_prep(const Blk& S, const Lp& L, int lane) {
    const int kd = L.kd;
    if (kd <= 0) return;
    const int nrf = S.nrf, nE = S.nE;
    const bool two = kd > 1;
    double g11 = frcp(L.dext()[L.nsx]);
    double g22 = two ? frcp(L.dext()[L.nsx + 1]) : 1.0;
    double g12 = 0.0;
    double x0 = 0.0, x1 = 0.0;
    if (lane < nrf) {
        x0 = L.X()[lane];
        x1 = two ? L.X()[NC + lane] : 0.0;
    }
    if (nE > 0) {
        const bool el = lane < nE;
        const double xe0 = el ? L.X()[nrf + lane] : 0.0;
        const double xe1 = (el && two) ? L.X()[NC + nrf + lane] : 0.0;
        const double id = el ? L.iD()[lane] : 0.0;
        const double h0 = xe0 * id, h1 = xe1 * id;
        if (el) {
            L.hX()[lane] = h0;
            L.hX()[MAXNE + lane] = h1;
        }
        wsync();
        if (lane < nrf)
            for (int e = 0; e < nE; ++e) {
                const double g = L.gE[e * GS + lane];
                x0 = fma(-L.hX()[e], g, x0);
                x1 = fma(-L.hX()[MAXNE + e], g, x1);
            }
        g11 += wave_sum(h0 * xe0);
        if (two) {
            g12 = wave_sum(h0 * xe1);
            g22 += wave_sum(h1 * xe1);
        }
    }
    const double i1 = frcp(g11);
    const double l = g12 * i1;
    const double i2 = two ? frcp(fma(-l, g12, g22)) : 0.0;
    x1 = fma(-l, x0, x1);
    if (lane < NF) {
        const bool on = lane < nrf && ((L.act >> lane) & 1ULL);
        L.xh()[lane] = on ? x0 : 0.0;
        L.xh()[NF + lane] = (on && two) ? x1 : 0.0;
    }
    if (lane == 0) {
        L.dn()[0] = l;
        L.dn()[1] = i1;
        L.dn()[2] = i2;
    }
    wsync();
}

// M = A' diag(d) A reduced to the factorised columns, in L.M; L.db() = its diagonal.
// Entry: dvec (= L.u0) and L.dext() visible.  Exit: workgroup barrier passed.
template <int NTILE>
__device__ __forceinline__ void form_normal_matrix(const Blk& S, const Lp& L, const Ctx& B,
                                                   const double* dvec) {
    int tid = pin(B.tid);
    if (S.nE > 0) {
        form_eliminated(S, L, dvec, tid);
        __syncthreads();
    }
    EHM4_TICK(3);
    // phase-one kinds: the products of the column of -1's with d (see fix_last_columns)
    if (L.spec_mpc) cols_times<false>(S, L, B, dvec, dvec, nullptr);
    form_tiles<NTILE>(S, L, B, dvec);
    fix_last_columns(S, L, pin(B.tid));
    __syncthreads();
    EHM4_TICK(4);
    if (L.has_beta) {
        to_beta_form(S, L, pin(B.tid));
        __syncthreads();
    }
    EHM4_TICK(5);
    if (B.wave == 0) dense_prep(S, L, pin(B.lane));
    __syncthreads();
    EHM4_TICK(6);
    // columns the LP does not have: zero rows and columns (their pivots are frozen by the guard);
    // the dense rows as rank-one terms; the diagonal for the dependent-pivot guard
    {
        tid = pin(B.tid);
        const int nrf = S.nrf;
        const double ce0 = (L.kd > 0) ? L.dn()[1] : 0.0, ce1 = (L.kd > 1) ? L.dn()[2] : 0.0;
        // thread -> row tid >> 3, columns 6 (tid & 7) .. + 5
        const int r = tid >> 3, c0 = 6 * (tid & 7);
        if (r < nrf) {
            const bool ron = (L.act >> r) & 1ULL;
            // (the reduced dense rows exist only where kd > 0: L.xh is NOT written otherwise, and
            // 0 x whatever-LDS-held is NaN as soon as that is a NaN pattern -- selects, not products)
            const double x0 = (L.kd > 0) ? ce0 * L.xh()[r] : 0.0;
            const double x1 = (L.kd > 1) ? ce1 * L.xh()[NF + r] : 0.0;
            double v[6], h0[6], h1[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                v[k] = L.M[r * SQ + c0 + k];
                h0[k] = (L.kd > 0) ? L.xh()[c0 + k] : 0.0;
                h1[k] = (L.kd > 1) ? L.xh()[NF + c0 + k] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int c = c0 + k;
                const bool on = ron && ((L.act >> c) & 1ULL);
                double w = fma(x0, h0[k], v[k]);
                w = fma(x1, h1[k], w);
                w = on ? w : 0.0;
                if (c < nrf) {
                    L.M[r * SQ + c] = w;
                    if (r == c) L.db()[r] = w;
                }
            }
        }
    }
    __syncthreads();
    EHM4_TICK(7);
}

// ---------------------------------------------------------------------------------------
// blocked factorisation  S = L D L'  and the triangular solves
// ---------------------------------------------------------------------------------------
// Panel kb (columns 16 kb ..): wavefront 0, lane j = row j with its 16 panel entries in registers.
// Step k: pivot and pivot row come out of lane (16 kb + k)'s registers by v_readlane (by
// symmetry of the Schur complement row k is what the lanes hold as column k, but the row of the
// pivot lane is just as good and needs no LDS round trip), every lane forms its multiplier and
// updates its panel entries; no lane predicates -- finished rows compute values nobody reads.
// The multipliers go back to the strictly lower triangle of L.M, the (guarded) pivots to L.pv() /
// L.rinv().  Dependent-pivot guard (LIPSOL/PCx) against the original diagonal.
template <int KB>
__device__ __forceinline__ void factor_panel(const Lp& L, int lane, int nrows) {
    constexpr int c0 = 16 * KB;
    const int jr = (lane < nrows) ? lane : (nrows - 1);
    double a[16];
    {
        const double* mrow = L.M + jr * SQ + c0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = mrow[q];
    }
    const double orig_l = L.db()[(lane < NF) ? lane : (NF - 1)];
    double piv_own = 0.0, rinv_own = 0.0;
    // the reciprocal of pivot k+1 (v_rcp_f64 + two Newton steps: five dependent double-precision
    // instructions) is started as soon as entry k+1 of the pivot row is updated, so that it runs
    // beside the rest of step k's row update instead of heading step k+1
    double piv, rinv;
    {
        piv = readlane_d(a[0], c0);
        const double orig = readlane_d(orig_l, c0);
        const bool bad = !(piv > EHM4_PIVOT_REL * orig) || !(piv > 0.0);
        piv = bad ? EHM4_PIVOT_BIG : piv;
        rinv = frcp(piv);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        double u[16];
#pragma unroll
        for (int q = k + 1; q < 16; ++q) u[q] = readlane_d(a[q], c0 + k);
        piv_own = (lane == c0 + k) ? piv : piv_own;
        rinv_own = (lane == c0 + k) ? rinv : rinv_own;
        const double l = a[k] * rinv;
        a[k] = l;
        if (k + 1 < 16) {
            a[k + 1] = fma(-l, u[k + 1], a[k + 1]);
            piv = readlane_d(a[k + 1], c0 + k + 1);
            const double orig = readlane_d(orig_l, c0 + k + 1);
            const bool bad = !(piv > EHM4_PIVOT_REL * orig) || !(piv > 0.0);
            piv = bad ? EHM4_PIVOT_BIG : piv;
            rinv = frcp(piv);
        }
#pragma unroll
        for (int q = k + 2; q < 16; ++q) a[q] = fma(-l, u[q], a[q]);
    }
    if (lane >= c0 && lane < nrows) {
        double* mrow = L.M + lane * SQ + c0;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (c0 + q < lane) {
                mrow[q] = a[q];                     // strictly lower: the multipliers
                // ... and their transpose in the upper triangle (free once the panel is in
                // registers): the backward solve then reads ROWS, like the forward one
                L.M[(c0 + q) * SQ + lane] = a[q];
            }
        if (lane < c0 + 16) {
            L.pv()[lane] = piv_own;
            L.rinv()[lane] = rinv_own;
        }
    }
}

// Trailing tile (I, J), I >= J > KB:  S_IJ -= L_I,KB D_KB L_J,KB'  -- four K-steps of the matrix
// cores, A[i][k] = -L[16 I + i][16 KB + k] d_k,  B[k][j] = L[16 J + j][16 KB + k].
template <int KB>
__device__ __forceinline__ void trailing_tile(const Lp& L, int lane, int I, int J, int nrows) {
    constexpr int c0 = 16 * KB;
    const int li = lane & 15, lk = lane >> 4;
    const int ra = 16 * I + li, rb = 16 * J + li;
    const double* pa = L.M + ((ra < nrows) ? ra : 0) * SQ + c0 + lk;
    const double* pb = L.M + ((rb < nrows) ? rb : 0) * SQ + c0 + lk;
    double4v C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + lk + 4 * r, col = 16 * J + li;
        C[r] = (row < nrows && col < nrows) ? L.M[row * SQ + col] : 0.0;
    }
    double av[4], bv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const double d = L.pv()[c0 + 4 * ks + lk];
        av[ks] = (ra < nrows) ? -pa[4 * ks] * d : 0.0;
        bv[ks] = (rb < nrows) ? pb[4 * ks] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        C = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], C, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + lk + 4 * r, col = 16 * J + li;
        // (diagonal tiles keep both triangles -- the next panel reads whole rows of its block --
        // as mirror images of the lower one)
        if (row < nrows && col < nrows && (I != J || row >= col)) {
            L.M[row * SQ + col] = C[r];
            if (I == J) L.M[col * SQ + row] = C[r];
        }
    }
}

// Entry: L.M / L.db() visible to the workgroup.  Exit: workgroup barrier passed, L.M holds the
// multipliers below the diagonal, L.pv() / L.rinv() the pivots.
template <int NTILE>
__device__ __forceinline__ void ldl_factor(const Blk& S, const Lp& L, const Ctx& B) {
    const int lane = pin(B.lane), wave = B.wave;
    const int nrows = S.srows;
    if (wave == 0) factor_panel<0>(L, lane, nrows);
    __syncthreads();
    EHM4_TICK(8);
    if (NTILE >= 2) {
        // trailing tiles of panel 0: (1,1) [, (2,1), (2,2)]
        if (wave == 0) trailing_tile<0>(L, lane, 1, 1, nrows);
        if (NTILE == 3) {
            if (wave == 1) trailing_tile<0>(L, lane, 2, 1, nrows);
            if (wave == 2) trailing_tile<0>(L, lane, 2, 2, nrows);
        }
        __syncthreads();
        EHM4_TICK(9);
        if (wave == 0) factor_panel<1>(L, lane, nrows);
        __syncthreads();
        EHM4_TICK(10);
    }
    if (NTILE == 3) {
        if (wave == 0) {
            trailing_tile<1>(L, lane, 2, 2, nrows);
            wsync();
            factor_panel<2>(L, lane, nrows);
        }
        __syncthreads();
        EHM4_TICK(11);
    }
}

// Solve (L D L') x = rhs in wavefront 0; lane j passes rhs_j (0 for lanes >= nrf), receives x_j;
// L.t[j] gets x_j too.  Forward by rows (lane j reads its row of L: stride SQ, conflict free),
// backward by columns (lane j reads column j of L below the diagonal: consecutive addresses).
// Lane k keeps y_k / x_k itself the moment it is final, so entries outside the triangle (whatever
// the matrix region holds there) only ever touch values that are already dead; the lanes of
// columns that do not exist (>= nrf) see zeros instead of matrix entries, their components stay
// exactly 0 and nothing needs a per-step test.
template <int NTILE>
__device__ __forceinline__ double ldl_solve(const Blk& S, const Lp& L, const Ctx& B, double rhs,
                                            int lane) {
    constexpr int NR = 16 * NTILE;
    const int nrf = S.nrf;
    const bool dl = lane < nrf;
    const int jr = dl ? lane : 0;
    // lane j takes entry (j, k) of L only where it belongs to the strict triangle (k < j < nrf):
    // a component that is final is then never touched again, so no per-step select is needed
    // (the first version kept 48 broadcast values alive for deferred selects and spilled them)
    const int lim = dl ? lane : 0;
    double bv = dl ? rhs : 0.0;
    const double* lrow = L.M + jr * SQ;
    const double rinv = L.rinv()[jr];
#pragma unroll
    for (int k0 = 0; k0 < NR; k0 += 16) {
        double lv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) lv[u] = lrow[k0 + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) lv[u] = (k0 + u < lim) ? lv[u] : 0.0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const double yk = readlane_d(bv, k0 + u);
            bv = fma(-lv[u], yk, bv);
        }
    }
    EHM4_TICK(23);
    bv = dl ? bv * rinv : 0.0;      // z = D^-1 y
#pragma unroll
    for (int k0 = NR - 16; k0 >= 0; k0 -= 16) {
        double lv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) lv[u] = lrow[k0 + u];      // row `lane` of L' (upper triangle)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // entries (lane, k), lane < k < nrf (nrf is wave-uniform); everything else the row
            // holds -- the diagonal, the lower triangle, columns beyond the matrix -- is discarded
            const bool take = (k0 + u < nrf) && (lane < k0 + u);
            lv[u] = take ? lv[u] : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 15; u >= 0; --u) {
            const double xk = readlane_d(bv, k0 + u);
            bv = fma(-lv[u], xk, bv);
        }
    }
    if (dl) L.t()[lane] = bv;
    wsync();
    EHM4_TICK(24);
    return bv;
}

// sum_e a[e * stride] b[e], e < n: eight terms per trip, their loads issued together, two chains
__device__ __forceinline__ double dot_strided(const double* a, int stride, const double* b, int n) {
    double s0 = 0.0, s1 = 0.0;
    int e = 0;
    for (; e + 8 <= n; e += 8) {
        double av[8], bw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = lds1(a + (e + u) * stride);
            bw[u] = b[e + u];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            s0 = fma(av[u], bw[u], s0);
            s1 = fma(av[u + 1], bw[u + 1], s1);
        }
    }
    for (; e < n; ++e) s0 = fma(a[e * stride], b[e], s0);
    return s0 + s1;
}

// The Newton system of one iteration through the reduction (ehm_ipm2.h solve_full): wavefront 0,
// lane j < nrf passes entry j of the right-hand side over the factorised columns (rD), lane
// e < nE entry nrf + e (rE).  L.t receives the solution, L.xw its psi-form, L.dn[4..5] the
// products of the dense rows with it.
template <int NTILE>
__device__ __forceinline__ void solve_full(const Blk& S, const Lp& L, const Ctx& B, double rD,
                                           double rE, int lane) {
    const int nrf = S.nrf, nE = S.nE, kd = L.kd;
    const bool two = kd > 1;
    const bool dl = lane < nrf, el = lane < nE;
    const int jl = dl ? lane : 0, je = el ? lane : 0;
    double rho0 = 0.0, rho1 = 0.0;
    double rr = dl ? rD : 0.0;
    const double xe0 = (kd > 0) ? L.X()[nrf + je] : 0.0;
    const double xe1 = two ? L.X()[NC + nrf + je] : 0.0;
    const double xd0 = (kd > 0) ? L.X()[jl] : 0.0;
    const double xd1 = two ? L.X()[NC + jl] : 0.0;
    const double ide = L.iD()[je];
    if (nE > 0) {
        const double q = el ? rE * ide : 0.0;
        if (el) L.qE()[lane] = q;
        wsync();
        rr -= dot_strided(L.gE + jl, GS, L.qE(), nE);
        if (kd > 0) {       // rho = X_E Delta^-1 r_E
            rho0 = wave_sum(xe0 * q);
            if (two) rho1 = wave_sum(xe1 * q);
        }
    }
    double l = 0.0, i1 = 0.0, i2 = 0.0;
    double xh0 = 0.0, xh1 = 0.0;
    if (kd > 0) {
        l = L.dn()[0];
        i1 = L.dn()[1];
        i2 = L.dn()[2];
        xh0 = L.xh()[jl];
        xh1 = L.xh()[NF + jl];
        rho1 = fma(-l, rho0, rho1);                     // L^-1 rho
        rr = fma(-xh0, i1 * rho0, rr);
        rr = fma(-xh1, i2 * rho1, rr);
    }
    const bool on = dl && ((L.act >> lane) & 1ULL);
    EHM4_TICK(22);
    const double xD = ldl_solve<NTILE>(S, L, B, on ? rr : 0.0, lane);
    double y0 = 0.0, y1 = 0.0;
    if (kd > 0 && nE > 0) {
        const double v0 = wave_sum(dl ? xh0 * xD : 0.0) + rho0;
        const double v1 = two ? (wave_sum(dl ? xh1 * xD : 0.0) + rho1) : 0.0;
        y1 = v1 * i2;
        y0 = fma(-l, y1, v0 * i1);
    }
    double xE = 0.0;
    if (nE > 0) {
        double acc = rE - dot_strided(L.gE + je * GS, 1, L.t(), nrf);
        acc = fma(-xe0, y0, acc);
        acc = fma(-xe1, y1, acc);
        xE = el ? acc * ide : 0.0;
        if (el) L.t()[nrf + lane] = xE;
    }
    // products of the dense rows with the step (their thread reads them in rows_times)
    if (kd > 0) {
        const double s0 = wave_sum((dl ? xd0 * xD : 0.0) + xe0 * xE);
        const double s1 = two ? wave_sum((dl ? xd1 * xD : 0.0) + xe1 * xE) : 0.0;
        if (lane == 0) {
            L.dn()[4] = s0;
            L.dn()[5] = s1;
        }
    }
    wsync();
    EHM4_TICK(25);
    to_block_columns(S, L, lane);
    EHM4_TICK(26);
}

// ---------------------------------------------------------------------------------------
// The solver.  On entry: L.X(), L.c(), L.E() and the flags are set, b in a register (row i = tid: MPC
// rows first, simplex rows at m4 .., dense rows behind them).  On exit L.xb() holds the best
// iterate.  Every thread returns the same result.
// gout (optional, LDS, p doubles; point problems): gradient of the optimal value with respect to
// the parameter, -S' lambda; NaN unless the solve converged to the tolerances.
// ---------------------------------------------------------------------------------------
template <int NTILE>
__device__ __forceinline__ IpmResult ipm_solve(const Blk& S, const Lp& L, Ctx& B, double b,
                                               bool sign_only, double step_frac,
                                               double* gout = nullptr) {
    int tid = pin(B.tid), lane = tid & 63;
    const int wave = B.wave;
    const int m = S.m, xb = S.m4;
    const int ne = L.nsx + L.kd;
    const int m_lp = m + ne;
    const int nrf = S.nrf, nE = S.nE;
    const int n_lp = nrf + nE;
    const bool valid = (tid < m) || (tid >= xb && tid < xb + ne);
    double v = valid ? b : 0.0;                 // x0 = 0  =>  b - A x0 = b
    double s = valid ? fmax(b, 1.0) : 1.0;
    double lam = valid ? 1.0 : 0.0;
    for (int k = tid; k < S.m4 + XPAD; k += NT) {       // rows nobody owns read as zero
        L.u0[k] = 0.0;
        L.u1[k] = 0.0;
    }
    // the matrix region: what lies outside the triangle that is factorised is read (and multiplied
    // by exact zeros) by the solves -- it must hold numbers
    for (int k = tid; k < (int)m_doubles(S.srows); k += NT) L.M[k] = 0.0;
    double cj = 0.0;
    if (tid < NC) {
        cj = (tid < n_lp) ? L.c()[tid] : 0.0;
        L.x()[tid] = 0.0;
        L.xb()[tid] = 0.0;
    }
    double mx[2] = {fabs(v), fabs(cj)}, sm[4] = {0.0, 0.0, 0.0, 0.0};
    block_reduce<2, 0>(B, mx, sm);
    // (reciprocals once per solve: an f64 division is a dozen dependent instructions, ~150 ticks)
    const double inv_bnorm = frcp(1.0 + mx[0]);
    const double inv_cnorm = frcp(1.0 + mx[1]);
    const double inv_m = frcp((double)m_lp);

    IpmResult res;
    res.obj = 0.0;
    res.merit = 1e300;
    res.margin = 0.0;
    res.iters = 0;
    res.status = 1;
    int stall = 0;

    EHM4_TICK_INIT();
    for (int it = 0; it <= EHM4_MAX_ITER; ++it) {
        // ---- residuals -----------------------------------------------------------------
        tid = pin(B.tid);
        lane = tid & 63;
        const double r_p = valid ? (s - v) : 0.0;       // A x + s - b
        const double rs = frcp(s);
        if (valid) {
            L.u0[tid] = lam;
            L.u1[tid] = lam * rs * r_p;
        }
        __syncthreads();
        EHM4_TICK(0);
        cols_times<true>(S, L, B, L.u0, L.u1, L.M);
        EHM4_TICK(1);
        double atl = 0.0, atdr = 0.0, xj = 0.0;
        if (tid < n_lp) {
            atl = L.g0()[tid];
            atdr = L.g1()[tid];
            xj = L.x()[tid];
        }
        const bool colon = (tid < n_lp) && (tid >= nrf || ((L.act >> tid) & 1ULL));
        const double r_d = colon ? (atl + cj) : 0.0;
        mx[0] = fmax(fabs(r_p) * inv_bnorm, fabs(r_d) * inv_cnorm);
        mx[1] = 0.0;
        sm[0] = s * lam;
        sm[1] = v * lam + xj * atl;        // b' lam = v' lam + x' (A' lam)
        sm[2] = cj * xj;
        sm[3] = 0.0;
        if (!valid) sm[0] = 0.0;
        block_reduce<1, 3>(B, mx, sm);
        const double emax = mx[0];
        const double mu = sm[0] * inv_m;
        const double dobj = -sm[1];
        const double pobj = sm[2];
        const double e_g = fabs(pobj - dobj) * frcp(1.0 + fabs(pobj));
        const double merit = fmax(emax * (1.0 / EHM4_TOL_RES), e_g * (1.0 / EHM4_TOL_GAP));
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
            if (tid < n_lp) L.xb()[tid] = xj;
        } else if (res.merit < EHM4_STALL_ZONE) {
            ++stall;
        }
        res.iters = it;
        if (merit <= 1.0) {
            res.status = 0;
            break;
        }
        if (sign_only && emax <= EHM4_SIGN_RES && pobj * dobj > 0.0) {
            const double lo = fmin(fabs(pobj), fabs(dobj));
            if (lo >= L.sign_floor && fabs(pobj - dobj) <= EHM4_SIGN_GAP * lo &&
                emax * (1.0 + fabs(pobj)) <= EHM4_SIGN_RES_REL * lo) {
                res.obj = pobj;
                res.merit = merit;
                res.margin = lo;
                res.status = 0;
                if (tid < n_lp) L.xb()[tid] = xj;
                __syncthreads();
                return res;
            }
        }
        if (stall >= 3 || it == EHM4_MAX_ITER || !(merit == merit)) break;

        // ---- normal matrix and its factorisation ----------------------------------------
        tid = pin(B.tid);
        lane = tid & 63;
        if (valid) {
            const double d = lam * rs;
            L.u0[tid] = d;
            if (tid >= xb) L.dext()[tid - xb] = d;
        }
        __syncthreads();
        EHM4_TICK(2);
        form_normal_matrix<NTILE>(S, L, B, L.u0);
        ldl_factor<NTILE>(S, L, B);
        // ---- predictor ------------------------------------------------------------------
        tid = pin(B.tid);
        lane = tid & 63;
        const double rhs_aff = colon ? (-cj - atdr) : 0.0;
        if (wave == 0) {
            // lane j: factorised column j and eliminated column j
            const double rE = (lane < nE) ? (-L.c()[nrf + lane] - L.g1()[nrf + lane]) : 0.0;
            solve_full<NTILE>(S, L, B, rhs_aff, rE, lane);
        }
        EHM4_TICK(12);
        __syncthreads();
        tid = pin(B.tid);
        double adx = rows_times(S, L, tid);
        EHM4_TICK(13);
        const double ds_a = valid ? (-r_p - adx) : 0.0;
        // dl = -(s lam + lam ds)/s = -lam - (lam/s) ds ;  -dl/lam = 1 + ds/s
        const double dl_a = valid ? (-lam - lam * rs * ds_a) : 0.0;
        mx[0] = -ds_a * rs;
        mx[1] = valid ? fma(ds_a, rs, 1.0) : 0.0;
        sm[0] = sm[1] = sm[2] = sm[3] = 0.0;
        block_reduce<2, 0>(B, mx, sm);
        double ap = (mx[0] > 1.0) ? frcp(mx[0]) : 1.0;
        double ad = (mx[1] > 1.0) ? frcp(mx[1]) : 1.0;
        mx[0] = mx[1] = 0.0;
        sm[0] = valid ? (s + ap * ds_a) * (lam + ad * dl_a) : 0.0;
        block_reduce<0, 1>(B, mx, sm);
        const double mu_aff = sm[0] * inv_m;
        const double ratio = mu_aff * frcp(mu);
        const double sigma = ratio * ratio * ratio;
        const double smu = sigma * mu;

        // ---- corrector ------------------------------------------------------------------
        tid = pin(B.tid);
        lane = tid & 63;
        const double corr = valid ? (ds_a * dl_a - smu) * rs : 0.0;
        if (valid) L.u1[tid] = corr;
        __syncthreads();
        EHM4_TICK(14);
        cols_times<false>(S, L, B, L.u1, L.u1, nullptr);
        EHM4_TICK(15);
        tid = pin(B.tid);
        lane = tid & 63;
        if (wave == 0) {
            const double rD = colon ? (rhs_aff + L.g0()[lane]) : 0.0;
            const double rE = (lane < nE)
                ? (-L.c()[nrf + lane] - L.g1()[nrf + lane] + L.g0()[nrf + lane]) : 0.0;
            solve_full<NTILE>(S, L, B, rD, rE, lane);
        }
        EHM4_TICK(16);
        __syncthreads();
        tid = pin(B.tid);
        adx = rows_times(S, L, tid);
        EHM4_TICK(17);
        const double rl = frcp(valid ? lam : 1.0);
        const double ds = valid ? (-r_p - adx) : 0.0;
        // dl = -(s lam + corr_num + lam ds)/s = -lam - corr - (lam/s) ds
        const double dl = valid ? (-lam - corr - lam * rs * ds) : 0.0;
        mx[0] = -ds * rs;
        mx[1] = -dl * rl;
        sm[0] = 0.0;
        block_reduce<2, 0>(B, mx, sm);
        ap = (mx[0] > step_frac) ? step_frac * frcp(mx[0]) : 1.0;
        ad = (mx[1] > step_frac) ? step_frac * frcp(mx[1]) : 1.0;
        if (tid < n_lp) L.x()[tid] = fma(ap, L.t()[tid], xj);
        if (valid) {
            s = fma(ap, ds, s);
            lam = fma(ad, dl, lam);
            v = fma(-ap, adx, v);
        }
        __syncthreads();
        EHM4_TICK(18);
#ifdef EHM4_PROFILE
        if (B.tid == 0) atomicAdd(&g_prof4[39], 1ULL);
#endif
    }
    __syncthreads();
    if (gout) {
        // lam is the multiplier of the LAST iterate = the returned one when the loop left through
        // the convergence test
        const bool conv = (res.status == 0) && (res.merit <= 1.0);
        for (int q0 = 0; q0 < S.p; q0 += 4) {
            double mx2[2] = {0.0, 0.0}, sm4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (q0 + c < S.p && tid < m)
                    sm4[c] = S.Wb[(size_t)(S.nd0 + q0 + c) * S.ld + tid] * lam;
            block_reduce<0, 4>(B, mx2, sm4);
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (q0 + c < S.p) gout[q0 + c] = conv ? sm4[c] : __builtin_nan("");
            }
        }
        __syncthreads();
    }
    if (res.status != 0 && res.merit <= EHM4_ACCEPT_MERIT) res.status = 0;
    res.margin = fabs(res.obj);
    return res;
}

}  // namespace ehm4
