// Second-generation wave-local interior-point LP solver for gfx950 (CDNA4).
//
// What changed against ehm_ipm.h (one wavefront per workgroup, whole LP matrix private):
//   * every LP of one commutation shares the SAME dense block.  With the simplex
//     parametrised by  psi = theta - R0  instead of barycentric weights, the MPC rows are
//         G z - S psi <= w + S R0
//     for every node, so the workgroup keeps ONE copy of  Wc = [G | -S | -1 | 0]  in LDS and
//     8-16 wavefronts solve their own LPs against it.  What is private to an LP is tiny:
//     the right-hand side, a handful of "extra" rows X (simplex facets in psi coordinates,
//     the two suboptimality rows, the phase-one bound) and the n x n normal matrix;
//   * the normal matrix is eliminated with lane j holding row j in registers, but the pivot
//     row travels through LDS: by symmetry of the Schur complements row k equals column k,
//     and column k is "register k of every lane" -- one ds_write per step publishes it and
//     the lanes read it back as uniform (broadcast) ds_reads.  No v_readlane / SGPR traffic;
//   * wave reductions use DPP row shifts / row broadcasts instead of ds_bpermute shuffles;
//   * one reciprocal per row and iteration (v_rcp_f64 + Newton) instead of ~25 divisions.
// The iteration itself (Mehrotra predictor-corrector, infeasible start, stall acceptance,
// dependent-pivot guard) is unchanged: the psi-LP is the beta-LP under a linear change of
// variables, and Newton's method is affine invariant (checked in oracle/ipm_numpy.py terms:
// identical iteration counts, optimal values equal to 1e-12).
//
// Round 4 -- ELIMINATED COLUMNS (oracle/schur_numpy.py).  The z-columns
// of an infinity-norm MPC law end with its epigraph variables, and every MPC row holds at most
// one of them: on that trailing range E the normal matrix of the MPC rows is DIAGONAL.  The
// solver eliminates E from the Newton system exactly (Schur complement of a positive definite
// matrix; the dense extra rows are carried along as an augmented unknown, so every term that is
// added is positive semidefinite) and factorises what is left: 15 instead of 25 columns for the
// suboptimality-test LP of config 2, 10 instead of 20 for its midpoint LP.  The iterates are the
// same to rounding (same iteration counts on every LP of the tests).  With the reduced system at
// <= 16 columns the normal matrix of the MPC rows is ONE 16x16 tile of v_mfma_f64_16x16x4_f64 and
// the Schur update three more instructions of the same accumulation.
//   LP columns, INTERNAL order (what W.c / W.x / W.xb / W.t and the extra rows X use):
//       [ z_D (nd0) | beta (p) | t or tau ]  = the nr columns that are factorised,
//       [ z_E (nE) ]                           behind them (zcol() maps a z index).
//   The workgroup's LDS image (DevProblem::Wc4) holds  [G_D | -S | -1 | 0 | aE]  and the tables
//   of E: aE[i] / eidx[i] = entry and E-column of row i, erow / eval[e][k] = the rows of column e.
//
// Reference arithmetic replaced: the `Problem.solve(solver=MOSEK)` call sites of
// lib/oracle.py:131,134,166,169,203,276,305,350.
// NO include guard: one inclusion per column capacity (see EHM2_NS below; ehm_kp.hip has two).
#include "ehm_dev.h"

#ifndef EHM_NP
#error "EHM_NP (column capacity, multiple of 4) must be defined"
#endif
#ifndef EHM_SLOTS
#error "EHM_SLOTS (row slots per lane) must be defined"
#endif

// rows / columns per software-pipelined group in the matrix loops (loads in flight per wait)
#ifndef EHM2_UNROLL
#define EHM2_UNROLL 2
#endif
// Normal matrix of the MPC rows: one tile of the matrix cores when the factorised columns with MPC
// entries are <= 16 (form_mfma), vector FMAs on 4 x 4 blocks beyond (form_blocks); 1 = matrix
// cores at every size.  Round 2 measured the two-panel case on the bench tree (25 columns then):
// identical results, decide sweep 59 ms either way, expand sweep 35 ms (MFMA) vs 32 ms (vector).
#ifndef EHM2_FORM_MFMA
#define EHM2_FORM_MFMA 0
#endif
#ifndef EHM2_LU_CHUNK
#define EHM2_LU_CHUNK 8
#endif
#define EHM2_TOL_RES      1e-10
#define EHM2_TOL_GAP      1e-10
#define EHM2_MAX_ITER     40
// fraction of the step to the boundary: 0.999 saves ~12 % of the iterations over 0.99; the few
// solves per million that stall with it are repeated (re-assembled) with 0.99, and -- seen only
// on hybrid instances, a few per ten million -- with 0.9
#define EHM2_STEP_FRAC    0.999
#define EHM2_STEP_FRAC_SAFE 0.99
#define EHM2_STEP_FRAC_LAST 0.9
#define EHM2_ATTEMPTS     3
#define EHM2_PIVOT_REL    1e-13
#define EHM2_PIVOT_BIG    1e128
#define EHM2_STALL_ZONE   1e4
#define EHM2_ACCEPT_MERIT 1e3

// every compiled instance lives in its own namespace: the sizes below differ per instance,
// and inline host functions of the same name would otherwise be merged by the linker
#define EHM2_CAT2(a, b, c) a##b##_##c
#define EHM2_CAT(a, b, c) EHM2_CAT2(a, b, c)
// EHM2_QUAD = 1 compiles the instance with the quadratic block (convex QP / QCQP, see
// ehm_ipm.h "Quadratic costs" and DESIGN.md section 3.3b): same solver, plus per wavefront the
// Hessian Q of the cost over the LP variables, and the two suboptimality rows -- already private
// "extra rows" here -- rewritten with their current gradients every iteration.  The LP
// instances (EHM2_QUAD = 0) contain none of it: their register budget is untouched.
#ifndef EHM2_QUAD
#define EHM2_QUAD 0
#endif
#if EHM2_QUAD
#define EHM2_NS EHM2_CAT(ehm2q_, EHM_NP, EHM_SLOTS)
#else
#define EHM2_NS EHM2_CAT(ehm2_, EHM_NP, EHM_SLOTS)
#endif

// -DEHM2_PROF=1 (an experimental build: EHM_BUILD_FLAGS=-DEHM2_PROF=1 EHM_BUILD_TAG=prof): the
// shader-clock cycles lane 0 of every wavefront spends in each phase of ipm_solve, summed into
// DevCounters::phase (ehm_solver_phase_ticks; tools/solver_phases.py prints the table).
#ifndef EHM2_PROF
#define EHM2_PROF 0
#endif

namespace EHM2_NS {

using namespace ehm;

#if EHM2_PROF
#define EHM2_PT(k)                                                                   \
    {                                                                                \
        const long long _t = (long long)__builtin_readcyclecounter();               \
        if (lane0 == 0) W.pf[k] += (unsigned long long)(_t - _tp);                   \
        _tp = _t;                                                                    \
    }
#define EHM2_PDUMP()                                                                 \
    {                                                                                \
        wsync();                                                                     \
        if (S.gprof && lane0 < 24)                                                   \
            atomicAdd(&S.gprof[lane0], W.pf[lane0] + (lane0 == 23 ? 1ULL : 0ULL));   \
    }
#elif defined(EHM2_ISA_MARKS)
// -DEHM2_ISA_MARKS: the same places as comments in the ISA (tools/isa_phases.py counts the
// instructions of every phase in the compiler's output; phase k ENDS at "@@PHASE k")
#define EHM2_PT(k) asm volatile("; @@PHASE " #k);
#define EHM2_PDUMP()
#else
#define EHM2_PT(k)
#define EHM2_PDUMP()
#endif

constexpr int NP = EHM_NP;
constexpr int SLOTS = EHM_SLOTS;
constexpr int MROWS = 64 * SLOTS;

// ---------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------
// LDS hand-off between the lanes of ONE wavefront: DS operations of a wave execute in issue
// order, so only the compiler has to be kept from reordering.
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
// row_shr:1,2,4,8 leave each 16-lane row's total in its lane 15; row_bcast:15 / :31 carry
// it into the following rows, so lane 63 ends with the wave total (the reduction LLVM's
// atomic optimizer emits on GFX9).
#define EHM2_DPP_REDUCE(OP, IDENT)                                       \
    v = OP(v, dpp_move<0x111, 0xf>(IDENT, v));                           \
    v = OP(v, dpp_move<0x112, 0xf>(IDENT, v));                           \
    v = OP(v, dpp_move<0x114, 0xf>(IDENT, v));                           \
    v = OP(v, dpp_move<0x118, 0xf>(IDENT, v));                           \
    v = OP(v, dpp_move<0x142, 0xa>(IDENT, v));                           \
    v = OP(v, dpp_move<0x143, 0xc>(IDENT, v));                           \
    return lane63(v);
__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
__device__ __forceinline__ double op_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double op_min(double a, double b) { return fmin(a, b); }
#ifdef EHM2_REDUCE_SHFL
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
#else
__device__ __forceinline__ double wave_sum(double v) { EHM2_DPP_REDUCE(op_add, 0.0) }
__device__ __forceinline__ double wave_max(double v) {
    EHM2_DPP_REDUCE(op_max, -__builtin_huge_val())
}
#endif
__device__ __forceinline__ double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// Keeps everything derived from x inside the current loop iteration: without it LLVM hoists
// dozens of loop-invariant per-lane LDS addresses out of the IPM iteration and spills them.
__device__ __forceinline__ int pin(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
// One ds_read_b64.  hipcc fuses neighbouring 8-byte LDS loads into ds_read2_b64 /
// ds_read2st64_b64, which the LDS serves at half the bytes per clock of ds_read_b64
// (MI355X_MICROARCH.md, LDS table); volatile loads are left alone.
typedef __attribute__((address_space(3))) const volatile double lds_cvdouble;
__device__ __forceinline__ double lds1(const double* p) {
    return *(lds_cvdouble*)p;      // explicit LDS pointer: a volatile generic load stays flat
}
// A wave-uniform double parked in an SGPR pair: there is no scalar FP unit, so uniform doubles
// (norms, tolerances, the best merit) otherwise occupy two VGPRs each for the whole solve.
// (The empty asm pins the register class: hipcc folds readfirstlane of a value it knows to be
// uniform -- literals included -- and then keeps the f64 in a hoisted VGPR pair again.)
__device__ __forceinline__ double uniform_d(double v) {
    int lo, hi;
    // explicit v_readfirstlane: the builtin is folded away for values hipcc knows to be uniform
    // (they then stay in VGPRs); nops: VGPR write -> readfirstlane, SGPR write -> VALU read
    asm("s_nop 0\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3\n\ts_nop 1"
        : "=s"(lo), "=s"(hi)
        : "v"(__double2loint(v)), "v"(__double2hiint(v)));
    return __hiloint2double(hi, lo);
}
// An f64 literal materialised with two s_mov right where it is used: as a VALU literal it would
// be a hoisted v_mov pair that lives (or spills) across the whole kernel.
__device__ __forceinline__ double const_d(double literal) {
    int lo = __double2loint(literal), hi = __double2hiint(literal);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __hiloint2double(hi, lo);
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

// ---------------------------------------------------------------------------------------
// LDS layout
// ---------------------------------------------------------------------------------------
// Shared by the workgroup: the image of DevProblem::Wc2 of one commutation in LDS.  The right-hand
// side w and the cost c of that commutation follow it in LDS (P.wc_lds = 1) or are read where they
// are, in device memory (P.wc_lds = 0): the launcher keeps them in LDS unless the 180 doubles
// they take at config 2 cost the workgroup a wavefront (the persistent frontier kernel's twelfth).
struct Shared {
    const double* Wc;   // [ncw4][lda]  G_D | -S | -1 | 0 | aE   (LDS), tables behind it
    const double* wv;   // [m]
    const double* cv;   // [n]   cost of the z-columns in their ORIGINAL order
    const double* aE;   // [m]   entry of row i in ITS eliminated column (0: the row has none)
    const double* eval; // [nE][LE]  entries of eliminated column e (0-padded)
    const int* erow;    // [nE][LE]  their rows (0-padded)
    const int* eidx;    // [m]   eliminated column of row i (0 where aE = 0)
    int n, m, p, lda;
    int nd0, nE, LE;    // z-columns [0, nd0) stay, [nd0, n) are eliminated (nE = n - nd0)
    int colS, colOne, colZero;      // image columns of -S, of -1 and of zeros
#if EHM2_PROF
    unsigned long long* gprof;      // DevCounters::phase (null: not collected)
#endif
};
__host__ __device__ inline size_t shared_doubles(const DevProblem& P) {
    return (((size_t)P.tot4 + (P.wc_lds ? P.m + P.n : 0)) + 1) & ~(size_t)1;
}
__device__ inline void carve_shared(Shared& S, double* base, const DevProblem& P) {
    S.n = P.n; S.m = P.m; S.p = P.p; S.lda = P.lda4;
    S.nd0 = P.nd0; S.nE = P.n - P.nd0; S.LE = P.LE4;
    S.colS = P.nd0; S.colOne = P.nd0 + P.p; S.colZero = P.nd0 + P.p + 1;
#if EHM2_PROF
    S.gprof = nullptr;
#endif
    S.Wc = base;
    S.aE = base + (size_t)(P.nd0 + P.p + 2) * P.lda4;
    S.eval = base + (size_t)P.ncw4 * P.lda4;
    S.erow = reinterpret_cast<const int*>(S.eval + (size_t)S.nE * S.LE);
    S.eidx = S.erow + (size_t)S.nE * S.LE;
    if (P.wc_lds) {
        S.wv = base + P.tot4;
        S.cv = S.wv + P.m;
    } else {
        S.wv = P.w;     // commutation 0; use_commutation() after a load_shared of another one
        S.cv = P.c;
    }
}
__device__ inline void use_commutation(Shared& S, const DevProblem& P, int d) {
    if (!P.wc_lds) S.wv = P.w + (size_t)d * P.m;
}
// all threads of the workgroup; caller brackets it with __syncthreads()
__device__ inline void load_shared(const DevProblem& P, int d, double* base, int tid, int nthr) {
    const int tot = P.tot4;
    const double* src = P.Wc4 + (size_t)d * tot;
    for (int k = tid; k < tot; k += nthr) base[k] = src[k];
    if (P.wc_lds) {
        const double* wd = P.w + (size_t)d * P.m;
        for (int k = tid; k < P.m; k += nthr) base[tot + k] = wd[k];
        for (int k = tid; k < P.n; k += nthr) base[tot + P.m + k] = P.c[k];
    }
}

// Private to a wavefront.
struct Wave {
    double* M;      // region A: scratch / NP x LDM normal matrix / packed U (see A_DOUBLES)
    double* X;      // [n_lp][ldx] the DENSE extra rows (extra rows nsx .. ne-1), column-major,
                    // columns in the internal order; the simplex rows 0 .. nsx-1 are not stored
    double* vm0;    // MROWS
    double* vm1;    // MROWS
    double* c;      // n_lp objective           } internal column order:
    double* x;      // n_lp iterate             }   [0, nr) the factorised columns,
    double* xb;     // n_lp best iterate        }   [nr, nr + nE) the eliminated ones
    double* t;      // n_lp scratch / solution of the last Newton system
    double* ub;     // NP pivot-row broadcast / d of the extra rows
    double* db;     // NP original diagonal, then reciprocal pivots
    double* sc;     // 256 doubles inside A: partial column products while A holds the U factor
    int n_lp, ne, ldx, ldm, xbase;
    int nr;         // factorised columns: n_lp - nE
    int nE;         // eliminated columns (Shared::nE, 0 where the instance has none)
    int n_lin;      // LP columns j < n_lin are image columns j
    int spec_col;   // image column of LP column n_lin (when n_lin < nr)
    int n_mpc;      // LP columns j < n_mpc (<= nr) have entries in the MPC rows; [n_mpc, nr) only in X
    // simplex problems: LP columns [psi0, psi0 + npsi) are the barycentric weights beta; the
    // shared block holds -S, so on the MPC rows they act through psi = E beta
    // (E[r][q] = R[q+1][r] - R[0][r], row-major p x p in LDS).  npsi = 0 for point problems.
    const double* E;
    double* yv;     // NP + 16: psi-form copy of a vector / hand-off of psi-form column products
    int psi0, npsi;
    int nsx;        // extra rows 0..nsx-1 are the simplex rows (-beta_q <= 0, sum beta <= 1):
                    // their normal-matrix terms are added analytically, rows >= nsx densely
    double sign_floor;  // a sign-only stop must establish |optimum| >= this (near-threshold
                        // routing, EHM_ROUTE_TOL): closer calls run to full accuracy
    // eliminated block (section 3.2b): filled every iteration by form_normal_matrix
    double* gE;     // [nE][GS]  (A0' D0 A0)_DE, beta-form after the transform
    double* iD;     // [nE]      1 / Delta_e
    double* qE;     // [nE]      r_E / Delta of the current right-hand side
    double* hX;     // [2][nE]   X_E Delta^-1 of the dense extra rows
    double* xh;     // [2][NP]   the dense extra rows, reduced, in L D L' form
    double* dn;     // 8: [0] l, [1] 1/delta_1, [2] 1/delta_2 of  Gh = L diag(delta) L'
#if EHM2_PROF
    unsigned long long* pf;     // 24 phase accumulators of this wavefront's current solve
#endif
#if EHM2_QUAD
    // quadratic block: objective c'x + kap0 V(x), extra rows eq / eq+1 are
    // kap_i V(x) + a_i'x <= bq_i,  V(x) = 1/2 x'Q x + qv'x (+ v0 in the reported objective)
    double* Q;      // NP x LDM
    double* qv;     // NP
    double* a1;     // NP
    double* a2;     // NP
    double* gv;     // NP  grad V at the iterate
    double* lq;     // 2   multipliers of the two quadratic rows
    int quad;       // 0: linear programme
    int eq;         // extra-row index of the first quadratic row, -1: none
    double kap0, kap1, kap2, v0, bq1, bq2;
#endif
};
constexpr int LDM = NP + 1;     // odd: row- and column-wise access of the square matrix are both
                                // bank-conflict free (the factor U is kept packed, see below)
constexpr int GS = NP + 1;      // row stride of gE: odd for the same reason
typedef double double2v __attribute__((ext_vector_type(2)));
typedef double double4v __attribute__((ext_vector_type(4)));

// Packed upper-triangular factor: row k keeps its columns (k & ~1) .. NP-1 (an even start keeps
// every row 16-byte aligned); element (k, q) lives at U[uoff(k) + q].
__host__ __device__ constexpr int u_row_start(int k) {
    // sum_{j<k} (NP - (j & ~1)) : k = 2t -> 2t*NP - 2t(t-1) ; k = 2t+1 -> (2t+1)*NP - 2t*t
    return (k & 1) ? (k * NP - 2 * (k / 2) * (k / 2)) : (k * NP - 2 * (k / 2) * (k / 2 - 1));
}
__host__ __device__ constexpr int uoff(int k) { return u_row_start(k) - (k & ~1); }
constexpr int U_SIZE = NP * (NP + 2) / 2;
// One phase-shared region per wavefront ("A"):
//   residuals      : lam / d r_p  at vm0 / vm1,  partial column products at A[0..512)
//   normal matrix  : d at vm0 (read by the row loops), K-slices at A[0..768) (form_blocks only,
//                    i.e. instances with more than 16 columns), then the square NP x LDM matrix
//   factor + solves: packed U at A[0..U_SIZE), corrector input at vm1, its partials at sc
// vm0 and sc share [U_SIZE, U_SIZE+256); vm1 follows.  Every overwrite happens after the last
// read of what it overwrites (same wavefront, program order).
constexpr size_t A_MIN = (size_t)U_SIZE + 256 + MROWS;
constexpr size_t A_KS = 512;     // K-slice scratch of form_blocks (in rounds when it needs more)
constexpr size_t A_SQ = ((size_t)NP * LDM < A_KS) ? A_KS : (size_t)NP * LDM;
constexpr size_t A_DOUBLES = (A_MIN < A_SQ) ? A_SQ : A_MIN;
// dense extra rows of an LP with ne extra rows: the simplex kinds (ne > p) lead with p + 1 simplex
// rows, which are never stored
__host__ __device__ inline int dense_rows(int ne, int p) { return (ne > p) ? ne - (p + 1) : ne; }
__host__ __device__ inline size_t wave_lp_doubles(int n_lp, int ne, int nE, int p) {
    const int kd = dense_rows(ne, p);
    const size_t ldx = kd ? ((size_t)kd | 1) : 0;
    const size_t nf = ((size_t)n_lp + 1) & ~(size_t)1;
    const size_t nEp = ((size_t)nE + 1) & ~(size_t)1;
    // X | c x xb t | ub db | yv | gE iD qE hX | xh dn
    size_t tot = A_DOUBLES + (size_t)n_lp * ldx + 4 * nf + 2 * (size_t)NP + ((size_t)NP + 16) +
                 (size_t)nE * GS + 4 * nEp + 2 * (size_t)NP + 8;
#if EHM2_QUAD
    tot += 5 * (size_t)NP + (size_t)NP * LDM + 2;
#endif
#if EHM2_PROF
    tot += 24;
#endif
    return (tot + 1) & ~(size_t)1;
}
// nsx: the first nsx extra rows are the simplex rows (0 for point problems)
__device__ inline void carve_wave(Wave& W, double* base, int n_lp, int ne, int m, int nE, int nsx) {
    W.n_lp = n_lp;
    W.nE = nE;
    W.nr = n_lp - nE;
    W.ne = ne;
    W.ldm = LDM;
    W.ldx = (ne > nsx) ? ((ne - nsx) | 1) : 0;
    W.xbase = lp_xbase(m, ne);
    // an instance compiled with more slots than the LP needs keeps the extras in ITS last slot
    if (ne > 0 && W.xbase < 64 * (SLOTS - 1)) W.xbase = 64 * (SLOTS - 1);
    const int nf = (n_lp + 1) & ~1;
    const int nEp = (nE + 1) & ~1;
    W.M = base;
    W.vm0 = base + U_SIZE;
    W.sc = base + U_SIZE;
    W.vm1 = base + U_SIZE + 256;
    base += A_DOUBLES;
    W.X = base;   base += (size_t)n_lp * W.ldx;
    W.c = base;   base += nf;
    W.x = base;   base += nf;
    W.xb = base;  base += nf;
    W.t = base;   base += nf;
    W.ub = base;  base += NP;
    W.db = base;  base += NP;
    W.yv = base;  base += NP + 16;
    W.gE = base;  base += (size_t)nE * GS;
    W.iD = base;  base += nEp;
    W.qE = base;  base += nEp;
    W.hX = base;  base += 2 * nEp;
    W.xh = base;  base += 2 * NP;
    W.dn = base;  base += 8;
#if EHM2_PROF
    W.pf = reinterpret_cast<unsigned long long*>(base);  base += 24;
#endif
#if EHM2_QUAD
    W.qv = base;  base += NP;
    W.a1 = base;  base += NP;
    W.a2 = base;  base += NP;
    W.gv = base;  base += NP;
    W.lq = base;  base += 2;
    W.Q = base;
    W.quad = 0;
    W.eq = -1;
    W.kap0 = W.kap1 = W.kap2 = W.v0 = W.bq1 = W.bq2 = 0.0;
#endif
    W.E = nullptr;
    W.psi0 = W.nr;
    W.npsi = 0;
    W.nsx = nsx;
}
// internal index of the z-column with ORIGINAL index j (see the header: eliminated columns
// sit behind the factorised ones)
__device__ __forceinline__ int zcol(const Wave& W, const Shared& S, int j) {
    return (j < S.nd0) ? j : (W.nr + j - S.nd0);
}

__device__ __forceinline__ int wc_col(const Wave& W, int j) {
    return (j < W.n_lin) ? j : W.spec_col;
}

__device__ __forceinline__ double step_fraction(int attempt) {
    return attempt == 0 ? EHM2_STEP_FRAC : (attempt == 1 ? EHM2_STEP_FRAC_SAFE : EHM2_STEP_FRAC_LAST);
}

struct IpmResult {
    double obj;
    double merit;
    double margin;   // lower bound of |optimum| when the solve stopped on its sign, else |obj|
    int iters;
    int status;   // 0 optimal / accepted, 1 stalled
};

// Sign-only termination (the suboptimality test needs sign(t*), not t*; the reference's
// bar_E is a feasibility problem, lib/oracle.py:285-309): stop as soon as the primal and the
// dual objective agree in sign, the duality gap is smaller than the smaller one of them (factor
// 0.9), and the relative residuals are below 1e-6 AND three orders below that objective -- the
// second condition is the binding one: a decision with |t*| ~ 1e-8 is only taken at residuals
// ~ 1e-11, i.e. at full accuracy.  (1e-7 / 0.5 before: 6.55 instead of 6.32 iterations per
// test on the bench tree, identical tree.)
#ifndef EHM2_SIGN_RES
#define EHM2_SIGN_RES      1e-6
#endif
#ifndef EHM2_SIGN_GAP
#define EHM2_SIGN_GAP      0.9
#endif
#ifndef EHM2_SIGN_RES_REL
#define EHM2_SIGN_RES_REL  1e-3
#endif

// Per-lane description of the rows this lane owns.
struct RowMap {
    bool valid[SLOTS];
    const double* last_base;   // last slot: address of column 0 for this lane's row
    int last_stride;           // its column stride
    const double* last_spec;   // its address in the special column
    bool last_extra;           // last slot: this lane's row is a DENSE extra row (beta-form vector)
    int last_sx;               // last slot: this lane's row is simplex row last_sx (-1: it is not)
    int lane;
};
__device__ __forceinline__ void make_rowmap(RowMap& rm, const Shared& S, const Wave& W, int lane) {
    rm.lane = lane;
    const int zero_off = S.colZero * S.lda;
#pragma unroll
    for (int sl = 0; sl < SLOTS - 1; ++sl) rm.valid[sl] = (lane + 64 * sl) < S.m;
    const int i = lane + 64 * (SLOTS - 1);
    const bool spec = W.n_lin < W.nr;
    rm.last_extra = false;
    rm.last_sx = -1;
    if (i < S.m) {
        rm.valid[SLOTS - 1] = true;
        rm.last_base = S.Wc + i;
        rm.last_stride = S.lda;
        rm.last_spec = S.Wc + (spec ? W.spec_col * S.lda : zero_off) + i;
    } else if (i >= W.xbase + W.nsx && i < W.xbase + W.ne) {
        rm.valid[SLOTS - 1] = true;
        rm.last_extra = true;
        rm.last_base = W.X + (i - W.xbase - W.nsx);
        rm.last_stride = W.ldx;
        rm.last_spec = W.X + (size_t)W.n_lin * W.ldx + (i - W.xbase - W.nsx);
    } else if (i >= W.xbase && i < W.xbase + W.nsx) {
        // simplex row: -beta_e <= 0 (e < npsi) or sum beta <= 1; nothing stored, see rows_times
        rm.valid[SLOTS - 1] = true;
        rm.last_sx = i - W.xbase;
        rm.last_base = S.Wc + zero_off;
        rm.last_stride = 0;
        rm.last_spec = S.Wc + zero_off;
    } else {
        rm.valid[SLOTS - 1] = false;
        rm.last_base = S.Wc + zero_off;
        rm.last_stride = 0;
        rm.last_spec = S.Wc + zero_off;
    }
}

// psi-form copy of an LP vector: yv = T v with T = blockdiag(I, E, I)
__device__ __forceinline__ void to_psi(const Wave& W, const double* v, int lane) {
    if (lane < NP) {
        double y = (lane < W.nr) ? v[lane] : 0.0;
        const int r = lane - W.psi0;
        if (r >= 0 && r < W.npsi) {
            y = 0.0;
            for (int q = 0; q < W.npsi; ++q) y = fma(W.E[r * W.npsi + q], v[W.psi0 + q], y);
        }
        W.yv[lane] = y;
    }
    wsync();
}

// out_i = sum_j A[i][j] v[j] for this lane's rows; v: n_lp doubles in LDS (beta-form).
// The MPC rows see psi = E beta (shared block), the extra rows see beta itself.
__device__ __forceinline__ void rows_times(const Shared& S, const Wave& W, int lane,
                                           const double* v, double (&out)[SLOTS]) {
    RowMap rm;      // rebuilt per call from the pinned lane id: no long-lived address registers
    make_rowmap(rm, S, W, lane);
    to_psi(W, v, rm.lane);
    const double* y = W.yv;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) out[sl] = 0.0;
    const double* pa = S.Wc + rm.lane;      // slots < SLOTS-1 : MPC rows, offset 64*sl
    const double* pb = rm.last_base;
    const double* pv = rm.last_extra ? v : y;       // last slot: per-lane vector
    const int lda = S.lda;
    // columns in groups: UR*SLOTS matrix loads in flight per s_waitcnt
    const int str = rm.last_stride;
    int j = 0;
    constexpr int UR = EHM2_UNROLL;
    for (; j + UR - 1 < W.n_lin; j += UR) {
        double a[UR][SLOTS], yj[UR], vl[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            yj[u] = y[j + u];
            vl[u] = lds1(pv + j + u);
#pragma unroll
            for (int sl = 0; sl < SLOTS - 1; ++sl) a[u][sl] = lds1(pa + u * lda + 64 * sl);
            a[u][SLOTS - 1] = lds1(pb + u * str);
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
#pragma unroll
            for (int sl = 0; sl < SLOTS - 1; ++sl) out[sl] = fma(a[u][sl], yj[u], out[sl]);
            out[SLOTS - 1] = fma(a[u][SLOTS - 1], vl[u], out[SLOTS - 1]);
        }
        pa += UR * lda;
        pb += UR * str;
    }
    for (; j < W.n_lin; ++j) {
        const double yj = y[j];
#pragma unroll
        for (int sl = 0; sl < SLOTS - 1; ++sl) out[sl] = fma(lds1(pa + 64 * sl), yj, out[sl]);
        out[SLOTS - 1] = fma(lds1(pb), lds1(pv + j), out[SLOTS - 1]);
        pa += lda;
        pb += str;
    }
    if (W.n_lin < W.nr) {
        const double vj = v[W.n_lin];       // the special column is not a beta column
        const double* ps = S.Wc + (size_t)W.spec_col * lda + rm.lane;
#pragma unroll
        for (int sl = 0; sl < SLOTS - 1; ++sl) out[sl] = fma(ps[64 * sl], vj, out[sl]);
        out[SLOTS - 1] = fma(*rm.last_spec, vj, out[SLOTS - 1]);
    }
    if (W.nE > 0) {
        // eliminated columns: an MPC row holds at most one of them (gather), the dense extra
        // rows hold all of them
        const double* vE = v + W.nr;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int i = rm.lane + 64 * sl;
            const int ic = (i < S.m) ? i : 0;
            const double a = (i < S.m) ? S.aE[ic] : 0.0;
            out[sl] = fma(a, vE[S.eidx[ic]], out[sl]);
        }
        if (rm.last_extra) {
            const int xe = rm.lane + 64 * (SLOTS - 1) - W.xbase - W.nsx;
            const double* xr = W.X + (size_t)W.nr * W.ldx + xe;
            double a = 0.0;
            for (int e = 0; e < W.nE; ++e) a = fma(xr[(size_t)e * W.ldx], vE[e], a);
            out[SLOTS - 1] += a;
        }
    }
    if (W.nsx > 0) {
        // the simplex rows: -beta_e, and the sum of the weights for the last one
        double sb = 0.0;
        for (int q = 0; q < W.npsi; ++q) sb += v[W.psi0 + q];
        if (rm.last_sx >= 0)
            out[SLOTS - 1] = (rm.last_sx < W.npsi) ? -v[W.psi0 + (rm.last_sx < W.npsi ? rm.last_sx : 0)]
                                                  : sb;
    }
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) out[sl] = rm.valid[sl] ? out[sl] : 0.0;
}

// Column pointers of the 4-column block cb for the MPC rows (pc) and the extra rows (px);
// columns beyond n_lp point at the zero column.
__device__ __forceinline__ void block_cols_mpc(const Shared& S, const Wave& W, int cb,
                                               const double* (&pc)[4]) {
    const double* zero = S.Wc + (size_t)S.colZero * S.lda;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = 4 * cb + q;
        pc[q] = (j < W.nr) ? (S.Wc + (size_t)wc_col(W, j) * S.lda) : zero;
    }
}
__device__ __forceinline__ void block_cols_ext(const Shared& S, const Wave& W, int cb,
                                               const double* (&px)[4]) {
    const double* zero = S.Wc + (size_t)S.colZero * S.lda;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = 4 * cb + q;
        px[q] = (j < W.nr && W.ldx > 0) ? (W.X + (size_t)j * W.ldx) : zero;
    }
}

// (A^T u0)_j and (A^T u1)_j in lane j (j < n_lp); u0/u1: m-vectors in LDS (row-indexed).
// Lanes = (column block, K-split); partial sums meet in `sc` (4*64 doubles per vector).
template <bool TWO>
__device__ __forceinline__ void cols_times(const Shared& S, const Wave& W, const double* u0,
                                           const double* u1, double* sc, int lane, double& r0,
                                           double& r1) {
    constexpr int nb = NP / 4;
    constexpr int ks = 64 / nb;
    // the extra rows first (beta-form already): u0 / u1 may be overwritten by the partials
    double ex0 = 0.0, ex1 = 0.0;
    if (lane < W.n_lp && W.ne > W.nsx) {       // the dense extra rows
        const double* xl = W.X + (size_t)lane * W.ldx;
        const int xb0 = W.xbase + W.nsx;
        for (int e = 0; e < W.ne - W.nsx; ++e) {
            const double a = xl[e];
            ex0 = fma(a, u0[xb0 + e], ex0);
            if (TWO) ex1 = fma(a, u1[xb0 + e], ex1);
        }
    }
    {   // the simplex rows: column beta_q holds -1 in row q and +1 in the row of the sum
        const int q = lane - W.psi0;
        if (W.nsx > 0 && q >= 0 && q < W.npsi) {
            ex0 += u0[W.xbase + W.npsi] - u0[W.xbase + q];
            if (TWO) ex1 += u1[W.xbase + W.npsi] - u1[W.xbase + q];
        }
    }
    // the eliminated columns: lane nr + e sums over the (few) rows of column e
    const int ecol = lane - W.nr;
    const bool elane = W.nE > 0 && ecol >= 0 && ecol < W.nE;
    if (W.nE > 0) {
        // (LE is a multiple of 4: the gathers of four rows are in flight together, not one dependent
        // LDS round trip after the other)
        const int eb = (elane ? ecol : 0) * S.LE;
        double s0 = 0.0, s1 = 0.0;
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
                w0[u] = lds1(u0 + i[u]);
                w1[u] = TWO ? lds1(u1 + i[u]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0 = fma(a[u], w0[u], s0);
                if (TWO) s1 = fma(a[u], w1[u], s1);
            }
        }
        ex0 = elane ? (ex0 + s0) : ex0;
        if (TWO) ex1 = elane ? (ex1 + s1) : ex1;
    }
    // (Round 4 tried this product on the matrix cores -- [A^T u0 | A^T u1] = W^T [u0 u1 0 ..] as one
    // 16 x 16 tile, 2 loads + 1 instruction per 4 rows: 10 % SLOWER end to end.  An FP64
    // v_mfma_f64_16x16x4 holds the SIMD's double-precision pipe for 64 cycles -- 16 vector FMAs'
    // worth -- and only 2 of the tile's 16 columns were wanted; profiles/r4/README.md.)
    const int cb = pin(lane % nb);
    const int h = pin(lane / nb);
    const bool active = h < ks;
    double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
    if (active) {
        const double* pc[4];
        block_cols_mpc(S, W, cb, pc);
        int i = h;
        constexpr int UR = EHM2_UNROLL;
        for (; i + (UR - 1) * ks < S.m; i += UR * ks) {
            double a[UR][4], v0[UR], v1[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                v0[u] = lds1(u0 + i + u * ks);
                v1[u] = TWO ? lds1(u1 + i + u * ks) : 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) a[u][q] = lds1(pc[q] + i + u * ks);
            }
#pragma unroll
            for (int u = 0; u < UR; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a0[q] = fma(a[u][q], v0[u], a0[q]);
                    if (TWO) a1[q] = fma(a[u][q], v1[u], a1[q]);
                }
        }
        for (; i < S.m; i += ks) {
            const double v0 = lds1(u0 + i);
            const double v1 = TWO ? lds1(u1 + i) : 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double a = lds1(pc[q] + i);
                a0[q] = fma(a, v0, a0[q]);
                if (TWO) a1[q] = fma(a, v1, a1[q]);
            }
        }
    }
    // scratch[vec][q][h][cb]: consecutive lanes write consecutive doubles
    constexpr int plane = ks * nb;
    wsync();
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q * plane + lane] = a0[q];
            if (TWO) sc[(4 + q) * plane + lane] = a1[q];
        }
    }
    wsync();
    r0 = 0.0;
    r1 = 0.0;
    if (lane < NP) {
        const int q = lane & 3, c = lane >> 2;
        const double* s0 = sc + q * plane + c;
#pragma unroll
        for (int hh = 0; hh < ks; ++hh) {
            r0 += s0[hh * nb];
            if (TWO) r1 += s0[4 * plane + hh * nb];
        }
    }
    wsync();
    // psi-form -> beta-form on the weight columns:  g_beta = E^T g_psi
    if (W.npsi > 0) {
        const int r = lane - W.psi0;
        if (r >= 0 && r < W.npsi) {
            W.yv[r] = r0;
            if (TWO) W.yv[W.npsi + r] = r1;
        }
        wsync();
        if (r >= 0 && r < W.npsi) {
            double g0 = 0.0, g1 = 0.0;
            for (int k = 0; k < W.npsi; ++k) {
                const double e = W.E[k * W.npsi + r];
                g0 = fma(e, W.yv[k], g0);
                if (TWO) g1 = fma(e, W.yv[W.npsi + k], g1);
            }
            r0 = g0;
            r1 = g1;
        }
        wsync();
    }
    // (lanes of eliminated columns hold no block sums: whatever they read above is discarded)
    r0 = elane ? ex0 : (r0 + ex0);
    r1 = elane ? ex1 : (r1 + ex1);
}

// M = A^T diag(dvec) A into W.M (full symmetric), dvec an m-vector in LDS.
//
// Work split: the nbA = ceil(n_mpc/4) column blocks that have entries in the MPC rows give
// TA = nbA(nbA+1)/2 block pairs; lane = (pair, K-slice h), KS = min(4, 64/TA) slices, every
// lane accumulates a 4x4 block over the rows i = h (mod KS) (MPC rows from the shared Wc,
// extra rows from the private X).  The slices meet in LDS.  Row loop unrolled by 4 so that
// the 9 loads of a row step use immediate offsets off 9 address registers.
template <int KS, int R>
__device__ __forceinline__ void form_blocks(const Shared& S, const Wave& W, const double* dvec,
                                            int lane, int nbA, int TA) {
    // lane = h * TA + task without a division by the run-time TA
    int hq = 0;
#pragma unroll
    for (int u = 1; u <= KS; ++u) hq += (lane >= u * TA) ? 1 : 0;
    const int task = pin(lane - hq * TA);
    const int h = pin(hq);
    const bool active = h < KS;
    int bj = 0, rem = task;
    while (rem > bj) { rem -= (bj + 1); ++bj; }
    const int bk = rem;
    double acc[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
    if (active) {
        const double *cj[4], *ck[4];
        block_cols_mpc(S, W, bj, cj);
        block_cols_mpc(S, W, bk, ck);
        const int m = S.m;
        int i = h;
#define EHM2_FORM_ROW(CJ, CK, DV, I)                                                  \
        {                                                                              \
            const double d = lds1((DV) + (I));                                         \
            double aj[4], ak[4];                                                       \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                            \
                aj[q] = lds1((CJ)[q] + (I)) * d;                                       \
                ak[q] = lds1((CK)[q] + (I));                                           \
            }                                                                          \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                              \
                _Pragma("unroll") for (int r = 0; r < 4; ++r)                          \
                    acc[q][r] = fma(aj[q], ak[r], acc[q][r]);                          \
        }
#if EHM2_UNROLL >= 4
        for (; i + 3 * KS < m; i += 4 * KS) {
            EHM2_FORM_ROW(cj, ck, dvec, i)
            EHM2_FORM_ROW(cj, ck, dvec, i + KS)
            EHM2_FORM_ROW(cj, ck, dvec, i + 2 * KS)
            EHM2_FORM_ROW(cj, ck, dvec, i + 3 * KS)
        }
#else
        for (; i + KS < m; i += 2 * KS) {
            EHM2_FORM_ROW(cj, ck, dvec, i)
            EHM2_FORM_ROW(cj, ck, dvec, i + KS)
        }
#endif
        for (; i < m; i += KS) EHM2_FORM_ROW(cj, ck, dvec, i)
#undef EHM2_FORM_ROW
        // The Schur update of the eliminated block in the same accumulators:
        //     M <- M - G Delta^-1 G'  =  sum_e (-g_e / Delta_e) g_e' ,
        // every eliminated column e is one more "row" of weight -1 / Delta_e whose entries are g_e
        // (form_eliminated; psi-form like the block's columns).  Columns >= NP of G do not exist.
        for (int e = h; e < W.nE; e += KS) {
            const double* ge = W.gE + e * GS;
            const double d = -W.iD[e];
            double aj[4], ak[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                aj[q] = ge[4 * bj + q] * d;
                ak[q] = ge[4 * bk + q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][r] = fma(aj[q], ak[r], acc[q][r]);
        }
    }
    // The KS slices meet in W.M: every slice parks EPR of its 16 entries per round
    // ([slice][entry][pair]: lanes write consecutively), ALL lanes add the slices up -- value v of a
    // round sits at v, v + EPR TA, v + 2 EPR TA, ... -- and the lanes of slice 0 take the sums back.
    // Rounds: as many as keep KS x EPR x TA doubles inside the region (A_KS).
    constexpr int EPR = 16 / R;     // R: chosen by the caller (form_blocks_ks)
    if (KS > 1) {
#pragma unroll
        for (int rd = 0; rd < R; ++rd) {
            wsync();
            if (active) {
                double* sc = W.M + (size_t)h * EPR * TA + task;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int en = 4 * q + r - rd * EPR;
                        if (en >= 0 && en < EPR) sc[en * TA] = acc[q][r];
                    }
            }
            wsync();
            for (int v = lane; v < EPR * TA; v += 64) {
                double sum = W.M[v];
#pragma unroll
                for (int hh = 1; hh < KS; ++hh) sum += W.M[(size_t)hh * EPR * TA + v];
                W.M[v] = sum;
            }
            wsync();
            if (h == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int en = 4 * q + r - rd * EPR;
                        if (en >= 0 && en < EPR) acc[q][r] = W.M[en * TA + task];
                    }
            }
        }
    }
    wsync();
    if (h == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jj = 4 * bj + q, kk = 4 * bk + r;
                W.M[jj * LDM + kk] = acc[q][r];
                W.M[kk * LDM + jj] = acc[q][r];
            }
    }
}

// rounds of the slice exchange: as many as keep KS x (16 / R) x TA doubles inside A_KS
template <int KS>
__device__ __forceinline__ void form_blocks_ks(const Shared& S, const Wave& W, const double* dvec,
                                               int lane, int nbA, int TA) {
    const int tot1 = KS * 16 * TA;
    if (tot1 <= (int)A_KS) form_blocks<KS, 1>(S, W, dvec, lane, nbA, TA);
    else if (tot1 <= 2 * (int)A_KS) form_blocks<KS, 2>(S, W, dvec, lane, nbA, TA);
    else form_blocks<KS, 4>(S, W, dvec, lane, nbA, TA);
}

// The same contraction on the matrix cores: M_mpc = W^T diag(d) W over the columns that have MPC
// entries, as 16x16 output tiles of v_mfma_f64_16x16x4_f64 (K = 4 rows per instruction).
// Operand layout (one f64 per lane): A[i = lane%16][k = lane/16] = d_k W[k][16I + i],
// B[k = lane/16][j = lane%16] = W[k][16J + j]; result D[(lane>>4) + 4r][lane&15], r = 0..3.
// With the eliminated columns gone the suboptimality-test LP of config 2 has 14 such columns and
// its midpoint LP 10: ONE tile, 2 LDS loads + 1 multiply + 1 matrix instruction per 4 rows, where
// the vector form (form_blocks) issues 9 loads and 20 FMAs per row and lane.  The matrix pipe is
// otherwise idle in this kernel and works beside the vector instructions of the other wavefronts.
// Two column panels (<= 32 columns) give the tiles (0,0), (1,0), (1,1); FP64 MFMA runs at the
// vector-FMA rate on MI355X and those tiles compute both triangles, so beyond 16 columns the
// vector form is used (measured equal or better there in round 2).
// The Schur update of the eliminated block rides on the same accumulators:
//     M <- M - G Delta^-1 G'    =  sum_e (-g_e / Delta_e) g_e'      (K = 4 columns e per instruction)
__device__ __forceinline__ void form_mfma(const Shared& S, const Wave& W, const double* dvec,
                                          int lane, int ncols) {
    const int li = lane & 15, lk = lane >> 4;
    const bool two = (NP > 16) && ncols > 16;
    const double* zero = S.Wc + (size_t)S.colZero * S.lda;
    const int j0 = li, j1 = 16 + li;
    const double* p0 = (j0 < ncols) ? (S.Wc + (size_t)wc_col(W, j0) * S.lda) : zero;
    const double* p1 = (j1 < ncols) ? (S.Wc + (size_t)wc_col(W, j1) * S.lda) : zero;
    double4v c00 = {0.0, 0.0, 0.0, 0.0}, c10 = {0.0, 0.0, 0.0, 0.0}, c11 = {0.0, 0.0, 0.0, 0.0};
    const int m = S.m;
    const int m4 = m & ~3;
    p0 += lk;
    p1 += lk;
    const double* pd = dvec + lk;
    int i0 = 0;
    if (!two) {
        // single tile: four K-steps per trip, their eight loads in flight together, two
        // accumulators so that consecutive matrix instructions do not wait for each other
        double4v c0b = {0.0, 0.0, 0.0, 0.0};
        for (; i0 + 16 <= m4; i0 += 16) {
            double d[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[u] = lds1(pd + i0 + 4 * u);
                w[u] = lds1(p0 + i0 + 4 * u);
            }
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[0] * d[0], w[0], c00, 0, 0, 0);
            c0b = __builtin_amdgcn_mfma_f64_16x16x4f64(w[1] * d[1], w[1], c0b, 0, 0, 0);
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[2] * d[2], w[2], c00, 0, 0, 0);
            c0b = __builtin_amdgcn_mfma_f64_16x16x4f64(w[3] * d[3], w[3], c0b, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) c00[r] += c0b[r];
    }
    for (; i0 < m4; i0 += 4) {
        const double d = lds1(pd + i0);
        const double w0 = lds1(p0 + i0);
        const double a0 = w0 * d;
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, w0, c00, 0, 0, 0);
        if (two) {      // wave-uniform
            const double w1 = lds1(p1 + i0);
            const double a1 = w1 * d;
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, w0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, w1, c11, 0, 0, 0);
        }
    }
    if (m4 < m) {       // the last 1..3 rows
        const bool in = m4 + lk < m;
        const int ic = in ? m4 : (m - 1 - lk);
        const double d = in ? lds1(pd + ic) : 0.0;
        const double w0 = lds1(p0 + ic);
        const double a0 = w0 * d;
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, w0, c00, 0, 0, 0);
        if (two) {
            const double w1 = lds1(p1 + ic);
            const double a1 = w1 * d;
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, w0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, w1, c11, 0, 0, 0);
        }
    }
    // Schur update of the eliminated block (gE, iD: this iteration's, psi-form)
    for (int e0 = 0; e0 < W.nE; e0 += 4) {
        const int e = e0 + lk;
        const bool in = e < W.nE;
        const int ec = in ? e : 0;
        const double id = W.iD[ec];
        const double g0 = (in && j0 < NP) ? W.gE[ec * GS + (j0 < NP ? j0 : 0)] : 0.0;
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-g0 * id, g0, c00, 0, 0, 0);
        if (two) {
            const double g1 = (in && j1 < NP) ? W.gE[ec * GS + (j1 < NP ? j1 : 0)] : 0.0;
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-g1 * id, g0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-g1 * id, g1, c11, 0, 0, 0);
        }
    }
    wsync();
    // D[(lane>>4) + 4r][lane&15] of every tile; rows / columns >= NP do not exist
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = lk + 4 * r, col = li;
        if (row < NP && col < NP) W.M[row * LDM + col] = c00[r];
        if (two) {
            if (16 + row < NP && col < NP) {
                W.M[(16 + row) * LDM + col] = c10[r];
                W.M[col * LDM + 16 + row] = c10[r];
            }
            if (16 + row < NP && 16 + col < NP) W.M[(16 + row) * LDM + 16 + col] = c11[r];
        }
    }
}

// G = (A0' D0 A0)_DE and Delta of the eliminated block, psi-form (before the transform):
//     g_re = sum_{i in rows(e)} d_i a_ir a_ie ,   Delta_e = sum_{i in rows(e)} d_i a_ie^2 .
// Task (r, e) = one lane; a column e has at most Shared::LE rows, listed in erow / eval.
__device__ __forceinline__ void form_eliminated(const Shared& S, const Wave& W, const double* dvec,
                                                int lane) {
    constexpr int NPG = (NP > 16) ? 32 : 16;        // tasks per eliminated column
    const double* zero = S.Wc + (size_t)S.colZero * S.lda;
    const int ntask = W.nE * NPG;
    for (int t0 = 0; t0 < ntask; t0 += 64) {
        const int t = t0 + lane;
        const int r = t & (NPG - 1);
        const int e = t / NPG;
        const bool in = t < ntask;
        const double* col = (in && r < W.n_mpc) ? (S.Wc + (size_t)wc_col(W, r) * S.lda) : zero;
        const int eb = (in ? e : 0) * S.LE;
        double g = 0.0, dl = 0.0;
        for (int k = 0; k < S.LE; k += 4) {         // LE is a multiple of 4
            int i[4];
            double ev[4], dv[4], av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                i[u] = S.erow[eb + k + u];
                ev[u] = S.eval[eb + k + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dv[u] = lds1(dvec + i[u]);
                av[u] = lds1(col + i[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double de = dv[u] * ev[u];
                g = fma(de, av[u], g);
                dl = fma(de, ev[u], dl);
            }
        }
        if (in && r < NP) W.gE[e * GS + r] = g;
        if (in && r == 0) W.iD[e] = frcp(dl);       // Delta_e > 0: d > 0, the column is not empty
    }
    wsync();
}

// Sum over the lanes 0..15 (one DPP row), returned to every lane; lanes >= 16 must pass 0 --
// 14 instructions where the wave reduction takes 22.
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_move<0x111, 0xf>(0.0, v);
    v += dpp_move<0x112, 0xf>(0.0, v);
    v += dpp_move<0x114, 0xf>(0.0, v);
    v += dpp_move<0x118, 0xf>(0.0, v);
    return readlane_d(v, 15);
}
// sum of v over the lanes [0, cnt) (0 elsewhere): one DPP row when they fit, else the wave
__device__ __forceinline__ double few_sum(double v, int cnt) {
    return (cnt <= 16) ? row16_sum(v) : wave_sum(v);
}

// psi-form -> beta-form of the weight block of the square matrix in LDS and of the eliminated
// block's rows:  M <- T^T M T,  G <- G T,  T = blockdiag(I, E, I)  (E: P x P, row-major), then the
// terms of the simplex rows  -beta_q <= 0,  sum beta <= 1  (extra rows 0..P):
// sum_e d_e a_e a_e^T = diag(d_q) + d_sum 1 1^T  on the weight block.
template <int P>
__device__ __forceinline__ void to_beta_form(const Wave& W, const double* dext, int lane) {
    const int p0 = W.psi0;
    const double* Em = W.E;      // P x P, row-major (uniform reads)
    double tmp[P];
    if (lane < NP) {            // columns: row `lane` times E
        double* mr = W.M + lane * LDM + p0;
        double v[P];
#pragma unroll
        for (int r = 0; r < P; ++r) v[r] = mr[r];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < P; ++r) a = fma(v[r], Em[r * P + q], a);
            tmp[q] = a;
        }
#pragma unroll
        for (int q = 0; q < P; ++q) mr[q] = tmp[q];
    }
    // the eliminated block's rows of the weights, same product: task = (column e, 16 at a time)
    for (int e = lane; e < W.nE; e += 64) {
        double* gr = W.gE + e * GS + p0;
        double v[P];
#pragma unroll
        for (int r = 0; r < P; ++r) v[r] = gr[r];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < P; ++r) a = fma(v[r], Em[r * P + q], a);
            gr[q] = a;
        }
    }
    wsync();
    if (lane < NP) {            // rows: E^T times column `lane`
        double* mc = W.M + p0 * LDM + lane;
        double v[P];
#pragma unroll
        for (int r = 0; r < P; ++r) v[r] = mc[r * LDM];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < P; ++r) a = fma(Em[r * P + q], v[r], a);
            tmp[q] = a;
        }
#pragma unroll
        for (int q = 0; q < P; ++q) mc[q * LDM] = tmp[q];
    }
    wsync();
    if (lane < P) {
        const double dsum = dext[P];
        double* mr = W.M + (p0 + lane) * LDM + p0;
#pragma unroll
        for (int q = 0; q < P; ++q) mr[q] += dsum + ((q == lane) ? dext[lane] : 0.0);
    }
    wsync();
}

// dext[e] = d of extra row e (the square matrix overwrites dvec before the extras are added)
__device__ __forceinline__ void form_normal_matrix(const Shared& S, const Wave& W, const double* dvec,
                                          const double* dext, int lane
#if EHM2_PROF
                                          , int lane0, long long& _tp
#endif
                                          ) {
    if (W.nE > 0) form_eliminated(S, W, dvec, lane);
    EHM2_PT(2)
    int nbA;
    // One 16 x 16 tile of the matrix cores where the columns fit (measured on the headline tree,
    // round 4: 29.0 ms per partition against 30.5 with the vector blocks below), the vector blocks
    // beyond -- three tiles compute 1024 entries for at most 528 distinct ones, and an FP64
    // v_mfma holds the SIMD's double-precision pipe like 16 vector FMAs (EHM2_FORM_MFMA = 1:
    // matrix cores at every size).
    if (EHM2_FORM_MFMA || W.n_mpc <= 16) {
        // columns >= n_mpc are zero columns for the tiles; what they leave in W.M is zero too.
        // A single panel (n_mpc <= 16) writes rows / columns 0..15 only.
        nbA = (W.n_mpc > 16 || NP <= 16) ? (NP >> 2) : 4;
        form_mfma(S, W, dvec, lane, W.n_mpc);
    } else {
        nbA = (W.n_mpc + 3) >> 2;
        const int TA = nbA * (nbA + 1) / 2;
        const int ks = 64 / TA;     // TA <= 36
        // TA = 1, 3, 6, 10, 15, 21, 28, 36 pairs for 1..8 column blocks: 64 / TA slices each
        // (the instance's column capacity bounds the block count: the rest is not compiled)
        if (ks >= 8) form_blocks_ks<8>(S, W, dvec, lane, nbA, TA);              // <= 3 blocks
        else if (NP >= 16 && ks >= 6) form_blocks_ks<6>(S, W, dvec, lane, nbA, TA);     // 4
        else if (NP >= 20 && ks >= 4) form_blocks_ks<4>(S, W, dvec, lane, nbA, TA);     // 5
        else if (NP >= 24 && ks == 3) form_blocks_ks<3>(S, W, dvec, lane, nbA, TA);     // 6
        else if (NP >= 28 && ks == 2) form_blocks_ks<2>(S, W, dvec, lane, nbA, TA);     // 7
        else if (NP >= 32) form_blocks<1, 1>(S, W, dvec, lane, nbA, TA);                // 8
    }
    // rows / columns 4*nbA .. NP-1 (no entries in the MPC rows): zero
    const int c0 = 4 * nbA;
    if (c0 < NP) {
        const int nrest = NP - c0;
        for (int k = lane; k < nrest * LDM; k += 64) W.M[c0 * LDM + k] = 0.0;
        for (int c = lane & 15; c < nrest; c += 16)
            for (int r = lane >> 4; r < c0; r += 4) W.M[r * LDM + c0 + c] = 0.0;
    }
    wsync();
    EHM2_PT(3)
    // psi-form -> beta-form of the weight block:  M <- T^T M T,  T = blockdiag(I, E, I)
    if (W.npsi > 0) {
        switch (W.npsi) {       // compile-time width: the p x p products live in registers
            case 1: to_beta_form<1>(W, dext, lane); break;
            case 2: to_beta_form<2>(W, dext, lane); break;
            case 3: to_beta_form<3>(W, dext, lane); break;
            case 4: to_beta_form<4>(W, dext, lane); break;
            case 5: to_beta_form<5>(W, dext, lane); break;
            case 6: to_beta_form<6>(W, dext, lane); break;
            case 7: to_beta_form<7>(W, dext, lane); break;
            default: to_beta_form<8>(W, dext, lane); break;
        }
    }
}

// Row-owned elimination of the n x n normal matrix (lane j holds row j in registers), pivot
// rows broadcast through LDS and NO per-step lane predicates:
//   * step k publishes column k of the current Schur complement (register k of every lane;
//     by symmetry it is row k) as row k of the packed factor U in region A (the square matrix
//     is dead by then: every lane holds its row).  After the loop A holds U and W.db[k]
//     holds 1/U[k][k] (guarded);
//   * the elimination always runs over all NP columns: columns n_lp..NP-1 are identically
//     zero (their pointers aim at the zero column), so their pivots are frozen by the guard
//     and their solution components are exactly 0 -- no data-dependent control flow at all;
//   * every lane forms its multiplier and updates its trailing row, also the lanes <= k whose
//     rows are finished: what they destroy is their copy of U (read from LDS from now on),
//     never their multipliers L[lane][q], q < lane, which stay in row[q].
// On entry W.db[j] = original diagonal (dependent-pivot guard, LIPSOL/PCx).
__device__ __forceinline__ void lu_factor(double (&row)[NP], const Wave& W, int lane) {
    double* U = W.M;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int kk = k & ~1;
        if (lane >= kk && lane < NP) U[uoff(k) + lane] = row[k];
        // the pivot itself comes from lane k's register: its reciprocal (v_rcp + two Newton steps)
        // is under way while the column travels through LDS
        double piv = readlane_d(row[k], k);
        const double orig = W.db[k];
        wsync();
        const bool bad = !(piv > const_d(EHM2_PIVOT_REL) * orig) || !(piv > 0.0);
        piv = bad ? const_d(EHM2_PIVOT_BIG) : piv;
        const double rinv = frcp(piv);
        if (lane == 0) W.db[k] = rinv;
        const double l = row[k] * rinv;
        row[k] = l;
        // pivot row: uniform 16-byte reads (packed rows start 16-byte aligned)
        if (((k + 1) & 1) && k + 1 < NP) {
            const double ukq = U[uoff(k) + k + 1];
            row[k + 1] = fma(-l, ukq, row[k + 1]);
        }
#pragma unroll
        for (int q = (k + 2) & ~1; q < NP; q += 2) {
            const double2v u = *reinterpret_cast<const double2v*>(U + uoff(k) + q);
            row[q] = fma(-l, u.x, row[q]);
            row[q + 1] = fma(-l, u.y, row[q + 1]);
            // at most EHM2_LU_CHUNK broadcast values in flight (register budget)
            if (((q - ((k + 2) & ~1)) / 2) % (EHM2_LU_CHUNK / 2) == EHM2_LU_CHUNK / 2 - 1)
                __builtin_amdgcn_sched_barrier(0);
        }
        // the trailing update of step k stays in step k: left alone, the compiler sinks each
        // FMA chain to where row[q] is next read (step q) and keeps n^2/2 broadcast values alive
#pragma unroll
        for (int q = k + 1; q < NP; ++q) asm volatile("" : "+v"(row[q]));
    }
    wsync();
}

// bv <- bv - a * s in the lanes of `mask` only (s: wave-uniform, in an SGPR pair).  The two
// triangular solves below run over the lanes of one register row; the lanes whose component is
// already final (forward) or not yet needed (backward) must not be touched, and the mask of
// step k is a compile-time constant: one scalar instruction sets EXEC, none compares lane ids.
// (s_nop: a VALU result must not be read by v_readlane in the next cycle, and the hazard
// recogniser does not look into an asm block.)
__device__ __forceinline__ void masked_fnma(double& bv, double a, double s, unsigned long long mask) {
    unsigned long long sv;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "v_fma_f64 %[bv], -%[a], %[s], %[bv]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_nop 0"
                 : [bv] "+v"(bv), [sv] "=&s"(sv)
                 : [a] "v"(a), [s] "s"(s), [m] "s"(mask)
                 : "scc");
}

// Solve (LU) x = rhs; lane j passes rhs_j and 1/U[j][j], receives x_j; W.t[0..nr) gets x too.
// Forward: L (the multipliers row[k], k < lane, zeros elsewhere) with the right-hand side in a
// register row; step k broadcasts y_k by v_readlane and every lane subtracts row[k] y_k.  Backward: U row `lane` from LDS, step k
// broadcasts x_k and updates the lanes < k.  No LDS writes, no per-step lane compares: round 3
// parked every y_k / x_k in LDS through lane 0 (9 instructions a step; 3 now).
__device__ __forceinline__ double lu_solve(const double (&row)[NP], const Wave& W, double rinv,
                                           double rhs, int lane) {
    double bv = rhs;
    // (row[k] = 0 for k >= lane, see ipm_solve: a finished component is not touched again)
#pragma unroll
    for (int k = 0; k < NP - 1; ++k) {
        const double yk = readlane_d(bv, k);
        bv = fma(-row[k], yk, bv);
    }
    const int jl = (lane < NP) ? lane : (NP - 1);
    const double* urow = W.M + uoff(jl);
#pragma unroll
    for (int k = NP - 1; k >= 1; --k) {
        const double xk = readlane_d(bv * rinv, k);
        masked_fnma(bv, urow[k], xk, (1ull << k) - 1ull);
    }
    const double x = bv * rinv;
    if (lane < W.nr) W.t[lane] = x;
    wsync();
    return x;
}

// The dense extra rows (suboptimality rows, phase-one bound: extra rows nsx .. ne-1, at most two)
// in the reduced system (header, oracle/schur_numpy.py):
//     hX = X_E Delta^-1 ,  Xh = X_D - hX G' ,  Gh = diag(1 / d_r) + hX X_E' = L diag(delta) L' ,
//     xh = L^-1 Xh ,  dn = (l, 1/delta_1, 1/delta_2)
// after which the factorised matrix is  S0 + sum_r xh_r xh_r' / delta_r.  Without eliminated
// columns this is the rank-one term d_r x_r x_r' of the dense row itself.
__device__ __forceinline__ void dense_prep(const Shared& S, const Wave& W, const double* dext,
                                           int lane) {
    const int kd = W.ne - W.nsx;
    if (kd <= 0) return;
    const int e0 = W.nsx;           // first dense row among the extra rows (dext is indexed by those)
    const int nEp = (W.nE + 1) & ~1;
    const bool two = kd > 1;
    double g11 = frcp(dext[e0]);
    double g22 = two ? frcp(dext[e0 + 1]) : 1.0;
    double g12 = 0.0;
    double x0 = 0.0, x1 = 0.0;
    if (lane < W.nr) {
        x0 = W.X[(size_t)lane * W.ldx];
        x1 = two ? W.X[(size_t)lane * W.ldx + 1] : 0.0;
    }
    if (W.nE > 0) {
        const bool el = lane < W.nE;
        const double* xr = W.X + (size_t)(W.nr + (el ? lane : 0)) * W.ldx;
        const double xe0 = el ? xr[0] : 0.0;
        const double xe1 = (el && two) ? xr[1] : 0.0;
        const double id = el ? W.iD[el ? lane : 0] : 0.0;
        const double h0 = xe0 * id, h1 = xe1 * id;
        if (el) {
            W.hX[lane] = h0;
            W.hX[nEp + lane] = h1;
        }
        wsync();
        if (lane < W.nr)
            for (int e = 0; e < W.nE; ++e) {
                const double g = W.gE[e * GS + lane];
                x0 = fma(-W.hX[e], g, x0);
                x1 = fma(-W.hX[nEp + e], g, x1);
            }
        g11 += few_sum(h0 * xe0, W.nE);
        if (two) {
            g12 = few_sum(h0 * xe1, W.nE);
            g22 += few_sum(h1 * xe1, W.nE);
        }
    }
    const double i1 = frcp(g11);
    const double l = g12 * i1;
    const double i2 = two ? frcp(fma(-l, g12, g22)) : 0.0;
    x1 = fma(-l, x0, x1);
    if (lane < NP) {
        W.xh[lane] = x0;
        W.xh[NP + lane] = two ? x1 : 0.0;
    }
    if (lane == 0) {
        W.dn[0] = l;
        W.dn[1] = i1;
        W.dn[2] = i2;
    }
    wsync();
}

// The Newton system of one iteration, solved through the reduction: lane j < n_lp passes entry j
// of the right-hand side (internal column order) and receives entry j of the solution, W.t gets
// all of it.  `row` / rinv: the factor of the reduced matrix (lu_factor).
__device__ __forceinline__ double solve_full(const double (&row)[NP], const Shared& S,
                                             const Wave& W, double rinv, double rhs, int lane) {
    const int kd = W.ne - W.nsx;
    const bool two = kd > 1;
    const int e = lane - W.nr;
    const bool el = W.nE > 0 && e >= 0 && e < W.nE;
    const int jl = (lane < NP) ? lane : (NP - 1);
    double rho0 = 0.0, rho1 = 0.0;
    double rr = (lane < W.nr) ? rhs : 0.0;
    if (W.nE > 0) {
        if (el) W.qE[e] = rhs * W.iD[e];
        wsync();
        if (lane < W.nr)
            for (int e2 = 0; e2 < W.nE; ++e2) rr = fma(-W.gE[e2 * GS + lane], W.qE[e2], rr);
        if (kd > 0) {       // rho = X_E Delta^-1 r_E
            const bool l2 = lane < W.nE;
            const double* xr = W.X + (size_t)(W.nr + (l2 ? lane : 0)) * W.ldx;
            const double q = l2 ? W.qE[l2 ? lane : 0] : 0.0;
            rho0 = few_sum(l2 ? xr[0] * q : 0.0, W.nE);
            if (two) rho1 = few_sum(l2 ? xr[1] * q : 0.0, W.nE);
        }
    }
    double l = 0.0, i1 = 0.0, i2 = 0.0;
    if (kd > 0) {
        l = W.dn[0];
        i1 = W.dn[1];
        i2 = W.dn[2];
        rho1 = fma(-l, rho0, rho1);                     // L^-1 rho
        rr = fma(-W.xh[jl], i1 * rho0, rr);
        rr = fma(-W.xh[NP + jl], i2 * rho1, rr);
    }
    const double xD = lu_solve(row, W, rinv, rr, lane);
    double y0 = 0.0, y1 = 0.0;
    if (kd > 0 && W.nE > 0) {       // (without eliminated columns nothing needs y)
        const bool lr = lane < W.nr;
        const double v0 = few_sum(lr ? W.xh[jl] * xD : 0.0, W.nr) + rho0;
        const double v1 = two ? (few_sum(lr ? W.xh[NP + jl] * xD : 0.0, W.nr) + rho1) : 0.0;
        y1 = v1 * i2;
        y0 = fma(-l, y1, v0 * i1);
    }
    if (W.nE > 0) {
        double xE = 0.0;
        if (el) {
            double acc = rhs;
            const double* ge = W.gE + e * GS;
            for (int j = 0; j < W.nr; ++j) acc = fma(-ge[j], W.t[j], acc);
            if (kd > 0) {
                const double* xr = W.X + (size_t)(W.nr + e) * W.ldx;
                acc = fma(-xr[0], y0, acc);
                if (two) acc = fma(-xr[1], y1, acc);
            }
            xE = acc * W.iD[e];
            W.t[W.nr + e] = xE;
        }
        wsync();
        return el ? xE : xD;
    }
    return xD;
}

// ---------------------------------------------------------------------------------------
// The solver.  On entry: W.X, W.c filled, b in registers (row layout of RowMap).
// On exit: W.xb holds the best primal iterate.
// ---------------------------------------------------------------------------------------
// gout (optional, LDS, S.p doubles; point problems of a linear-cost handle): the gradient of the
// optimal value with respect to the parameter, -S^T lambda = sum_i lambda_i Wc[n+q][i], NaN
// unless the solve converged to the tolerances (an accepted-inaccurate solve has no usable dual).
__device__ __forceinline__ IpmResult ipm_solve(const Shared& S, const Wave& W, const double (&b)[SLOTS],
                                      int lane0, int sign_only, double step_frac,
                                      double* gout = nullptr) {
    int lane = lane0;
    const int n = W.n_lp;
    const int m_lp = S.m + W.ne;
    RowMap rm;
    make_rowmap(rm, S, W, lane);
    // v = b - A x is carried instead of b (v <- v - alpha_p A dx): one matrix-vector product
    // per iteration less, and b need not stay in registers
    double s[SLOTS], lam[SLOTS], v[SLOTS];
    double bmax = 0.0;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        v[sl] = rm.valid[sl] ? b[sl] : 0.0;                // x0 = 0  =>  b - A x0 = b
        s[sl] = rm.valid[sl] ? fmax(b[sl], 1.0) : 1.0;
        lam[sl] = rm.valid[sl] ? 1.0 : 0.0;
        bmax = fmax(bmax, fabs(v[sl]));
    }
    const double bnorm = uniform_d(1.0 + wave_max(bmax));
    const double cj = (lane < n) ? W.c[lane] : 0.0;
    const double cnorm = uniform_d(1.0 + wave_max(fabs(cj)));
    step_frac = uniform_d(step_frac);
    if (lane < n) {
        W.x[lane] = 0.0;
        W.xb[lane] = 0.0;
    }
    // rows nobody owns must read as zero in the column products
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        if (!rm.valid[sl]) {
            W.vm0[lane + 64 * sl] = 0.0;
            W.vm1[lane + 64 * sl] = 0.0;
        }
    }
    wsync();

#if EHM2_PROF
    if (lane0 < 24) W.pf[lane0] = 0ULL;
    wsync();
    long long _tp = (long long)__builtin_readcyclecounter();
#endif
    IpmResult res;
    res.obj = 0.0;
    res.merit = const_d(1e300);
    res.iters = 0;
    res.status = 1;
    int stall = 0;
    int m_op = __builtin_amdgcn_readfirstlane(m_lp);
    asm volatile("" : "+s"(m_op));      // keeps the reciprocal below out of the kernel prologue
    const double inv_m = uniform_d(frcp((double)m_op));

    for (int it = 0; it <= EHM2_MAX_ITER; ++it) {
        lane = pin(lane0);      // per-lane addresses are re-derived every iteration (see pin)
#if EHM2_QUAD
        // ---- quadratic block: grad V, V, and the two quadratic rows at this iterate ----------
        double xQx = 0.0, qx = 0.0, gjj = 0.0;
        if (W.quad) {
            const double xj = (lane < n) ? W.x[lane] : 0.0;
            const double qj = (lane < n) ? W.qv[lane] : 0.0;
            double g = 0.0;
            if (lane < n) {
                const double* qrow = W.Q + lane * LDM;
                for (int k = 0; k < n; ++k) g = fma(qrow[k], W.x[k], g);
            }
            xQx = wave_sum(xj * g);
            qx = wave_sum(xj * qj);
            const double gvj = g + qj;
            gjj = (lane < n) ? W.kap0 * gvj : 0.0;
            if (W.eq >= 0) {
                const double a1j = (lane < n) ? W.a1[lane] : 0.0;
                const double a2j = (lane < n) ? W.a2[lane] : 0.0;
                if (lane < n) {             // current gradients kap_i (Qx+q) + a_i
                    W.X[(size_t)lane * W.ldx + W.eq - W.nsx] = fma(W.kap1, gvj, a1j);
                    W.X[(size_t)lane * W.ldx + W.eq - W.nsx + 1] = fma(W.kap2, gvj, a2j);
                }
                const double a1x = wave_sum(a1j * xj);
                const double a2x = wave_sum(a2j * xj);
                // exact b_i - g_i(x) replaces the carried value on the two quadratic rows
                const int e = lane + 64 * (SLOTS - 1) - W.xbase;
                const double Vx = fma(0.5, xQx, qx);
                if (rm.last_extra && e == W.eq) v[SLOTS - 1] = W.bq1 - fma(W.kap1, Vx, a1x);
                if (rm.last_extra && e == W.eq + 1) v[SLOTS - 1] = W.bq2 - fma(W.kap2, Vx, a2x);
            }
            wsync();
        }
#endif
        // ---- residuals -----------------------------------------------------------------
        double r_p[SLOTS], rs[SLOTS];
        double rpmax = 0.0, sl_sum = 0.0, vl_sum = 0.0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            r_p[sl] = rm.valid[sl] ? (s[sl] - v[sl]) : 0.0;       // A x + s - b
            rpmax = fmax(rpmax, fabs(r_p[sl]));
            sl_sum = fma(s[sl], lam[sl], sl_sum);
            vl_sum = fma(v[sl], lam[sl], vl_sum);
            rs[sl] = frcp(s[sl]);
        }
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            if (rm.valid[sl]) {
                W.vm0[lane + 64 * sl] = lam[sl];
                W.vm1[lane + 64 * sl] = lam[sl] * rs[sl] * r_p[sl];
            }
        }
        wsync();
        EHM2_PT(0)
        double atl, atdr;
        cols_times<true>(S, W, W.vm0, W.vm1, W.M, lane, atl, atdr);   // W.M is free here
        EHM2_PT(1)
        const double cjj = (lane < n) ? W.c[lane] : 0.0;
        const double xjj = (lane < n) ? W.x[lane] : 0.0;
#if EHM2_QUAD
        // the dual residual in the metric of the Hessian's diagonal (see ehm_ipm.h)
        const double r_d = (lane < n) ? (atl + cjj + gjj) *
            (W.quad ? rsqrt(1.0 + W.Q[lane * LDM + lane]) : 1.0) : 0.0;
        const double cn = W.quad ? (1.0 + wave_max(fmax(fabs(cjj), fabs(gjj)))) : cnorm;
        const double emax = wave_max(fmax(rpmax / bnorm, fabs(r_d) / cn));
        const double sl_tot = wave_sum(sl_sum);
        const double mu = sl_tot * inv_m;
        double pobj = wave_sum(cjj * xjj);
        double dobj = -wave_sum(vl_sum + xjj * atl);
        if (W.quad) {       // s'lam measures the gap; pobj - s'lam stands in for the dual value
            pobj += W.kap0 * (fma(0.5, xQx, qx) + W.v0);
            dobj = pobj - sl_tot;
        }
        const double e_g = fabs(pobj - dobj) / (1.0 + fabs(pobj));
#else
        const double r_d = (lane < n) ? (atl + cjj) : 0.0;
        const double emax = wave_max(fmax(rpmax / bnorm, fabs(r_d) / cnorm));
        const double mu = wave_sum(sl_sum) * inv_m;
        // b^T lam = v^T lam + x^T (A^T lam)
        const double dobj = -wave_sum(vl_sum + xjj * atl);
        const double pobj = wave_sum(cjj * xjj);
        const double e_g = fabs(pobj - dobj) / (1.0 + fabs(pobj));
#endif
        const double merit = fmax(emax / const_d(EHM2_TOL_RES), e_g / const_d(EHM2_TOL_GAP));
        // fmax / wave_max drop NaNs: a non-finite input (parameter, vertex, cost) would pass
        // as "converged".  NaN in x, s or lambda always reaches one of these two sums.
        if (!(mu == mu) || !(pobj == pobj) || fabs(pobj) > const_d(1e300)) {
            res.merit = const_d(1e300);
            res.obj = pobj;
            res.status = 1;
            res.iters = it;
            break;
        }
        if (merit < res.merit) {
            res.merit = uniform_d(merit);
            res.obj = uniform_d(pobj);
            stall = 0;
            if (lane < n) W.xb[lane] = W.x[lane];
        } else if (res.merit < const_d(EHM2_STALL_ZONE)) {
            ++stall;
        }
        res.iters = it;
        if (merit <= 1.0) {
            res.status = 0;
            break;
        }
        // sign_only: 1 = stop as soon as the sign of the optimum is certain; 2 = only when it is
        // positive (hybrid suboptimality tests: a commutation with t* = -optimum < 0 only has
        // to be known as such, one with t* >= 0 is ranked by its value)
        if (sign_only && emax <= const_d(EHM2_SIGN_RES) && pobj * dobj > 0.0 &&
            (sign_only == 1 || pobj > 0.0)) {
            const double lo = fmin(fabs(pobj), fabs(dobj));
            if (lo >= W.sign_floor && fabs(pobj - dobj) <= const_d(EHM2_SIGN_GAP) * lo &&
                emax * (1.0 + fabs(pobj)) <= const_d(EHM2_SIGN_RES_REL) * lo) {
                res.obj = pobj;
                res.merit = merit;
                res.margin = lo;
                res.status = 0;
                if (lane < n) W.xb[lane] = W.x[lane];
                wsync();
                EHM2_PT(15)
                EHM2_PDUMP()
                return res;
            }
        }
        if (stall >= 3 || it == EHM2_MAX_ITER || !(merit == merit)) break;

        EHM2_PT(15)
        // ---- normal matrix and its factorisation ----------------------------------------
        lane = pin(lane0);      // fresh per phase: addresses derived above die here
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
            if (rm.valid[sl]) W.vm0[lane + 64 * sl] = lam[sl] * rs[sl];
        wsync();
        // d of the extra rows, for the terms added after the MPC part (W.ub is free here)
        if (rm.last_extra || rm.last_sx >= 0)       // d of every extra row, simplex rows included
            W.ub[lane + 64 * (SLOTS - 1) - W.xbase] = lam[SLOTS - 1] * rs[SLOTS - 1];
#if EHM2_QUAD
        if (W.quad && W.eq >= 0 && rm.last_extra) {
            const int e = lane + 64 * (SLOTS - 1) - W.xbase;
            if (e == W.eq) W.lq[0] = lam[SLOTS - 1];
            if (e == W.eq + 1) W.lq[1] = lam[SLOTS - 1];
        }
#endif
        wsync();
#if EHM2_PROF
        form_normal_matrix(S, W, W.vm0, W.ub, lane, lane0, _tp);
#else
        form_normal_matrix(S, W, W.vm0, W.ub, lane);
#endif
        EHM2_PT(4)
        lane = pin(lane0);
        dense_prep(S, W, W.ub, lane);
        EHM2_PT(5)
        lane = pin(lane0);
        double row[NP];
        {
            // lanes >= NP carry a copy of row NP-1; nothing they compute is ever read
            const int jr = (lane < NP) ? lane : (NP - 1);
            const double* mrow = W.M + jr * LDM;
#pragma unroll
            for (int q = 0; q < NP; ++q) row[q] = mrow[q];
            // dense extra rows (suboptimality rows, phase-one bound), reduced (dense_prep): rank-one
            // terms; dg follows the diagonal (original diagonal for the dependent-pivot guard)
            double dg = mrow[jr];
            for (int r = 0; r < W.ne - W.nsx; ++r) {
                const double* xh = W.xh + r * NP;
                const double xl = xh[jr];
                const double ce = W.dn[1 + r] * xl;
                dg = fma(ce, xl, dg);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    row[q] = fma(ce, xh[q], row[q]);
                    if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // register budget
                }
            }
#if EHM2_QUAD
            if (W.quad) {       // + (kap0 + kap1 lam_1 + kap2 lam_2) Q
                double wq = W.kap0;
                if (W.eq >= 0) wq = fma(W.kap1, W.lq[0], fma(W.kap2, W.lq[1], wq));
                const int jl = (lane < NP) ? lane : (NP - 1);
                const double* qrow = W.Q + jl * LDM;
                const bool in = jl < W.n_lp;
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    row[q] = fma(wq, (in && q < W.n_lp) ? qrow[q] : 0.0, row[q]);
                dg = fma(wq, in ? qrow[jl] : 0.0, dg);
            }
#endif
            if (lane < NP) W.db[lane] = dg;
        }
        wsync();
        EHM2_PT(6)
        lu_factor(row, W, lane);
        // what the solves need of the register row are the multipliers of L (row[k], k < lane);
        // zeros elsewhere make the forward substitutions plain FMAs (lu_solve)
#pragma unroll
        for (int k = 0; k < NP; ++k) row[k] = (k < lane) ? row[k] : 0.0;
        EHM2_PT(7)
        const double rinv_l = W.db[(lane < NP) ? lane : (NP - 1)];

        // ---- predictor ------------------------------------------------------------------
        lane = pin(lane0);
#if EHM2_QUAD
        const double rhs_aff = (lane < n) ? (-cjj - gjj - atdr) : 0.0;
#else
        const double rhs_aff = (lane < n) ? (-cjj - atdr) : 0.0;
#endif
        double dxj = solve_full(row, S, W, rinv_l, rhs_aff, lane);
        EHM2_PT(8)
        double adx[SLOTS];
        rows_times(S, W, lane, W.t, adx);
        EHM2_PT(9)
        double ds_a[SLOTS], dl_a[SLOTS];
        double rho_p = 0.0, rho_d = 0.0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            ds_a[sl] = rm.valid[sl] ? (-r_p[sl] - adx[sl]) : 0.0;
            // dl = -(s lam + lam ds)/s = -lam - (lam/s) ds ;  -dl/lam = 1 + ds/s
            dl_a[sl] = rm.valid[sl] ? (-lam[sl] - lam[sl] * rs[sl] * ds_a[sl]) : 0.0;
            rho_p = fmax(rho_p, -ds_a[sl] * rs[sl]);
            rho_d = fmax(rho_d, rm.valid[sl] ? fma(ds_a[sl], rs[sl], 1.0) : 0.0);
        }
        rho_p = wave_max(rho_p);
        rho_d = wave_max(rho_d);
        double ap = (rho_p > 1.0) ? 1.0 / rho_p : 1.0;
        double ad = (rho_d > 1.0) ? 1.0 / rho_d : 1.0;
#if EHM2_QUAD
        if (W.quad) ap = ad = fmin(ap, ad);     // one step length: r_d couples x and lambda
#endif
        double mu_aff = 0.0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
            if (rm.valid[sl])
                mu_aff = fma(s[sl] + ap * ds_a[sl], lam[sl] + ad * dl_a[sl], mu_aff);
        mu_aff = wave_sum(mu_aff) * inv_m;
        const double ratio = mu_aff / mu;
        const double sigma = ratio * ratio * ratio;
        const double smu = sigma * mu;

        // ---- corrector ------------------------------------------------------------------
        lane = pin(lane0);
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            // (ds_a dl_a - sigma mu)/s : parked in LDS, read back after the corrector solve
            if (rm.valid[sl]) W.vm1[lane + 64 * sl] = (ds_a[sl] * dl_a[sl] - smu) * rs[sl];
        }
        wsync();
        double atc, dummy;
        EHM2_PT(10)
        cols_times<false>(S, W, W.vm1, W.vm1, W.sc, lane, atc, dummy);
        EHM2_PT(11)
        const double rhs = (lane < n) ? (rhs_aff + atc) : 0.0;
        dxj = solve_full(row, S, W, rinv_l, rhs, lane);
        EHM2_PT(12)
        rows_times(S, W, lane, W.t, adx);
        EHM2_PT(13)
        double ds[SLOTS], dl[SLOTS];
        rho_p = 0.0;
        rho_d = 0.0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const double corr = rm.valid[sl] ? W.vm1[lane + 64 * sl] : 0.0;
            const double rl = frcp(rm.valid[sl] ? lam[sl] : 1.0);
            ds[sl] = rm.valid[sl] ? (-r_p[sl] - adx[sl]) : 0.0;
            // dl = -(s lam + corr_num + lam ds)/s = -lam - corr - (lam/s) ds
            dl[sl] = rm.valid[sl] ? (-lam[sl] - corr - lam[sl] * rs[sl] * ds[sl]) : 0.0;
            rho_p = fmax(rho_p, -ds[sl] * rs[sl]);
            rho_d = fmax(rho_d, -dl[sl] * rl);
        }
        rho_p = wave_max(rho_p);
        rho_d = wave_max(rho_d);
        ap = (rho_p > step_frac) ? step_frac / rho_p : 1.0;
        ad = (rho_d > step_frac) ? step_frac / rho_d : 1.0;
#if EHM2_QUAD
        if (W.quad) ap = ad = fmin(ap, ad);
#endif
        if (lane < n) W.x[lane] = fma(ap, dxj, W.x[lane]);
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            if (rm.valid[sl]) {
                s[sl] = fma(ap, ds[sl], s[sl]);
                lam[sl] = fma(ad, dl[sl], lam[sl]);
                v[sl] = fma(-ap, adx[sl], v[sl]);
            }
        }
        wsync();
        EHM2_PT(14)
    }
    wsync();
    EHM2_PDUMP()
    if (gout) {
        // lam is the multiplier of the LAST iterate = the returned one when the loop left
        // through the convergence test
        const bool conv = (res.status == 0) && (res.merit <= 1.0);
        for (int q = 0; q < S.p; ++q) {
            const double* col = S.Wc + (size_t)(S.colS + q) * S.lda + lane0;
            double a = 0.0;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl)
                if (lane0 + 64 * sl < S.m) a = fma(col[64 * sl], lam[sl], a);
            a = wave_sum(a);
            if (lane0 == 0) gout[q] = conv ? a : __builtin_nan("");
        }
        wsync();
    }
    if (res.status != 0 && res.merit <= const_d(EHM2_ACCEPT_MERIT)) res.status = 0;
    res.margin = fabs(res.obj);
    return res;
}

}  // namespace EHM2_NS
