// Wave-local primal-dual interior-point LP solver for gfx950 (CDNA4).
//
// One 64-lane wavefront owns one LP
//        min c^T x   s.t.  A x <= b ,      A is m x n,  n <= 32,  m <= 256,
// and runs Mehrotra's predictor-corrector on it (Mehrotra 1992; Nocedal & Wright
// Alg. 14.3) without touching HBM:
//   * A lives in LDS column-major (column stride `lda`, odd => the per-row and the
//     per-column access patterns below are both bank-conflict free for ds_read_b64);
//   * the m-vectors (b, s, lambda, residuals, steps) live in registers, row
//     i = lane + 64*slot;
//   * the normal matrix M = A^T diag(lambda/s) A is accumulated in 4x4 register blocks
//     (one block pair per lane, the two 32-lane halves take alternate rows), written to
//     LDS, and then eliminated with lane j holding row j of M in registers: pivots and
//     pivot rows travel through v_readlane (SGPR broadcast), so the factorisation and
//     the two triangular solves per iteration need no LDS round trips;
//   * a pivot that collapses relative to its original diagonal is frozen
//     (LIPSOL/PCx dependent-column guard), which is what keeps degenerate
//     infinity-norm-cost LPs (non-unique minimisers) well behaved.
// Quadratic costs (every MPC law of lib/mpc_library.py has a cvx.quad_form cost, :180-183,
// :515-517): with LpWork::quad set the same solver handles the convex programs
//        min c^T x + kap0 V(x)   s.t.  A x <= b  and, for the suboptimality test
//        (lib/oracle.py:89-97), two quadratic rows  kap_i V(x) + a_i^T x <= b_i ,
//        V(x) = 1/2 x^T Q x + q^T x ,
// Q (n x n) in LDS: the Newton matrix gains (kap0 + kap1 lam_1 + kap2 lam_2) Q, the two
// quadratic rows of A hold their CURRENT gradients kap_i (Q x + q) + a_i (rewritten every
// iteration), their residuals are evaluated exactly, primal and dual step share one length
// and the gap is measured by s^T lambda (oracle/ipm_numpy.py::solve_cp is the same in numpy).
// This is the arithmetic of the reference's `Problem.solve(solver=MOSEK)` call sites on
// the hot path (lib/oracle.py:131,134,166,169,203,276,305,350) for the LP instances of
// BASELINE.json; it mirrors oracle/ipm_numpy.py step for step.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define EHM_NP      32      // compile-time column capacity (KKT row lives in registers)
#define EHM_SLOTS   4       // row slots per lane  (m <= 64*EHM_SLOTS)
#define EHM_LDM     (EHM_NP + 1)

#define EHM_TOL_RES     1e-10
#define EHM_TOL_GAP     1e-10
#define EHM_MAX_ITER    40
#define EHM_STEP_FRAC   0.99
#define EHM_PIVOT_REL   1e-13
#define EHM_PIVOT_BIG   1e128
// non-improving iterations only count as a stall once the best iterate is within this factor
// of the tolerances (the merit of an infeasible-start method is not monotone early on)
#define EHM_STALL_ZONE  1e4
// a stalled solve whose best iterate is within this factor of the tolerances (residuals and
// gap <= 1e-7 relative) is accepted, like the reference accepts OPTIMAL_INACCURATE
// (lib/oracle.py:440-442): the floor of the dual residual in double precision sits at
// ~1e-10 for the worst-conditioned instances
#define EHM_ACCEPT_MERIT 1e3

namespace ehm {

// ---------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------
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
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
// broadcast lane `src` (wave-uniform, compile-time after unrolling) through SGPRs
__device__ __forceinline__ double readlane_d(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// LDS hand-off inside ONE wavefront (the workgroup is a single wave)
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// ---------------------------------------------------------------------------------------
// LP workspace carved out of dynamic LDS
// ---------------------------------------------------------------------------------------
struct LpWork {
    double* A;      // [4*ceil(n/4)] columns x lda
    double* M;      // EHM_NP x EHM_LDM
    double* vm0;    // lda
    double* vm1;    // lda
    double* c;      // EHM_NP
    double* x;      // EHM_NP
    double* xb;     // EHM_NP   best iterate
    double* t;      // EHM_NP   scratch n-vector
    // quadratic block (used when quad != 0)
    double* Q;      // EHM_NP x EHM_LDM   Hessian of V over the LP variables
    double* qv;     // EHM_NP   linear part of V
    double* a1;     // EHM_NP   linear part of quadratic row iq
    double* a2;     // EHM_NP   linear part of quadratic row iq+1
    double* gv;     // EHM_NP   scratch: grad V at the iterate
    int n, m, lda, slots, nb;
    int quad;       // 0: linear programme
    int iq;         // first of the two quadratic rows (iq, iq+1), -1: none
    double kap0, kap1, kap2;   // weights of V in the objective / the two quadratic rows
    double v0;      // constant of V (added to the reported objective)
};

__host__ __device__ inline int lp_lda(int m) { return m | 1; }
__host__ __device__ inline size_t lp_lds_doubles(int n, int m) {
    int nb = (n + 3) / 4;
    return (size_t)(4 * nb) * lp_lda(m) + 2 * (size_t)EHM_NP * EHM_LDM + 2 * (size_t)lp_lda(m) +
           8 * EHM_NP;
}

__device__ inline void lp_carve(LpWork& w, double* base, int n, int m) {
    w.n = n;
    w.m = m;
    w.lda = lp_lda(m);
    w.slots = (m + 63) >> 6;
    w.nb = (n + 3) >> 2;
    w.A = base;
    base += (size_t)(4 * w.nb) * w.lda;
    w.M = base;
    base += EHM_NP * EHM_LDM;
    w.vm0 = base;
    base += w.lda;
    w.vm1 = base;
    base += w.lda;
    w.c = base;
    base += EHM_NP;
    w.x = base;
    base += EHM_NP;
    w.xb = base;
    base += EHM_NP;
    w.t = base;
    base += EHM_NP;
    w.Q = base;
    base += EHM_NP * EHM_LDM;
    w.qv = base;
    base += EHM_NP;
    w.a1 = base;
    base += EHM_NP;
    w.a2 = base;
    base += EHM_NP;
    w.gv = base;
    w.quad = 0;
    w.iq = -1;
    w.kap0 = w.kap1 = w.kap2 = w.v0 = 0.0;
}

// zero A (all 4*nb columns) and c
__device__ inline void lp_clear(const LpWork& w, int lane) {
    const int tot = 4 * w.nb * w.lda;
    for (int k = lane; k < tot; k += 64) w.A[k] = 0.0;
    if (lane < EHM_NP) w.c[lane] = 0.0;
}

struct IpmResult {
    double obj;     // c^T x at the best iterate
    double merit;   // max(rel primal res, rel dual res)/tol_res, rel gap/tol_gap  (<=1: optimal)
    int iters;
    int status;     // 0 optimal, 1 stalled / iteration limit (best iterate kept)
};

// y_i = sum_j A[i][j] v[j] for this lane's rows; v is an n-vector in LDS
__device__ __forceinline__ void rows_times(const LpWork& w, const double* v, int lane,
                                           double (&out)[EHM_SLOTS]) {
#pragma unroll
    for (int sl = 0; sl < EHM_SLOTS; ++sl) out[sl] = 0.0;
    for (int j = 0; j < w.n; ++j) {
        const double vj = v[j];
        const double* col = w.A + (size_t)j * w.lda;
#pragma unroll
        for (int sl = 0; sl < EHM_SLOTS; ++sl) {
            if (sl < w.slots) {
                const int i = lane + 64 * sl;
                if (i < w.m) out[sl] = fma(col[i], vj, out[sl]);
            }
        }
    }
}

// (A^T u0)_j and (A^T u1)_j for j = lane & 31, u0/u1 m-vectors in LDS.
// The two 32-lane halves take alternate rows; both halves end up with the full sums.
__device__ __forceinline__ void cols_times2(const LpWork& w, const double* u0,
                                            const double* u1, int lane, double& r0,
                                            double& r1) {
    const int j = lane & 31;
    const int h = lane >> 5;
    double a0 = 0.0, a1 = 0.0;
    if (j < w.n) {
        const double* col = w.A + (size_t)j * w.lda;
        for (int i = h; i < w.m; i += 2) {
            const double a = col[i];
            a0 = fma(a, u0[i], a0);
            a1 = fma(a, u1[i], a1);
        }
    }
    r0 = a0 + __shfl_xor(a0, 32, 64);
    r1 = a1 + __shfl_xor(a1, 32, 64);
}
__device__ __forceinline__ double cols_times1(const LpWork& w, const double* u0, int lane) {
    const int j = lane & 31;
    const int h = lane >> 5;
    double a0 = 0.0;
    if (j < w.n) {
        const double* col = w.A + (size_t)j * w.lda;
        for (int i = h; i < w.m; i += 2) a0 = fma(col[i], u0[i], a0);
    }
    return a0 + __shfl_xor(a0, 32, 64);
}

// M = A^T diag(dvec) A  into w.M (full symmetric), dvec an m-vector in LDS.
__device__ inline void form_normal_matrix(const LpWork& w, const double* dvec, int lane) {
    const int nb = w.nb;
    const int T = nb * (nb + 1) / 2;
    const bool split = (T <= 32);
    const int task = split ? (lane & 31) : lane;
    const int h = split ? (lane >> 5) : 0;
    const int stride = split ? 2 : 1;
    // task -> (bj >= bk)
    int bj = 0, rem = task;
    while (rem > bj) { rem -= (bj + 1); ++bj; }
    const int bk = rem;
    const bool active = task < T;
    double acc[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
    if (active) {
        const double* cj = w.A + (size_t)(4 * bj) * w.lda;
        const double* ck = w.A + (size_t)(4 * bk) * w.lda;
        const int lda = w.lda;
        for (int i = h; i < w.m; i += stride) {
            const double d = dvec[i];
            double aj[4], ak[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                aj[q] = cj[q * lda + i] * d;
                ak[q] = ck[q * lda + i];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][r] = fma(aj[q], ak[r], acc[q][r]);
        }
    }
    if (split) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] += __shfl_xor(acc[q][r], 32, 64);
    }
    if (active && h == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jj = 4 * bj + q, kk = 4 * bk + r;
                w.M[jj * EHM_LDM + kk] = acc[q][r];
                w.M[kk * EHM_LDM + jj] = acc[q][r];
            }
    }
}

// Row-owned elimination of the n x n normal matrix: lane j holds row j.
// After it, row[q] (q<j) are the multipliers L[j][q], row[q] (q>=j) is U[j][q],
// rpiv is 1/U[j][j] (guarded).
__device__ __forceinline__ void lu_factor(double (&row)[EHM_NP], double diag0, double& rpiv,
                                          int n, int lane) {
    rpiv = 0.0;
#pragma unroll
    for (int k = 0; k < EHM_NP; ++k) {
        if (k < n) {   // wave-uniform
            double piv = readlane_d(row[k], k);
            const double orig = readlane_d(diag0, k);
            const bool bad = !(piv > EHM_PIVOT_REL * orig) || !(piv > 0.0);
            piv = bad ? EHM_PIVOT_BIG : piv;
            const double rinv = 1.0 / piv;
            if (lane == k) {
                rpiv = rinv;
                row[k] = piv;
            }
            const double l = (lane > k) ? row[k] * rinv : 0.0;
            if (lane > k) row[k] = l;
#pragma unroll
            for (int q0 = ((k + 1) & ~3); q0 < EHM_NP; q0 += 4) {
                if (q0 < n) {   // wave-uniform chunk skip
#pragma unroll
                    for (int q = q0; q < q0 + 4; ++q) {
                        if (q > k) {
                            const double ukq = readlane_d(row[q], k);
                            row[q] = fma(-l, ukq, row[q]);
                        }
                    }
                }
            }
        }
    }
}

// Solve (LU) x = rhs; lane j passes rhs_j and receives x_j.
__device__ __forceinline__ double lu_solve(const double (&row)[EHM_NP], double rpiv,
                                           double rhs, int n, int lane) {
    double bv = rhs;
#pragma unroll
    for (int k = 0; k < EHM_NP; ++k) {
        if (k < n) {
            const double yk = readlane_d(bv, k);
            bv = (lane > k) ? fma(-row[k], yk, bv) : bv;
        }
    }
    double xv = 0.0;
#pragma unroll
    for (int k = EHM_NP - 1; k >= 0; --k) {
        if (k < n) {
            const double xk = readlane_d(bv, k) * readlane_d(rpiv, k);
            xv = (lane == k) ? xk : xv;
            bv = (lane < k) ? fma(-row[k], xk, bv) : bv;
        }
    }
    return xv;
}

// ---------------------------------------------------------------------------------------
// The solver.  On entry: w.A, w.c filled, b in registers.  On exit: w.xb holds the best
// primal iterate.
// ---------------------------------------------------------------------------------------
__device__ inline IpmResult ipm_solve(const LpWork& w, const double (&b)[EHM_SLOTS],
                                      int lane) {
    const int n = w.n, m = w.m;
    bool valid[EHM_SLOTS];
    double s[EHM_SLOTS], lam[EHM_SLOTS];
    double bmax = 0.0;
#pragma unroll
    for (int sl = 0; sl < EHM_SLOTS; ++sl) {
        valid[sl] = (sl < w.slots) && (lane + 64 * sl < m);
        s[sl] = valid[sl] ? fmax(b[sl], 1.0) : 1.0;      // x0 = 0  =>  b - A x0 = b
        lam[sl] = valid[sl] ? 1.0 : 0.0;
        bmax = fmax(bmax, valid[sl] ? fabs(b[sl]) : 0.0);
    }
    const double bnorm = 1.0 + wave_max(bmax);
    const double cj = (lane < n) ? w.c[lane] : 0.0;
    const double cnorm = 1.0 + wave_max(fabs(cj));
    if (lane < EHM_NP) {
        w.x[lane] = 0.0;
        w.xb[lane] = 0.0;
    }
    wave_sync();

    IpmResult res;
    res.obj = 0.0;
    res.merit = 1e300;
    res.iters = 0;
    res.status = 1;
    int stall = 0;
    const double inv_m = 1.0 / (double)m;

    for (int it = 0; it <= EHM_MAX_ITER; ++it) {
        // ---- residuals -------------------------------------------------------------------
        double ax[EHM_SLOTS], r_p[EHM_SLOTS];
        double xQx = 0.0, qx = 0.0, gvj = 0.0;     // quadratic block: x'Qx, q'x, (Qx+q)_lane
        if (w.quad) {
            const double xj = (lane < n) ? w.x[lane] : 0.0;
            const double qj = (lane < n) ? w.qv[lane] : 0.0;
            double g = 0.0;
            if (lane < n)
                for (int k = 0; k < n; ++k) g = fma(w.Q[lane * EHM_LDM + k], w.x[k], g);
            xQx = wave_sum(xj * g);
            qx = wave_sum(xj * qj);
            gvj = g + qj;
            if (lane < EHM_NP) w.gv[lane] = gvj;
            if (w.iq >= 0 && lane < n) {           // current gradients of the quadratic rows
                w.A[(size_t)lane * w.lda + w.iq] = fma(w.kap1, gvj, w.a1[lane]);
                w.A[(size_t)lane * w.lda + w.iq + 1] = fma(w.kap2, gvj, w.a2[lane]);
            }
            wave_sync();
        }
        rows_times(w, w.x, lane, ax);
        double rpmax = 0.0, sl_sum = 0.0, bl_sum = 0.0;
#pragma unroll
        for (int sl = 0; sl < EHM_SLOTS; ++sl) {
            r_p[sl] = valid[sl] ? (ax[sl] + s[sl] - b[sl]) : 0.0;
            if (w.quad && w.iq >= 0) {
                // g_i(x) = J_i x - kap_i/2 x'Qx - b_i  (J_i = the row's current gradient)
                const int i = lane + 64 * sl;
                if (i == w.iq) r_p[sl] -= 0.5 * w.kap1 * xQx;
                if (i == w.iq + 1) r_p[sl] -= 0.5 * w.kap2 * xQx;
            }
            rpmax = fmax(rpmax, fabs(r_p[sl]));
            sl_sum = fma(s[sl], lam[sl], sl_sum);
            bl_sum = fma(b[sl], lam[sl], bl_sum);
        }
        // d = lam/s and d*r_p go to LDS for the column products / normal matrix
#pragma unroll
        for (int sl = 0; sl < EHM_SLOTS; ++sl) {
            if (valid[sl]) {
                const int i = lane + 64 * sl;
                w.vm0[i] = lam[sl];
                w.vm1[i] = lam[sl] / s[sl] * r_p[sl];
            }
        }
        wave_sync();
        double atl, atdr;
        cols_times2(w, w.vm0, w.vm1, lane, atl, atdr);
        const int jcol = lane & 31;
        const double cjj = (jcol < n) ? w.c[jcol] : 0.0;
        const double xjj = (jcol < n) ? w.x[jcol] : 0.0;
        const double gjj = (w.quad && jcol < n) ? w.kap0 * w.gv[jcol] : 0.0;
        // quadratic costs: the dual residual is measured in the metric of the Hessian's diagonal
        // (r_d_j / sqrt(1 + Q_jj)) -- a residual along a direction the cost bends by Q_jj moves the
        // optimum by r_d_j / Q_jj only, and with Q_jj ~ 1e6 (the reference's scaled input
        // weights) the unscaled measure stalls at 1e-6 on solves that are converged to 1e-11 in
        // the primal residual, the gap and the optimal value
        const double r_d = (jcol < n) ? (atl + cjj + gjj) *
            (w.quad ? rsqrt(1.0 + w.Q[jcol * EHM_LDM + jcol]) : 1.0) : 0.0;
        const double cn = w.quad ? (1.0 + wave_max(fmax(fabs(cjj), fabs(gjj)))) : cnorm;
        const double emax = wave_max(fmax(rpmax / bnorm, fabs(r_d) / cn));
        const double sl_tot = wave_sum(sl_sum);
        const double mu = sl_tot * inv_m;
        const double dobj = -wave_sum(bl_sum);
        double pobj = wave_sum((lane < 32) ? cjj * xjj : 0.0);
        if (w.quad) pobj += w.kap0 * (0.5 * xQx + qx + w.v0);
        const double e_g = w.quad ? sl_tot / (1.0 + fabs(pobj))
                                  : fabs(pobj - dobj) / (1.0 + fabs(pobj));
        double merit = fmax(emax / EHM_TOL_RES, e_g / EHM_TOL_GAP);
        // fmax / wave_max drop NaNs: a non-finite input (parameter, vertex, cost) would pass
        // as "converged".  NaN in x, s or lambda always reaches one of these two sums.
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
            if (lane < EHM_NP) w.xb[lane] = w.x[lane];
        } else if (res.merit < EHM_STALL_ZONE) {
            ++stall;
        }
        res.iters = it;
        if (merit <= 1.0) {
            res.status = 0;
            break;
        }
        if (stall >= 3 || it == EHM_MAX_ITER || !(merit == merit)) break;

        // ---- normal matrix and its factorisation ------------------------------------------
        wave_sync();   // everybody done reading vm0 (= lam)
#pragma unroll
        for (int sl = 0; sl < EHM_SLOTS; ++sl)
            if (valid[sl]) w.vm0[lane + 64 * sl] = lam[sl] / s[sl];
        wave_sync();
        form_normal_matrix(w, w.vm0, lane);
        wave_sync();
        double wq = 0.0;
        if (w.quad) {
            double l1 = 0.0, l2 = 0.0;
            if (w.iq >= 0) {
#pragma unroll
                for (int sl = 0; sl < EHM_SLOTS; ++sl) {
                    const int i = lane + 64 * sl;
                    l1 += (i == w.iq) ? lam[sl] : 0.0;
                    l2 += (i == w.iq + 1) ? lam[sl] : 0.0;
                }
                l1 = wave_sum(l1);
                l2 = wave_sum(l2);
            }
            wq = w.kap0 + w.kap1 * l1 + w.kap2 * l2;
        }
        double row[EHM_NP];
#pragma unroll
        for (int q = 0; q < EHM_NP; ++q) {
            row[q] = (lane < n && q < n) ? w.M[lane * EHM_LDM + q] : ((q == lane) ? 1.0 : 0.0);
            if (w.quad && lane < n && q < n) row[q] = fma(wq, w.Q[lane * EHM_LDM + q], row[q]);
        }
        double diag0 = (lane < n) ? w.M[lane * EHM_LDM + lane] : 1.0;
        if (w.quad && lane < n) diag0 = fma(wq, w.Q[lane * EHM_LDM + lane], diag0);
        double rpiv;
        lu_factor(row, diag0, rpiv, n, lane);

        // ---- predictor ------------------------------------------------------------------------
        // rhs_aff = -c - A^T (d r_p)
        const double rhs_aff = (lane < n) ? (-cjj - gjj - atdr) : 0.0;
        double dxj = lu_solve(row, rpiv, (lane < 32) ? rhs_aff : 0.0, n, lane);
        if (lane < EHM_NP) w.t[lane] = (lane < n) ? dxj : 0.0;
        wave_sync();
        double adx[EHM_SLOTS];
        rows_times(w, w.t, lane, adx);
        double ds_a[EHM_SLOTS], dl_a[EHM_SLOTS];
        double ap = 1e300, ad = 1e300;
#pragma unroll
        for (int sl = 0; sl < EHM_SLOTS; ++sl) {
            ds_a[sl] = valid[sl] ? (-r_p[sl] - adx[sl]) : 0.0;
            // dl = -(r_c + lam ds)/s with r_c = s lam
            dl_a[sl] = valid[sl] ? (-(s[sl] * lam[sl] + lam[sl] * ds_a[sl]) / s[sl]) : 0.0;
            if (valid[sl] && ds_a[sl] < 0.0) ap = fmin(ap, -s[sl] / ds_a[sl]);
            if (valid[sl] && dl_a[sl] < 0.0) ad = fmin(ad, -lam[sl] / dl_a[sl]);
        }
        ap = fmin(1.0, wave_min(ap));
        ad = fmin(1.0, wave_min(ad));
        if (w.quad) ap = ad = fmin(ap, ad);     // one step length: r_d couples x and lambda
        double mu_aff = 0.0;
#pragma unroll
        for (int sl = 0; sl < EHM_SLOTS; ++sl)
            if (valid[sl])
                mu_aff = fma(s[sl] + ap * ds_a[sl], lam[sl] + ad * dl_a[sl], mu_aff);
        mu_aff = wave_sum(mu_aff) * inv_m;
        const double ratio = mu_aff / mu;
        const double sigma = ratio * ratio * ratio;

        // ---- corrector -------------------------------------------------------------------------
        // rhs = rhs_aff + A^T [ (ds_a dl_a - sigma mu) / s ]
        wave_sync();
#pragma unroll
        for (int sl = 0; sl < EHM_SLOTS; ++sl)
            if (valid[sl])
                w.vm1[lane + 64 * sl] = (ds_a[sl] * dl_a[sl] - sigma * mu) / s[sl];
        wave_sync();
        const double atc = cols_times1(w, w.vm1, lane);
        const double rhs = (lane < n) ? (rhs_aff + atc) : 0.0;
        dxj = lu_solve(row, rpiv, (lane < 32) ? rhs : 0.0, n, lane);
        wave_sync();
        if (lane < EHM_NP) w.t[lane] = (lane < n) ? dxj : 0.0;
        wave_sync();
        rows_times(w, w.t, lane, adx);
        double ds[EHM_SLOTS], dl[EHM_SLOTS];
        ap = 1e300;
        ad = 1e300;
#pragma unroll
        for (int sl = 0; sl < EHM_SLOTS; ++sl) {
            ds[sl] = valid[sl] ? (-r_p[sl] - adx[sl]) : 0.0;
            const double rc = s[sl] * lam[sl] + ds_a[sl] * dl_a[sl] - sigma * mu;
            dl[sl] = valid[sl] ? (-(rc + lam[sl] * ds[sl]) / s[sl]) : 0.0;
            if (valid[sl] && ds[sl] < 0.0) ap = fmin(ap, -s[sl] / ds[sl]);
            if (valid[sl] && dl[sl] < 0.0) ad = fmin(ad, -lam[sl] / dl[sl]);
        }
        ap = fmin(1.0, EHM_STEP_FRAC * wave_min(ap));
        ad = fmin(1.0, EHM_STEP_FRAC * wave_min(ad));
        if (w.quad) ap = ad = fmin(ap, ad);
        if (lane < n) w.x[lane] = fma(ap, dxj, w.x[lane]);
#pragma unroll
        for (int sl = 0; sl < EHM_SLOTS; ++sl) {
            if (valid[sl]) {
                s[sl] = fma(ap, ds[sl], s[sl]);
                lam[sl] = fma(ad, dl[sl], lam[sl]);
            }
        }
        wave_sync();
    }
    wave_sync();
    if (res.status != 0 && res.merit <= EHM_ACCEPT_MERIT) res.status = 0;
    return res;
}

}  // namespace ehm
