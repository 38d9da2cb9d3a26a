// Device-side problem data, LP assembly for every oracle kind, and the kernels.
// (Included once, by ehm_capi.hip.)
#pragma once

#include "ehm_dev.h"
#include "ehm_ipm.h"

namespace ehm {

// ---------------------------------------------------------------------------------------
// quadratic block of the cost (DevProblem::Hq set): V over the LP variables
// ---------------------------------------------------------------------------------------
// Fixed parameter: Q = H, q = c + f0 + F theta, v0 = 1/2 theta'C theta + c1'theta + c0; the
// objective is V itself (kap0 = 1, linear objective cleared).
__device__ inline void quad_point(LpWork& w, const DevProblem& P, int d, const double* theta,
                                  int lane) {
    const int n = P.n, p = P.p;
    const double* H = P.Hq + (size_t)d * n * n;
    for (int k = lane; k < n * n; k += 64) w.Q[(k / n) * EHM_LDM + (k % n)] = H[k];
    if (lane < n) {
        const double* F = P.Fq + (size_t)d * p * n;
        double v = P.c[lane] + P.f0q[(size_t)d * n + lane];
        for (int q = 0; q < p; ++q) v = fma(F[(size_t)q * n + lane], theta[q], v);
        w.qv[lane] = v;
        w.c[lane] = 0.0;
    }
    const double* C = P.Cq + (size_t)d * p * p;
    double v0 = P.c0q[d];
    for (int r = 0; r < p; ++r) {
        double cr = P.c1q[(size_t)d * p + r];
        for (int q = 0; q < p; ++q) cr = fma(0.5 * C[r * p + q], theta[q], cr);
        v0 = fma(cr, theta[r], v0);
    }
    w.quad = 1;
    w.iq = -1;
    w.kap0 = 1.0;
    w.kap1 = w.kap2 = 0.0;
    w.v0 = v0;
}

// Over a simplex, theta = R0 + E beta, E[r][q] = R[q+1][r] - R0[r]: over (z, beta[, t])
//   Q = [H, F E; E'F', E'C E],  q = [c + f0 + F R0; E'(C R0 + c1)],  v0 as above at R0.
// slack = true: objective -t (set by the caller), rows iq, iq+1 are
//   kap_i V - sum_q beta_q (Vbar_{q+1}-Vbar_0) + t <= b_i - kap_i v0 .
__device__ inline void quad_simplex(LpWork& w, const DevProblem& P, int d, const double* R,
                                    const double* Vbar, bool slack, int lane) {
    const int n = P.n, p = P.p;
    const int nl = n + p + (slack ? 1 : 0);
    const double* H = P.Hq + (size_t)d * n * n;
    const double* F = P.Fq + (size_t)d * p * n;
    const double* C = P.Cq + (size_t)d * p * p;
    for (int k = lane; k < nl * EHM_LDM; k += 64) w.Q[k] = 0.0;
    wave_sync();
    for (int k = lane; k < n * n; k += 64) w.Q[(k / n) * EHM_LDM + (k % n)] = H[k];
    if (lane < n) {
        double v = P.c[lane] + P.f0q[(size_t)d * n + lane];
        for (int r = 0; r < p; ++r) v = fma(F[(size_t)r * n + lane], R[r], v);
        w.qv[lane] = v;
        for (int q = 0; q < p; ++q) {
            double acc = 0.0;
            for (int r = 0; r < p; ++r)
                acc = fma(F[(size_t)r * n + lane], R[(q + 1) * p + r] - R[r], acc);
            w.Q[lane * EHM_LDM + n + q] = acc;
            w.Q[(n + q) * EHM_LDM + lane] = acc;
        }
    }
    if (lane < p * p) {                       // E'C E, one entry per lane (p <= 8)
        const int q = lane / p, q2 = lane % p;
        double acc = 0.0;
        for (int r = 0; r < p; ++r) {
            double ce = 0.0;
            for (int r2 = 0; r2 < p; ++r2)
                ce = fma(C[r * p + r2], R[(q2 + 1) * p + r2] - R[r2], ce);
            acc = fma(R[(q + 1) * p + r] - R[r], ce, acc);
        }
        w.Q[(n + q) * EHM_LDM + n + q2] = acc;
    }
    if (lane < p) {                           // E'(C R0 + c1)
        double acc = 0.0;
        for (int r = 0; r < p; ++r) {
            double cr = P.c1q[(size_t)d * p + r];
            for (int r2 = 0; r2 < p; ++r2) cr = fma(C[r * p + r2], R[r2], cr);
            acc = fma(R[(lane + 1) * p + r] - R[r], cr, acc);
        }
        w.qv[n + lane] = acc;
    }
    double v0 = P.c0q[d];
    for (int r = 0; r < p; ++r) {
        double cr = P.c1q[(size_t)d * p + r];
        for (int q = 0; q < p; ++q) cr = fma(0.5 * C[r * p + q], R[q], cr);
        v0 = fma(cr, R[r], v0);
    }
    w.quad = 1;
    w.v0 = v0;
    if (slack) {
        if (lane < EHM_NP) {
            double a = 0.0;
            if (lane >= n && lane < n + p) a = -(Vbar[lane - n + 1] - Vbar[0]);
            if (lane == n + p) a = 1.0;
            w.a1[lane] = a;
            w.a2[lane] = a;
        }
        if (lane == 0) w.qv[n + p] = 0.0;
        w.iq = P.m + p + 1;
        w.kap0 = 0.0;
        w.kap1 = 1.0;
        w.kap2 = 1.0 + P.eps_r;
    } else {
        if (lane < nl) w.c[lane] = 0.0;
        w.iq = -1;
        w.kap0 = 1.0;
        w.kap1 = w.kap2 = 0.0;
    }
}

// ---------------------------------------------------------------------------------------
// LP assembly
// ---------------------------------------------------------------------------------------
// P_theta_delta at one parameter value (lib/oracle.py:141-173), or its phase-one form
//   min tau  s.t.  G z - tau <= h,  tau >= -1      (feasible  <=>  tau* <= 0).
// theta: p doubles readable by every lane (LDS or global).
__device__ inline void assemble_point(LpWork& w, double* smem, const DevProblem& P, int d,
                                      const double* theta, bool feas,
                                      double (&b)[EHM_SLOTS], int lane) {
    const int n = P.n, m = P.m;
    lp_carve(w, smem, n + (feas ? 1 : 0), m + (feas ? 1 : 0));
    lp_clear(w, lane);
    wave_sync();
    const double* Gt = P.Gt + (size_t)d * n * m;
    for (int j = 0; j < n; ++j) {
        const double* src = Gt + (size_t)j * m;
        double* dst = w.A + (size_t)j * w.lda;
        for (int i = lane; i < m; i += 64) dst[i] = src[i];
    }
    if (feas) {
        double* dst = w.A + (size_t)n * w.lda;
        for (int i = lane; i <= m; i += 64) dst[i] = -1.0;
        if (lane == 0) w.c[n] = 1.0;
    } else if (lane < n) {
        w.c[lane] = P.c[lane];
    }
    const double* wd = P.w + (size_t)d * m;
    const double* St = P.St + (size_t)d * P.p * m;
#pragma unroll
    for (int sl = 0; sl < EHM_SLOTS; ++sl) {
        const int i = lane + 64 * sl;
        double v = 0.0;
        if (i < m) {
            v = wd[i];
            for (int q = 0; q < P.p; ++q) v = fma(St[(size_t)q * m + i], theta[q], v);
        } else if (feas && i == m) {
            v = 1.0;
        }
        b[sl] = v;
    }
    if (P.Hq && !feas) quad_point(w, P, d, theta, lane);
    wave_sync();
}

// Problems over a simplex R (rows = vertices, in LDS), variables (z, beta[, t]) with
// theta = R[0] + sum_q beta_q (R[q+1]-R[0]), beta >= 0, sum beta <= 1:
//   LP_MIN_SIMPLEX : min V                       (lib/oracle.py:74-79)
//   LP_SLACK       : max t  s.t.  sum alpha_i Vbar_i - V - eps_a >= t,
//                                 sum alpha_i Vbar_i - (1+eps_r) V >= t   (lib/oracle.py:89-97)
//   LP_FEAS_SIMPLEX: min tau s.t. MPC rows relaxed by tau, tau >= -1  (is the commutation
//                    feasible anywhere in R?  feasible <=> tau* <= 0)
__device__ inline void assemble_simplex(LpWork& w, double* smem, const DevProblem& P, int d,
                                        const double* R, const double* Vbar, int mode,
                                        double (&b)[EHM_SLOTS], int lane) {
    const int n = P.n, m = P.m, p = P.p;
    const bool slack = (mode == SX_SLACK);
    const bool feas = (mode == SX_FEAS);
    const int n_lp = n + p + ((slack || feas) ? 1 : 0);
    const int m_lp = m + p + 1 + (slack ? 2 : 0) + (feas ? 1 : 0);
    lp_carve(w, smem, n_lp, m_lp);
    lp_clear(w, lane);
    wave_sync();
    const double* Gt = P.Gt + (size_t)d * n * m;
    const double* St = P.St + (size_t)d * p * m;
    for (int j = 0; j < n; ++j) {
        const double* src = Gt + (size_t)j * m;
        double* dst = w.A + (size_t)j * w.lda;
        for (int i = lane; i < m; i += 64) dst[i] = src[i];
    }
    // beta columns on the MPC rows: -(S Dv)[i][q]
    for (int i = lane; i < m; i += 64) {
        double srow[EHM_MAX_P_DEV];
        for (int r = 0; r < p; ++r) srow[r] = St[(size_t)r * m + i];
        for (int q = 0; q < p; ++q) {
            double acc = 0.0;
            for (int r = 0; r < p; ++r) acc = fma(srow[r], R[(q + 1) * p + r] - R[r], acc);
            w.A[(size_t)(n + q) * w.lda + i] = -acc;
        }
    }
    // simplex rows and (for the slack problem) the two suboptimality rows
    if (lane < p) {
        double* col = w.A + (size_t)(n + lane) * w.lda;
        col[m + lane] = -1.0;          // -beta_q <= 0
        col[m + p] = 1.0;              // sum beta <= 1
        if (slack) {
            const double dv = Vbar[lane + 1] - Vbar[0];
            col[m + p + 1] = -dv;
            col[m + p + 2] = -dv;
        }
    }
    const bool quad = (P.Hq != nullptr) && !feas;
    if (slack && quad) {
        // the two rows are quadratic: ipm_solve writes their gradients; objective -t
        if (lane == 0) w.c[n + p] = -1.0;
    } else if (slack) {
        if (lane < n) {
            const double cj = P.c[lane];
            double* col = w.A + (size_t)lane * w.lda;
            col[m + p + 1] = cj;
            col[m + p + 2] = (1.0 + P.eps_r) * cj;
        }
        if (lane == 0) {
            double* col = w.A + (size_t)(n + p) * w.lda;
            col[m + p + 1] = 1.0;
            col[m + p + 2] = 1.0;
            w.c[n + p] = -1.0;
        }
    } else if (feas) {
        double* col = w.A + (size_t)(n + p) * w.lda;
        for (int i = lane; i < m; i += 64) col[i] = -1.0;
        if (lane == 0) {
            col[m + p + 1] = -1.0;     // -tau <= 1
            w.c[n + p] = 1.0;
        }
    } else if (lane < n) {
        w.c[lane] = P.c[lane];
    }
    const double* wd = P.w + (size_t)d * m;
#pragma unroll
    for (int sl = 0; sl < EHM_SLOTS; ++sl) {
        const int i = lane + 64 * sl;
        double v = 0.0;
        if (i < m) {
            v = wd[i];
            for (int q = 0; q < p; ++q) v = fma(St[(size_t)q * m + i], R[q], v);
        } else if (i == m + p) {
            v = 1.0;
        } else if (feas && i == m + p + 1) {
            v = 1.0;
        } else if (slack && i == m + p + 1) {
            v = Vbar[0] - P.eps_a;
        } else if (slack && i == m + p + 2) {
            v = Vbar[0];
        }
        b[sl] = v;
    }
    if (quad) {
        wave_sync();
        quad_simplex(w, P, d, R, Vbar, slack, lane);
        if (slack) {
#pragma unroll
            for (int sl = 0; sl < EHM_SLOTS; ++sl) {
                const int i = lane + 64 * sl;
                if (i == m + p + 1) b[sl] -= w.kap1 * w.v0;
                if (i == m + p + 2) b[sl] -= w.kap2 * w.v0;
            }
        }
    }
    wave_sync();
}

}  // namespace ehm
