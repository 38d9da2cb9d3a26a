// Assembly of the oracle problems for the shared-block kernels: per-wavefront node scratch,
// the quadratic block of the cost, P_theta_delta / simplex problems (what used to be the top of
// ehm_k2.hip).  NO include guard on purpose: like ehm_ipm2.h it is compiled once per column
// capacity EHM_NP into its own namespace, and ehm_kp.hip includes it twice -- the persistent
// frontier kernel solves its suboptimality-test LPs and its midpoint LPs at different widths.
#include <hip/hip_runtime.h>

#include "ehm_k2.h"
#include "ehm_ipm2.h"

#ifndef EHM_K2_SMEM_DECLARED
#define EHM_K2_SMEM_DECLARED
extern __shared__ __attribute__((aligned(16))) char k2_smem[];
#endif

namespace EHM2_NS {

// ---------------------------------------------------------------------------------------
// per-wave scratch in front of the LP workspace: node record, simplex inverse, parameter
// ---------------------------------------------------------------------------------------
struct NodeBuf {
    double* rec;    // node record / simplex vertices (+ vertex costs)
    double* aug;    // (unused scratch)
    double* F;      // p x p edge matrix E[r][q] = R[q+1][r] - R[0][r]
    double* th;     // p doubles (parameter / midpoint)
    double* lp;     // start of the LP workspace
};
__device__ __forceinline__ void carve_node(NodeBuf& nb, double* base, int p, int n_u) {
    const int nrec = (rec_doubles(p, n_u) + 7) & ~7;
    nb.rec = base;
    nb.aug = base + nrec;
    nb.F = nb.aug + ((2 * p * p > K2_AUG_MIN) ? 2 * p * p : K2_AUG_MIN);
    nb.th = nb.F + p * p;
    nb.lp = base + k2_node_doubles(p, n_u);
}

#if EHM2_QUAD
// Quadratic block of the cost over the LP variables (DevProblem::Hq set; same construction as
// quad_point / quad_simplex of ehm_kernels.h, H / F / C read from L2).
__device__ inline void quad_point(Wave& W, const DevProblem& P, int d, const double* theta,
                                  int lane) {
    const int n = P.n, p = P.p;
    const double* H = P.Hq + (size_t)d * n * n;
    for (int k = lane; k < n * n; k += 64) W.Q[(k / n) * LDM + (k % n)] = H[k];
    if (lane < n) {
        const double* F = P.Fq + (size_t)d * p * n;
        double v = P.c[lane] + P.f0q[(size_t)d * n + lane];
        for (int q = 0; q < p; ++q) v = fma(F[(size_t)q * n + lane], theta[q], v);
        W.qv[lane] = v;
        W.c[lane] = 0.0;
    }
    const double* C = P.Cq + (size_t)d * p * p;
    double v0 = P.c0q[d];
    for (int r = 0; r < p; ++r) {
        double cr = P.c1q[(size_t)d * p + r];
        for (int q = 0; q < p; ++q) cr = fma(0.5 * C[r * p + q], theta[q], cr);
        v0 = fma(cr, theta[r], v0);
    }
    W.quad = 1;
    W.eq = -1;
    W.kap0 = 1.0;
    W.v0 = v0;
}

// theta = R0 + E beta (E = the edge matrix already in LDS): Q = [H, F E; E'F', E'C E],
// q = [c + f0 + F R0; E'(C R0 + c1)]; slack: rows p+1, p+2 of the extras are the quadratic ones.
__device__ inline void quad_simplex(Wave& W, const DevProblem& P, int d, const double* E,
                                    const double* R, const double* Vbar, bool slack,
                                    double eps_a, double eps_r, double (&b)[SLOTS], int lane) {
    const int n = P.n, p = P.p;
    const int nl = n + p + (slack ? 1 : 0);
    const double* H = P.Hq + (size_t)d * n * n;
    const double* F = P.Fq + (size_t)d * p * n;
    const double* C = P.Cq + (size_t)d * p * p;
    for (int k = lane; k < nl * LDM; k += 64) W.Q[k] = 0.0;
    wsync();
    for (int k = lane; k < n * n; k += 64) W.Q[(k / n) * LDM + (k % n)] = H[k];
    if (lane < n) {
        double v = P.c[lane] + P.f0q[(size_t)d * n + lane];
        for (int r = 0; r < p; ++r) v = fma(F[(size_t)r * n + lane], R[r], v);
        W.qv[lane] = v;
        for (int q = 0; q < p; ++q) {
            double acc = 0.0;
            for (int r = 0; r < p; ++r) acc = fma(F[(size_t)r * n + lane], E[r * p + q], acc);
            W.Q[lane * LDM + n + q] = acc;
            W.Q[(n + q) * LDM + lane] = acc;
        }
    }
    if (lane < p * p) {                       // E'C E, one entry per lane (p <= 8)
        const int q = lane / p, q2 = lane % p;
        double acc = 0.0;
        for (int r = 0; r < p; ++r) {
            double ce = 0.0;
            for (int r2 = 0; r2 < p; ++r2) ce = fma(C[r * p + r2], E[r2 * p + q2], ce);
            acc = fma(E[r * p + q], ce, acc);
        }
        W.Q[(n + q) * LDM + n + q2] = acc;
    }
    if (lane < p) {                           // E'(C R0 + c1)
        double acc = 0.0;
        for (int r = 0; r < p; ++r) {
            double cr = P.c1q[(size_t)d * p + r];
            for (int r2 = 0; r2 < p; ++r2) cr = fma(C[r * p + r2], R[r2], cr);
            acc = fma(E[r * p + lane], cr, acc);
        }
        W.qv[n + lane] = acc;
    }
    double v0 = P.c0q[d];
    for (int r = 0; r < p; ++r) {
        double cr = P.c1q[(size_t)d * p + r];
        for (int q = 0; q < p; ++q) cr = fma(0.5 * C[r * p + q], R[q], cr);
        v0 = fma(cr, R[r], v0);
    }
    W.quad = 1;
    W.v0 = v0;
    if (slack) {
        if (lane < NP) {
            double a = 0.0;
            if (lane >= n && lane < n + p) a = -(Vbar[lane - n + 1] - Vbar[0]);
            if (lane == n + p) a = 1.0;
            W.a1[lane] = a;
            W.a2[lane] = a;
        }
        if (lane == 0) W.qv[n + p] = 0.0;
        W.eq = p + 1;
        W.kap0 = 0.0;
        W.kap1 = 1.0;
        W.kap2 = 1.0 + eps_r;
        W.bq1 = Vbar[0] - eps_a - W.kap1 * v0;
        W.bq2 = Vbar[0] - W.kap2 * v0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int e = lane + 64 * sl - W.xbase;
            if (e == p + 1) b[sl] = W.bq1;
            if (e == p + 2) b[sl] = W.bq2;
        }
    } else {
        if (lane < NP) W.c[lane] = 0.0;
        W.eq = -1;
        W.kap0 = 1.0;
    }
}
#endif

#if EHM2_QUAD
// Gradient of the optimal value of a quadratic-cost point problem: the solver delivers the
// constraint part -S^T lambda (ipm_solve, gout); the envelope theorem adds the explicit
// dependence of the cost on the parameter at the optimum z*: F^T z* + C theta + c1.
__device__ inline void quad_grad_add(const Wave& W, const DevProblem& P, int d,
                                     const double* theta, double* gout, int lane) {
    if (!P.Hq) return;
    const int n = P.n, p = P.p;
    if (lane < p) {
        const double* Fq = P.Fq + ((size_t)d * p + lane) * n;      // column `lane` of F
        double a = P.c1q[(size_t)d * p + lane];
        for (int j = 0; j < n; ++j) a = fma(Fq[j], W.xb[j], a);
        const double* C = P.Cq + (size_t)d * p * p;
        for (int r = 0; r < p; ++r) a = fma(C[lane * p + r], theta[r], a);
        gout[lane] += a;                                            // NaN (unknown) stays NaN
    }
    wsync();
}
#endif

// P_theta_delta at one parameter value (lib/oracle.py:141-173) or its phase-one form
//   min tau  s.t.  G z - tau <= w + S theta,  tau >= -1.
__device__ inline void assemble_point(const Shared& S, Wave& W, double* lp_base,
                                      const double* theta, bool feas, double (&b)[SLOTS],
                                      int lane, const DevProblem& P, int d) {
    const int n = S.n, m = S.m, p = S.p, nd0 = S.nd0;
    carve_wave(W, lp_base, n + (feas ? 1 : 0), feas ? 1 : 0, m, S.nE, 0);
    // internal column order (ehm_ipm2.h): [z_D | tau] factorised, [z_E] behind them
    W.n_lin = nd0;
    W.spec_col = S.colOne;        // the column of -1
    W.n_mpc = W.nr;
    W.sign_floor = EHM_ROUTE_TOL;     // phase one: the optimum is compared with ~0
    if (lane < W.n_lp) {
        // original z index of internal column `lane` (-1: tau)
        const int jo = (lane < nd0) ? lane : ((lane >= W.nr) ? (nd0 + lane - W.nr) : -1);
        W.c[lane] = feas ? ((jo < 0) ? 1.0 : 0.0) : S.cv[jo < 0 ? 0 : jo];
        // extra row 0:  -tau <= 1      (ldx = 1)
        if (feas) W.X[lane] = (jo < 0) ? -1.0 : 0.0;
    }
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int i = lane + 64 * sl;
        double v = 0.0;
        if (i < m) {
            v = S.wv[i];
            for (int r = 0; r < p; ++r)
                v = fma(-S.Wc[(size_t)(S.colS + r) * S.lda + i], theta[r], v);
        } else if (feas && i == W.xbase) {
            v = 1.0;
        }
        b[sl] = v;
    }
#if EHM2_QUAD
    if (P.Hq && !feas) quad_point(W, P, d, theta, lane);
#endif
    wsync();
}

// Problems over a simplex R (rows = vertices, in LDS) in the variables (z, beta[, t]) with
// theta = R0 + E beta, beta >= 0, sum beta <= 1 (E[r][q] = R[q+1][r] - R0[r]):
//   SX_MIN   : min V                                              (lib/oracle.py:74-79)
//   SX_SLACK : max t  s.t.  Vbar0 + dV^T beta - V - eps_a >= t,
//                           Vbar0 + dV^T beta - (1+eps_r) V >= t   (lib/oracle.py:89-97)
//   SX_FEAS  : min tau s.t. MPC rows relaxed by tau, tau >= -1
// The MPC rows  G z - S E beta <= w + S R0  are not stored: the solver applies the shared
// block [G | -S] to psi = E beta (ehm_ipm2.h).  Extra rows: e < p facets -beta_e <= 0, e = p
// sum beta <= 1, then the dense ones.
__device__ inline void assemble_simplex(const Shared& S, Wave& W, const NodeBuf& nb,
                                        const double* R, const double* Vbar, int mode,
                                        double eps_a, double eps_r, double (&b)[SLOTS],
                                        int lane, const DevProblem& P, int d) {
    const int n = S.n, m = S.m, p = S.p, nd0 = S.nd0;
    const bool slack = (mode == SX_SLACK);
    const bool feas = (mode == SX_FEAS);
    const int n_lp = n + p + ((slack || feas) ? 1 : 0);
    const int ne = p + 1 + (slack ? 2 : 0) + (feas ? 1 : 0);
    carve_wave(W, nb.lp, n_lp, ne, m, S.nE, p + 1);
    // internal column order (ehm_ipm2.h): [z_D | beta | t or tau] factorised, [z_E] behind them;
    // nb0 = first weight column, nt = the column of t / tau
    const int nb0 = nd0, nt = nd0 + p;
    W.n_lin = nd0 + p;
    W.spec_col = slack ? S.colZero : S.colOne;    // zeros for t, -1 for tau
    W.n_mpc = slack ? (nd0 + p) : W.nr;
    W.E = nb.F;
    W.psi0 = nb0;
    W.npsi = p;
    W.sign_floor = EHM_ROUTE_TOL * (1.0 + (slack ? fabs(Vbar[0]) : 0.0));
    const int ldx = W.ldx;
    for (int k = lane; k < n_lp * ldx; k += 64) W.X[k] = 0.0;
    if (lane < n_lp) W.c[lane] = 0.0;
    {   // edge matrix
        const int r = lane >> 3, q = lane & 7;
        if (r < p && q < p) nb.F[r * p + q] = R[(q + 1) * p + r] - R[r];
    }
    wsync();
    // (the simplex rows -beta_q <= 0, sum beta <= 1 are not stored: the solver knows them;
    // X holds the dense rows only, row 0 = extra row p + 1)
    if (lane < p && slack) {
        const double dv = Vbar[lane + 1] - Vbar[0];
        W.X[(nb0 + lane) * ldx] = -dv;
        W.X[(nb0 + lane) * ldx + 1] = -dv;
    }
#if EHM2_QUAD
    const bool quadc = (P.Hq != nullptr) && !feas;
#else
    const bool quadc = false;
#endif
    if (slack && quadc) {
        // the two suboptimality rows are quadratic: ipm_solve writes their gradients
        if (lane == 0) W.c[nt] = const_d(-1.0);
    } else if (slack) {
        if (lane < n) {
            const double cj = S.cv[lane];
            const int jc = zcol(W, S, lane);
            W.X[jc * ldx] = cj;
            W.X[jc * ldx + 1] = fma(eps_r, cj, cj);           // (1 + eps_r) c_j
        }
        if (lane == 0) {
            W.X[nt * ldx] = const_d(1.0);
            W.X[nt * ldx + 1] = const_d(1.0);
            W.c[nt] = const_d(-1.0);
        }
    } else if (feas) {
        if (lane == 0) {
            W.X[nt * ldx] = const_d(-1.0);                // -tau <= 1
            W.c[nt] = const_d(1.0);
        }
    } else if (lane < n) {
        W.c[zcol(W, S, lane)] = S.cv[lane];
    }
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int i = lane + 64 * sl;
        double v = 0.0;
        if (i < m) {
            v = S.wv[i];
            for (int r = 0; r < p; ++r) v = fma(-S.Wc[(size_t)(S.colS + r) * S.lda + i], R[r], v);
        } else {
            const int e = i - W.xbase;
            if (e == p) v = 1.0;
            else if (feas && e == p + 1) v = 1.0;
            else if (slack && e == p + 1) v = Vbar[0] - eps_a;
            else if (slack && e == p + 2) v = Vbar[0];
        }
        b[sl] = v;
    }
#if EHM2_QUAD
    if (quadc) {
        wsync();
        quad_simplex(W, P, d, nb.F, R, Vbar, slack, eps_a, eps_r, b, lane);
    }
#endif
    wsync();
}

__device__ __forceinline__ void count_solve(DevCounters* cnt, const IpmResult& r, int lane) {
    if (lane == 0 && cnt) {
        atomicAdd(&cnt->lp_solves, 1ULL);
        atomicAdd(&cnt->ipm_iters, (unsigned long long)r.iters);
        if (r.status != 0) atomicAdd(&cnt->stalled, 1ULL);
    }
}

// Work distribution inside a launch: the workgroup owns a contiguous range of items and its
// wavefronts pull the next one from an LDS counter.
__device__ __forceinline__ int pull(int* ctr, int lane) {
    int f = 0;
    if (lane == 0) f = atomicAdd(ctr, 1);
    return __builtin_amdgcn_readfirstlane(f);
}


}  // namespace EHM2_NS
