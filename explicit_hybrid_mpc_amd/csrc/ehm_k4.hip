// Wide-LP kernels of libehmpc (gfx950) with the constant block RESIDENT IN LDS: one workgroup of
// 512 threads per LP and CU, the reduced block [G_D | -S] of the commutation copied into LDS once
// per workgroup and commutation, eliminated epigraph columns, the normal matrix AND the trailing
// updates of the blocked factorisation on the matrix cores (ehm_ipm4.h).  Taken instead of the
// streaming kernels of ehm_k3.hip whenever the reduced block fits in LDS (BASELINE.json configs
// 4 and 5: it does).  Same entry points and node semantics as ehm_k2.hip / ehm_k3.hip.
#include <hip/hip_runtime.h>

#include "ehm_k2.h"
#include "ehm_ipm4.h"

using namespace ehm;

namespace ehm4 {

// node record / parameter / gradient buffers: fixed offsets of the workgroup's LDS (ehm_ipm4.h)
struct NodeBuf {
    double* rec;    // node record / simplex vertices (+ vertex costs)
    double* th;     // p doubles (parameter / midpoint)
    double* g;      // p doubles: gradient of the optimal cost from a point solve
};
__device__ __forceinline__ void carve_node(NodeBuf& nb) {
    nb.rec = lds_at(O_REC);
    nb.th = lds_at(O_TH);
    nb.g = lds_at(O_G);
}

__device__ __forceinline__ unsigned long long low_bits(int k) {
    return (k < 64) ? ((1ULL << k) - 1ULL) : ~0ULL;
}

// P_theta_delta at one parameter value (lib/oracle.py:141-173) or its phase-one form
//   min tau  s.t.  G z - tau <= w + S theta,  tau >= -1.
// Returns the right-hand side of this thread's row.  Ends with a workgroup barrier.
__device__ __forceinline__ double assemble_point(const Blk& S, Lp& L, const double* theta, bool feas,
                                                 int tid) {
    const int n = S.n, m = S.m, p = S.p, nd0 = S.nd0;
    L.nsx = 0;
    L.kd = feas ? 1 : 0;
    L.has_beta = 0;
    L.sign_floor = EHM_ROUTE_TOL;
    L.spec_mpc = feas ? 1 : 0;
    L.act = low_bits(nd0) | (feas ? (1ULL << (nd0 + p)) : 0ULL);
    if (tid < NC) {
        L.c()[tid] = 0.0;
        L.X()[tid] = 0.0;
        L.X()[NC + tid] = 0.0;
    }
    __syncthreads();
    if (feas) {
        if (tid == 0) {
            L.c()[nd0 + p] = 1.0;
            L.X()[nd0 + p] = -1.0;            // dense row 0:  -tau <= 1
        }
    } else if (tid < n) {
        L.c()[zcol(S, tid)] = S.cv[tid];
    }
    double v = 0.0;
    if (tid < m) {
        v = S.wv[tid];
        for (int r = 0; r < p; ++r) v = fma(-S.Wb[(size_t)(nd0 + r) * S.ld + tid], theta[r], v);
    } else if (feas && tid == S.m4) {
        v = 1.0;
    }
    __syncthreads();
    return v;
}

// Problems over a simplex R (rows = vertices, in LDS) in the variables (z, beta[, t]) with
// theta = R0 + E beta, beta >= 0, sum beta <= 1 (E[r][q] = R[q+1][r] - R0[r]):
//   SX_MIN   : min V                                              (lib/oracle.py:74-79)
//   SX_SLACK : max t  s.t.  Vbar0 + dV^T beta - V - eps_a >= t,
//                           Vbar0 + dV^T beta - (1+eps_r) V >= t   (lib/oracle.py:89-97)
//   SX_FEAS  : min tau s.t. MPC rows relaxed by tau, tau >= -1
// Extra rows: the simplex rows -beta_e <= 0 (e < p), sum beta <= 1 (analytic), then the dense ones.
__device__ __forceinline__ double assemble_simplex(const Blk& S, Lp& L, const double* R,
                                                   const double* Vbar, int mode, double eps_a,
                                                   double eps_r, int tid) {
    const int n = S.n, m = S.m, p = S.p, nd0 = S.nd0;
    const bool slack = (mode == SX_SLACK);
    const bool feas = (mode == SX_FEAS);
    L.nsx = p + 1;
    L.kd = slack ? 2 : (feas ? 1 : 0);
    L.has_beta = 1;
    L.sign_floor = EHM_ROUTE_TOL * (1.0 + (slack ? fabs(Vbar[0]) : 0.0));
    L.spec_mpc = feas ? 1 : 0;
    L.act = low_bits(nd0 + p + ((slack || feas) ? 1 : 0));
    if (tid < NC) {
        L.c()[tid] = 0.0;
        L.X()[tid] = 0.0;
        L.X()[NC + tid] = 0.0;
    }
    if (tid >= 64 && tid < 64 + p * p) {
        const int k = tid - 64;
        const int r = k / p, q = k - r * p;
        L.E()[k] = R[(q + 1) * p + r] - R[r];
    }
    __syncthreads();
    if (slack) {
        if (tid < n) {
            const double cj = S.cv[tid];
            const int zc = zcol(S, tid);
            L.X()[zc] = cj;
            L.X()[NC + zc] = fma(eps_r, cj, cj);      // (1 + eps_r) c_j
        }
        if (tid >= 64 && tid < 64 + p) {
            const int q = tid - 64;
            const double dv = Vbar[q + 1] - Vbar[0];
            L.X()[nd0 + q] = -dv;
            L.X()[NC + nd0 + q] = -dv;
        }
        if (tid == 128) {
            L.X()[nd0 + p] = 1.0;
            L.X()[NC + nd0 + p] = 1.0;
            L.c()[nd0 + p] = -1.0;
        }
    } else if (feas) {
        if (tid == 0) {
            L.X()[nd0 + p] = -1.0;            // -tau <= 1
            L.c()[nd0 + p] = 1.0;
        }
    } else if (tid < n) {
        L.c()[zcol(S, tid)] = S.cv[tid];
    }
    double v = 0.0;
    if (tid < m) {
        v = S.wv[tid];
        for (int r = 0; r < p; ++r) v = fma(-S.Wb[(size_t)(nd0 + r) * S.ld + tid], R[r], v);
    } else if (tid >= S.m4) {
        const int e = tid - S.m4;
        if (e == p) v = 1.0;
        else if (feas && e == p + 1) v = 1.0;
        else if (slack && e == p + 1) v = Vbar[0] - eps_a;
        else if (slack && e == p + 2) v = Vbar[0];
    }
    __syncthreads();
    return v;
}

__device__ __forceinline__ void count_solve(DevCounters* cnt, const IpmResult& r, int tid) {
    if (tid == 0 && cnt) {
        atomicAdd(&cnt->lp_solves, 1ULL);
        atomicAdd(&cnt->ipm_iters, (unsigned long long)r.iters);
        if (r.status != 0) atomicAdd(&cnt->stalled, 1ULL);
    }
}

#define K4_PROLOGUE()                                                            \
    double* sm = reinterpret_cast<double*>(k4_smem);                             \
    const int tid0 = threadIdx.x;                                                \
    int tid = tid0;                                                              \
    Ctx B;                                                                       \
    B.tid = tid0;                                                                \
    B.lane = tid0 & 63;                                                          \
    B.wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);                          \
    B.flip = 0;                                                                  \
    NodeBuf nb;                                                                  \
    carve_node(nb);                                                              \
    Blk S;                                                                       \
    carve_blk(S, sm + O_VAR + lp_doubles(P.m, P.p, P.nd0, P.n - P.nd0), P);      \
    Lp L;                                                                        \
    carve_lp(L, sm + O_VAR, S);                                                  \
    /* the workspace starts as numbers: LDS is not cleared between kernels, and  \
       several phases multiply exact zeros with entries they never wrote */      \
    for (int k_ = tid0;                                                          \
         k_ < O_VAR + (int)lp_doubles(P.m, P.p, P.nd0, P.n - P.nd0); k_ += NT)   \
        sm[k_] = 0.0;                                                            \
    __syncthreads()

// the block of commutation d into LDS (all threads; barriers on both sides)
#define K4_USE_BLOCK(D)                                                          \
    {                                                                            \
        __syncthreads();                                                         \
        load_blk(S, P, (D), tid0);                                               \
        S.wv = P.w + (size_t)(D) * P.m;                                          \
        __syncthreads();                                                         \
    }

// ---- a2: P_theta_delta batch / its feasibility form; instances sorted by commutation -----
template <int NTILE>
EHM4_KERNEL void k4_point_batch(
    DevProblem P, long long n_inst, const double* __restrict__ theta,
    const int32_t* __restrict__ seg, int feas, double* __restrict__ J, double* __restrict__ u0,
    int32_t* __restrict__ status, int32_t* __restrict__ iters, DevCounters* cnt, K2Gather G) {
    K4_PROLOGUE();
    if (G.n_dev) n_inst = *G.n_dev;
    // instances are drawn from a ticket (DevCounters::ticket, zeroed by the launcher): they are
    // sorted by commutation, so a workgroup reloads its block only when the segment changes
    __shared__ unsigned int s_ticket;
    int d = 0, d_loaded = -1;
    for (;;) {
        __syncthreads();
        if (tid0 == 0) s_ticket = atomicAdd(&cnt->ticket, 1u);
        __syncthreads();
        const long long inst = (long long)s_ticket;
        if (inst >= n_inst) break;
        tid = pin(tid0);
        while (d + 1 < P.n_delta && seg[d + 1] <= inst) ++d;
        if (d != d_loaded) {
            K4_USE_BLOCK(d)
            d_loaded = d;
        }
        const double* tsrc = G.src ? theta + G.src[inst] : theta + inst * P.p;
        const long long o = G.dst ? (long long)G.dst[inst] : inst;
        if (tid < P.p) nb.th[tid] = tsrc[tid];
        __syncthreads();
        IpmResult r;
        int its = 0;
        for (int attempt = 0; attempt < EHM4_ATTEMPTS; ++attempt) {     // see EHM4_STEP_FRAC
            const double b = assemble_point(S, L, nb.th, feas != 0, pin(tid));
            r = ipm_solve<NTILE>(S, L, B, b, false, step_fraction(attempt));
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;
        count_solve(cnt, r, tid);
        if (tid == 0) {
            J[o] = r.obj;
            if (status) status[o] = ehm_status_word(r.status, r.merit);
            if (iters) iters[o] = r.iters;
        }
        if (u0 && tid < P.n_u) u0[o * P.n_u + tid] = L.xb()[tid];
        __syncthreads();
    }
}

// ---- a5 / a7': problems over a simplex, one commutation per instance (sorted) -------------
template <int NTILE>
EHM4_KERNEL void k4_simplex_batch(
    DevProblem P, long long n_inst, const double* __restrict__ R,
    const double* __restrict__ Vbar, const int32_t* __restrict__ seg, int mode,
    double* __restrict__ obj, double* __restrict__ alpha, int32_t* __restrict__ status,
    int32_t* __restrict__ iters, DevCounters* cnt, K2Gather G) {
    K4_PROLOGUE();
    const int p = P.p;
    const int nR = (p + 1) * p;
    if (G.n_dev) n_inst = *G.n_dev;
    __shared__ unsigned int s_ticket;
    int d = 0, d_loaded = -1;
    for (;;) {
        __syncthreads();
        if (tid0 == 0) s_ticket = atomicAdd(&cnt->ticket, 1u);
        __syncthreads();
        const long long inst = (long long)s_ticket;
        if (inst >= n_inst) break;
        tid = pin(tid0);
        while (d + 1 < P.n_delta && seg[d + 1] <= inst) ++d;
        if (d != d_loaded) {
            K4_USE_BLOCK(d)
            d_loaded = d;
        }
        double* Rl = nb.rec;
        double* Vl = nb.rec + nR;
        const double* Rsrc = G.src ? R + G.src[inst] : R + inst * nR;
        const double* Vsrc = G.src ? Rsrc + G.v_off : Vbar + inst * (p + 1);
        const long long o = G.dst ? (long long)G.dst[inst] : inst;
        for (int k = tid; k < nR; k += NT) Rl[k] = Rsrc[k];
        if (mode == SX_SLACK && tid <= p) Vl[tid] = Vsrc[tid];
        __syncthreads();
        IpmResult r;
        int its = 0;
        for (int attempt = 0; attempt < EHM4_ATTEMPTS; ++attempt) {
            const double b = assemble_simplex(S, L, Rl, Vl, mode, P.eps_a, P.eps_r, pin(tid));
            r = ipm_solve<NTILE>(S, L, B, b, false, step_fraction(attempt));
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;
        count_solve(cnt, r, tid);
        if (tid == 0) {
            obj[o] = (mode == SX_SLACK) ? -r.obj : r.obj;     // t* = -(min -t)
            if (status) status[o] = ehm_status_word(r.status, r.merit);
            if (iters) iters[o] = r.iters;
        }
        if (alpha && B.wave == 0) {
            const double beta = (tid < p) ? L.xb()[S.nd0 + tid] : 0.0;
            const double sb = wave_sum(beta);
            if (tid < p) alpha[o * (p + 1) + tid + 1] = beta;
            if (tid == 0) alpha[o * (p + 1)] = 1.0 - sb;
        }
        __syncthreads();
    }
}

// ---- frontier sweep (single commutation): epsilon-suboptimality decision per node --------
// (lib/worker.py:368-375)
template <int NTILE>
EHM4_KERNEL void k4_lcss_decide(
    DevProblem P, DevTree T, const int32_t* __restrict__ frontier, int nf,
    int32_t* __restrict__ open_flag, DevCounters* cnt, int sign_only) {
    K4_PROLOGUE();
    const int nrec = rec_doubles(P.p, P.n_u);
    K4_USE_BLOCK(0)
    // Workgroups draw frontier positions from a ticket (see ehm_k3.hip): a node costs anything
    // between a tangent-plane bound and a 20-iteration LP.
    __shared__ int s_ticket;
    __shared__ double s_bnd;
    for (;;) {
        __syncthreads();
        if (tid0 == 0) s_ticket = (int)atomicAdd(&cnt->ticket, 1u);
        __syncthreads();
        const int f = s_ticket;
        if (f >= nf) break;
        tid = pin(tid0);
        const int id = frontier[f];
        const double* rec = T.rec + (size_t)id * T.rec_stride;
        for (int k = tid; k < nrec; k += NT) nb.rec[k] = rec[k];
        __syncthreads();
        if (T.grad && sign_only) {
            // tangent-plane bound of t* (ehm_dev.h, cut_bound), evaluated by the first wavefront in
            // the LP workspace that is still free: negative => closed
            if (B.wave == 0) {
                const double bnd0 = cut_bound(nb.rec, T.grad + (size_t)id * (P.p + 1) * P.p, P.p,
                                              P.eps_a, P.eps_r, B.lane, L.M);
                if (B.lane == 0) s_bnd = bnd0;
            }
            __syncthreads();
            const double bnd = s_bnd;
            __syncthreads();
            if (bnd < -EHM_ROUTE_TOL * (1.0 + fabs(nb.rec[rec_off_vcost(P.p)]))) {
                if (tid == 0) {
                    atomicAdd(&cnt->cert_closed, 1ULL);
                    T.tstar[id] = bnd;
                    open_flag[f] = 0;
                    T.flags[id] |= 1;
                    atomicMin(&cnt->min_margin_bits,
                              (unsigned long long)__double_as_longlong(-bnd));
                }
                __syncthreads();
                continue;
            }
        }
        if (T.wit && sign_only) {
            // inherited witness (DevTree::wit): the point that proved an ancestor open, if this
            // node contains it and it still beats the interpolated cost
            const double* wv = T.wit + (size_t)id * (P.p + 2);
            const double* Vc = nb.rec + rec_off_vcost(P.p);
            double vbw = 0.0;
            for (int q = 0; q <= P.p; ++q) vbw = fma(wv[1 + q], Vc[q], vbw);
            const double cw = wv[0];
            const double tw = fmin(vbw - cw - P.eps_a, vbw - (1.0 + P.eps_r) * cw);
            if (tw > EHM_ROUTE_TOL * (1.0 + fabs(vbw))) {
                if (tid == 0) {
                    atomicAdd(&cnt->wit_inherited, 1ULL);
                    T.tstar[id] = tw;
                    open_flag[f] = 1;
                    atomicMin(&cnt->min_margin_bits,
                              (unsigned long long)__double_as_longlong(tw));
                }
                __syncthreads();
                continue;           // T.wit[id] stays: the expand kernel hands it on
            }
        }
        IpmResult r;
        int its = 0;
        for (int attempt = 0; attempt < EHM4_ATTEMPTS; ++attempt) {
            const double b = assemble_simplex(S, L, nb.rec, nb.rec + rec_off_vcost(P.p), SX_SLACK,
                                              P.eps_a, P.eps_r, pin(tid));
            r = ipm_solve<NTILE>(S, L, B, b, sign_only != 0, step_fraction(attempt));
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;
        count_solve(cnt, r, tid);
        if (T.wit && B.wave == 0) {
            // this node's own witness: the accepted iterate's parameter (barycentric) and the cost
            // of its z, raised by the safety amount (EHM_WIT_REL); none after a failed solve
            const bool ok = sign_only && r.status == 0 && -r.obj >= 0.0;
            double* wv = T.wit + (size_t)id * (P.p + 2);
            double cz = 0.0;
            for (int q = B.lane; q < P.n; q += 64) cz = fma(S.cv[q], L.xb()[zcol(S, q)], cz);
            cz = wave_sum(cz);
            const double beta = (B.lane < P.p) ? L.xb()[S.nd0 + B.lane] : 0.0;
            const double sb = wave_sum(beta);
            if (B.lane < P.p) wv[2 + B.lane] = ok ? beta : 0.0;
            if (B.lane == 0) {
                wv[0] = ok ? fma(EHM_WIT_REL, r.margin, cz) : 0.0;
                wv[1] = ok ? 1.0 - sb : 0.0;
            }
        }
        if (tid == 0) {
            if (r.status != 0) {
                atomicAdd(&cnt->errors, 1ULL);
                T.flags[id] |= 8;
            }
            atomicAdd(&cnt->slack_solves, 1ULL);
            atomicAdd(&cnt->slack_iters, (unsigned long long)r.iters);
            const double t = -r.obj;
            const bool open = (t >= 0.0);
            T.tstar[id] = t;
            open_flag[f] = open ? 1 : 0;
            if (!open) T.flags[id] |= 1;
            atomicMin(&cnt->min_margin_bits, (unsigned long long)__double_as_longlong(r.margin));
            if (r.margin < EHM_ROUTE_TOL * (1.0 + fabs(nb.rec[rec_off_vcost(P.p)])))
                atomicAdd(&cnt->routed, 1ULL);
        }
        __syncthreads();
    }
}

// ---- split every open node, solve P_theta_delta at the midpoint, write the children -------
// (lib/worker.py:403-414, 354-365)
template <int NTILE>
EHM4_KERNEL void k4_lcss_expand(
    DevProblem P, DevTree T, const int32_t* __restrict__ open_list, int n_open, int child_base,
    int32_t* __restrict__ next_frontier, DevCounters* cnt) {
    K4_PROLOGUE();
    const int p = P.p, n_u = P.n_u;
    const int nrec = rec_doubles(p, n_u);
    K4_USE_BLOCK(0)
    __shared__ int s_ticket;
    __shared__ int s_mt[3];
    __shared__ double s_mtv[2];
    const long long t_start = wall_clock64();
    for (;;) {
        __syncthreads();
        if (tid0 == 0) s_ticket = (int)atomicAdd(&cnt->ticket, 1u);
        __syncthreads();
        const int f = s_ticket;
        if (f >= n_open) break;
        tid = pin(tid0);
        const int id = open_list[f];
        const double* rec = T.rec + (size_t)id * T.rec_stride;
        double* node = nb.rec;
        double* mid = nb.th;
        for (int k = tid; k < nrec; k += NT) node[k] = rec[k];
        __syncthreads();
        int bi, bj;
        longest_edge(node, p, bi, bj);
        if (tid < p) {
#pragma clang fp contract(off)
            mid[tid] = (node[bi * p + tid] + node[bj * p + tid]) / 2.0;
        }
        __syncthreads();
        const int d = T.didx[id];
        IpmResult r;
        r.obj = 0.0;
        r.status = 0;
        r.merit = 0.0;
        // the table of midpoint optima (ehm_midtable.h) serves the sweeps: the simplices around an
        // edge all ask for this midpoint (see ehm_k3.hip)
        int mt_res = MT_NONE, mt_slot = 0;
        unsigned long long mt_tg = 0ull;
        if (T.mt.state) {
            unsigned int mt_i = 0u;
            mt_tg = mt_tag(mid, p, T.mt.mask, &mt_i);
            if (tid == 0) {
                int sl = 0;
                s_mt[0] = mt_claim(T.mt, mt_tg, mt_i, t_start, 60LL * 100000000LL, &sl);
                s_mt[1] = sl;
            }
            __syncthreads();
            mt_res = s_mt[0];
            mt_slot = s_mt[1];
            if (mt_res == MT_HIT) {
                if (B.wave == 0) {      // entry layout: ehm_midtable.h
                    bool same = false;
                    const double ev = mt_read(T.mt, mt_slot, B.lane, mid, p, &same);
                    if (B.lane == 8) s_mtv[0] = ev;
                    if (B.lane == 9) s_mtv[1] = ev;
                    if (B.lane >= 10 && B.lane < 10 + n_u) L.xb()[B.lane - 10] = ev;
                    if (T.grad && B.lane >= 18 && B.lane < 18 + p) nb.g[B.lane - 18] = ev;
                    if (B.lane == 0) s_mt[2] = same ? 1 : 0;
                }
                __syncthreads();
                if (!s_mt[2]) mt_res = MT_NONE;         // another midpoint with this tag
                else {
                    r.obj = s_mtv[0];
                    r.status = ((int)s_mtv[1]) & 0xff;
                    if (tid == 0) atomicAdd(&cnt->mid_shared, 1ULL);
                }
            }
        }
        if (mt_res != MT_HIT) {
            int its = 0;
            for (int attempt = 0; attempt < EHM4_ATTEMPTS; ++attempt) {
                const double b = assemble_point(S, L, mid, false, pin(tid));
                r = ipm_solve<NTILE>(S, L, B, b, false, step_fraction(attempt),
                                     T.grad ? nb.g : nullptr);
                its += r.iters;
                if (r.status == 0) break;
            }
            r.iters = its;
            count_solve(cnt, r, tid);
            if (mt_res == MT_OWN) {
                __syncthreads();
                if (B.wave == 0)
                    mt_publish(T.mt, mt_slot, mt_tg, B.lane, mid, p, r.obj, r.status,
                               (r.status == 0 && r.merit <= 1.0) ? 1 : 0, its, L.xb(), n_u,
                               T.grad ? nb.g : nullptr);
            }
        }
        if (r.status != 0 && tid == 0) {
            atomicAdd(&cnt->errors, 1ULL);
            T.flags[id] |= 16;
        }
        __syncthreads();
        const int c0 = child_base + 2 * f;
        if (T.grad) {       // the children inherit the vertex gradients, the midpoint's is new
            const int ng = (p + 1) * p;
            const double* gp_ = T.grad + (size_t)id * ng;
            double* g0 = T.grad + (size_t)c0 * ng;
            for (int k = tid; k < ng; k += NT) {
                const double gv = gp_[k];
                g0[k] = (k >= bi * p && k < bi * p + p) ? nb.g[k - bi * p] : gv;
                g0[ng + k] = (k >= bj * p && k < bj * p + p) ? nb.g[k - bj * p] : gv;
            }
        }
        if (T.wit && tid < p + 2) {
            // the node's witness goes to the child that contains it, its projection onto the
            // shared face to the other one (witness_for_children; all zero = none, stays so)
            const double* wv = T.wit + (size_t)id * (p + 2);
            double* w0 = T.wit + (size_t)c0 * (p + 2);
            double v0, v1;
            witness_for_children(wv, node + rec_off_vcost(p), bi, bj, tid, v0, v1);
            const bool none = wv[0] == 0.0 && wv[1 + bi] == 0.0 && wv[1 + bj] == 0.0;
            w0[tid] = none ? 0.0 : v0;
            w0[(p + 2) + tid] = none ? 0.0 : v1;
        }
        double* rec0 = T.rec + (size_t)c0 * T.rec_stride;
        double* rec1 = rec0 + T.rec_stride;
        const int ov = rec_off_vcost(p), ou = rec_off_vinput(p);
        for (int k = tid; k < nrec; k += NT) {
            double v0 = node[k], v1 = node[k];
            if (k < ov) {                       // vertices: row bi / bj replaced
                if (k >= bi * p && k < bi * p + p) v0 = mid[k - bi * p];
                if (k >= bj * p && k < bj * p + p) v1 = mid[k - bj * p];
            } else if (k < ou) {                // vertex costs
                if (k - ov == bi) v0 = r.obj;
                if (k - ov == bj) v1 = r.obj;
            } else {                            // vertex inputs
                const int q = k - ou;
                if (q >= bi * n_u && q < bi * n_u + n_u) v0 = L.xb()[q - bi * n_u];
                if (q >= bj * n_u && q < bj * n_u + n_u) v1 = L.xb()[q - bj * n_u];
            }
            rec0[k] = v0;
            rec1[k] = v1;
        }
        if (tid == 0) {
            T.left[id] = c0;
            const int dep = T.depth[id] + 1;
            T.left[c0] = -1;
            T.left[c0 + 1] = -1;
            T.didx[c0] = d;
            T.didx[c0 + 1] = d;
            T.depth[c0] = dep;
            T.depth[c0 + 1] = dep;
            T.flags[c0] = 2;
            T.flags[c0 + 1] = 2;
            T.tstar[c0] = 0.0;
            T.tstar[c0 + 1] = 0.0;
            next_frontier[2 * f] = c0;
            next_frontier[2 * f + 1] = c0 + 1;
        }
        __syncthreads();
    }
}


// ---- persistent frontier kernel of the wide family (round 6) --------------------------------
// One launch per partition, one workgroup per CU: the workgroup draws a node from the device queue
// of ehm_k2.hip's persistent kernel (same slot array and control block: PersistCtl), decides it
// exactly as k4_lcss_decide does (tangent-plane bound, inherited witness, suboptimality-test LP),
// and -- if it stays open -- bisects it as k4_lcss_expand does (midpoint optimum from the table or
// by its own LP) and queues both children.  What a consumer on another XCD reads of a child --
// record, gradients, witness, structure words -- is written THROUGH (agent-scope atomic stores),
// the stores are waited for, then the queue slots go out; the consumer reads its slot with an
// agent-scope load and invalidates before it reads the record (the protocol of ehm_k2.hip:
// no L2 write-back anywhere).  Same tree as the sweeps: a node's fate depends on its record only.
// Single rank, unbudgeted (PersistDeal: world <= 1, pop_limit = 0); anything else runs the sweeps.
#define K4_WT(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
template <int NTILE>
EHM4_KERNEL void k4_persist(
    DevProblem P, DevTree T, int32_t* slots, int n_slots, PersistCtl* ctl, int node_cap,
    DevCounters* cnt, int sign_only, int max_depth, PersistDeal deal) {
    K4_PROLOGUE();
    const int p = P.p, n_u = P.n_u;
    const int nrec = rec_doubles(p, n_u);
    K4_USE_BLOCK(0)
    __shared__ int s_id, s_c0, s_open;
    __shared__ int s_mt[3];
    __shared__ double s_mtv[2];
    __shared__ double s_bnd;
    const long long t_start = wall_clock64();
    unsigned long long n_closed = 0, n_splits = 0;      // thread 0's, added to ctl when it leaves
    int depth_seen = 0, trunc = 0;
    for (;;) {
        __syncthreads();
        if (tid0 == 0) {
            int id = -1;
            const int idx = atomicAdd(&ctl->head, 1);
            if (idx < n_slots) {
                for (;;) {
                    id = __hip_atomic_load(&slots[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (id >= 0) break;
                    if (__hip_atomic_load(&ctl->pending, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) <= 0 ||
                        __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) != 0)
                        break;
                    if (wall_clock64() - t_start > 60LL * 100000000LL) {
                        atomicMax(&ctl->abort, 3);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(64);
                }
            }
            s_id = id;
        }
        __syncthreads();
        const int id = s_id & ~EHM_REQUEUED;
        if (s_id < 0) break;
        tid = pin(tid0);
        // acquire (L1 / non-local L2 invalidate): the record behind the slot is visible
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const double* rec = T.rec + (size_t)id * T.rec_stride;
        for (int k = tid; k < nrec; k += NT) nb.rec[k] = rec[k];
        const int dep = T.depth[id];
        __syncthreads();
        // ---- the decision --------------------------------------------------------------------
        // tangent-plane bound, inherited witness (k4_lcss_decide); then -- the midpoint-first flow
        // of the shared-block persistent kernel -- the midpoint optimum BEFORE the
        // suboptimality-test LP: a node that is not closed by the bound is almost always open and
        // needs that optimum anyway, and at theta = mid the interpolated cost is (V_bi + V_bj) / 2,
        // so  t_mid = min(Vbar - J_mid - eps_a, Vbar - (1 + eps_r) J_mid) <= t*  and t_mid > 0
        // proves the node open without its LP; otherwise the LP decides as before.
        bool open = false, decided = false;
        bool hand_on = false;           // T.wit[id] is a witness worth handing to the children
        if (T.grad && sign_only) {
            if (B.wave == 0) {
                const double bnd0 = cut_bound(nb.rec, T.grad + (size_t)id * (p + 1) * p, p, P.eps_a,
                                              P.eps_r, B.lane, L.M);
                if (B.lane == 0) s_bnd = bnd0;
            }
            __syncthreads();
            const double bnd = s_bnd;
            __syncthreads();
            if (bnd < -EHM_ROUTE_TOL * (1.0 + fabs(nb.rec[rec_off_vcost(p)]))) {
                if (tid == 0) {
                    atomicAdd(&cnt->cert_closed, 1ULL);
                    T.tstar[id] = bnd;
                    T.flags[id] |= 1;
                    atomicMin(&cnt->min_margin_bits,
                              (unsigned long long)__double_as_longlong(-bnd));
                }
                decided = true;
            }
        }
        if (!decided && T.wit && sign_only) {
            const double* wv = T.wit + (size_t)id * (p + 2);
            const double* Vc = nb.rec + rec_off_vcost(p);
            double vbw = 0.0;
            for (int q = 0; q <= p; ++q) vbw = fma(wv[1 + q], Vc[q], vbw);
            const double cw = wv[0];
            const double tw = fmin(vbw - cw - P.eps_a, vbw - (1.0 + P.eps_r) * cw);
            if (tw > EHM_ROUTE_TOL * (1.0 + fabs(vbw))) {
                if (tid == 0) {
                    atomicAdd(&cnt->wit_inherited, 1ULL);
                    T.tstar[id] = tw;
                    atomicMin(&cnt->min_margin_bits,
                              (unsigned long long)__double_as_longlong(tw));
                }
                decided = true;
                open = true;
                hand_on = true;
            }
        }
        const bool can_split = !(max_depth > 0 && dep >= max_depth);
        double* node = nb.rec;
        double* mid = nb.th;
        double* xmid = lds_at(O_STASH);     // first input of the midpoint optimum
        int bi = 0, bj = 1;
        double Jm = 0.0;
        int mid_status = 0;
        if ((!decided || open) && can_split) {
            // ---- midpoint optimum: from the table of the simplices around the edge, or by an LP
            longest_edge(node, p, bi, bj);
            if (tid < p) {
#pragma clang fp contract(off)
                mid[tid] = (node[bi * p + tid] + node[bj * p + tid]) / 2.0;
            }
            __syncthreads();
            bool mid_conv = false;
            int mt_res = MT_NONE, mt_slot = 0;
            unsigned long long mt_tg = 0ull;
            if (T.mt.state) {
                unsigned int mt_i = 0u;
                mt_tg = mt_tag(mid, p, T.mt.mask, &mt_i);
                if (tid == 0) {
                    int sl = 0;
                    s_mt[0] = mt_claim(T.mt, mt_tg, mt_i, t_start, 60LL * 100000000LL, &sl);
                    s_mt[1] = sl;
                }
                __syncthreads();
                mt_res = s_mt[0];
                mt_slot = s_mt[1];
                if (mt_res == MT_HIT) {
                    if (B.wave == 0) {      // entry layout: ehm_midtable.h
                        bool same = false;
                        const double ev = mt_read(T.mt, mt_slot, B.lane, mid, p, &same);
                        if (B.lane == 8) s_mtv[0] = ev;
                        if (B.lane == 9) s_mtv[1] = ev;
                        if (B.lane >= 10 && B.lane < 10 + n_u) xmid[B.lane - 10] = ev;
                        if (T.grad && B.lane >= 18 && B.lane < 18 + p) nb.g[B.lane - 18] = ev;
                        if (B.lane == 0) s_mt[2] = same ? 1 : 0;
                    }
                    __syncthreads();
                    if (!s_mt[2]) mt_res = MT_NONE;         // another midpoint with this tag
                    else {
                        const int word = (int)s_mtv[1];
                        Jm = s_mtv[0];
                        mid_status = word & 0xff;
                        mid_conv = ((word >> 8) & 1) != 0;
                        if (tid == 0) atomicAdd(&cnt->mid_shared, 1ULL);
                    }
                }
            }
            if (mt_res != MT_HIT) {
                IpmResult r;
                int its = 0;
                for (int attempt = 0; attempt < EHM4_ATTEMPTS; ++attempt) {
                    const double b = assemble_point(S, L, mid, false, pin(tid));
                    r = ipm_solve<NTILE>(S, L, B, b, false, step_fraction(attempt),
                                         T.grad ? nb.g : nullptr);
                    its += r.iters;
                    if (r.status == 0) break;
                }
                r.iters = its;
                count_solve(cnt, r, tid);
                Jm = r.obj;
                mid_status = r.status;
                mid_conv = (r.status == 0) && (r.merit <= 1.0);     // not merely "accepted"
                if (tid < n_u) xmid[tid] = L.xb()[tid];
                if (mt_res == MT_OWN) {
                    __syncthreads();
                    if (B.wave == 0)
                        mt_publish(T.mt, mt_slot, mt_tg, B.lane, mid, p, Jm, mid_status,
                                   mid_conv ? 1 : 0, its, L.xb(), n_u, T.grad ? nb.g : nullptr);
                }
            }
            __syncthreads();
            if (sign_only && mid_conv && !decided) {
                const double* Vc = node + rec_off_vcost(p);
                const double vb = 0.5 * (Vc[bi] + Vc[bj]);
                const double tw = fmin(vb - Jm - P.eps_a, vb - (1.0 + P.eps_r) * Jm);
                if (tw > EHM_ROUTE_TOL * (1.0 + fabs(vb))) {
                    open = true;
                    decided = true;
                    if (tid == 0) {
                        atomicAdd(&cnt->wit_open, 1ULL);
                        T.tstar[id] = tw;
                        atomicMin(&cnt->min_margin_bits,
                                  (unsigned long long)__double_as_longlong(tw));
                    }
                }
            }
        }
        if (!decided && T.mt.state && sign_only) {
            // The OTHER edges' midpoints (the headline kernel's W_WITT): a neighbour that has
            // bisected one of this simplex's edges left the optimal cost at that edge's midpoint in
            // the table -- a candidate witness like the node's own midpoint (V*(mid') known,
            // interpolated cost (V_a + V_b) / 2) and, unlike it, a point the children inherit.
            // Wavefront 0: lane e looks edge e up (read-only, nobody waits).
            if (B.wave == 0) {
                const int lane = B.lane;
                const double* Vc = node + rec_off_vcost(p);
                double tw_l = -1e300, J_l = 0.0;
                int ea = 0, eb = 1, slot_l = -1;
                const int n_edges = (p + 1) * p / 2;
                double* em = L.M + 8 * (lane < n_edges ? lane : 0);     // lane e's midpoint
                if (lane < n_edges) {
                    int rem = lane;
                    while (rem >= p - ea) { rem -= (p - ea); ++ea; }
                    eb = ea + 1 + rem;
                    if (!(ea == bi && eb == bj && can_split)) {
                        {
#pragma clang fp contract(off)
                            for (int k = 0; k < p; ++k)
                                em[k] = (node[ea * p + k] + node[eb * p + k]) / 2.0;
                        }
                        unsigned int e_i = 0u;
                        const unsigned long long e_tg = mt_tag(em, p, T.mt.mask, &e_i);
                        slot_l = mt_find(T.mt, e_tg, e_i);
                    }
                }
                if (__builtin_amdgcn_ballot_w64(slot_l >= 0) != 0ull) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    if (slot_l >= 0) {
                        const double* e = T.mt.data + (size_t)slot_l * MT_DOUBLES;
                        bool same = true;
                        for (int k = 0; k < p; ++k)
                            same = same && __double_as_longlong(e[k]) == __double_as_longlong(em[k]);
                        const int word = (int)e[9];
                        if (same && (word & 0xff) == 0 && ((word >> 8) & 1)) {   // converged optimum
                            J_l = e[8];
                            const double vb = 0.5 * (Vc[ea] + Vc[eb]);
                            const double tw = fmin(vb - J_l - P.eps_a, vb - (1.0 + P.eps_r) * J_l);
                            if (tw > EHM_ROUTE_TOL * (1.0 + fabs(vb))) tw_l = tw;
                        }
                    }
                }
                const unsigned long long won = __builtin_amdgcn_ballot_w64(tw_l > -1e299);
                if (lane == 0) s_mt[0] = 0;
                if (won != 0ull) {
                    // first edge (enumeration order) whose midpoint proves the node open
                    const int src = __builtin_ctzll(won);
                    const double tw = __shfl(tw_l, src);
                    const double Jw = __shfl(J_l, src);
                    const int wa = __shfl(ea, src), wb = __shfl(eb, src);
                    if (T.wit) {
                        double* wv = T.wit + (size_t)id * (p + 2);
                        if (lane == 0) wv[0] = Jw;
                        if (lane <= p) wv[1 + lane] = (lane == wa || lane == wb) ? 0.5 : 0.0;
                    }
                    if (lane == 0) {
                        s_mt[0] = 1;
                        s_mtv[0] = tw;
                        atomicAdd(&cnt->wit_table, 1ULL);
                        T.tstar[id] = tw;
                        atomicMin(&cnt->min_margin_bits,
                                  (unsigned long long)__double_as_longlong(tw));
                    }
                }
            }
            __syncthreads();
            if (s_mt[0]) {
                open = true;
                decided = true;
                hand_on = T.wit != nullptr;
            }
            __syncthreads();
        }
        if (!decided) {
            IpmResult r;
            int its = 0;
            for (int attempt = 0; attempt < EHM4_ATTEMPTS; ++attempt) {
                const double b = assemble_simplex(S, L, nb.rec, nb.rec + rec_off_vcost(p), SX_SLACK,
                                                  P.eps_a, P.eps_r, pin(tid));
                r = ipm_solve<NTILE>(S, L, B, b, sign_only != 0, step_fraction(attempt));
                its += r.iters;
                if (r.status == 0) break;
            }
            r.iters = its;
            count_solve(cnt, r, tid);
            if (T.wit && B.wave == 0) {
                // this node's own witness (see k4_lcss_decide); its children read it below
                const bool ok = sign_only && r.status == 0 && -r.obj >= 0.0;
                double* wv = T.wit + (size_t)id * (p + 2);
                double cz = 0.0;
                for (int q = B.lane; q < P.n; q += 64) cz = fma(S.cv[q], L.xb()[zcol(S, q)], cz);
                cz = wave_sum(cz);
                const double beta = (B.lane < p) ? L.xb()[S.nd0 + B.lane] : 0.0;
                const double sb = wave_sum(beta);
                if (B.lane < p) wv[2 + B.lane] = ok ? beta : 0.0;
                if (B.lane == 0) {
                    wv[0] = ok ? fma(EHM_WIT_REL, r.margin, cz) : 0.0;
                    wv[1] = ok ? 1.0 - sb : 0.0;
                }
            }
            hand_on = true;             // (zeros after a failed or negative solve: "none")
            const double t = -r.obj;
            open = (t >= 0.0);
            if (tid == 0) {
                if (r.status != 0) {
                    atomicAdd(&cnt->errors, 1ULL);
                    T.flags[id] |= 8;
                }
                atomicAdd(&cnt->slack_solves, 1ULL);
                atomicAdd(&cnt->slack_iters, (unsigned long long)r.iters);
                T.tstar[id] = t;
                if (!open) T.flags[id] |= 1;
                atomicMin(&cnt->min_margin_bits, (unsigned long long)__double_as_longlong(r.margin));
                if (r.margin < EHM_ROUTE_TOL * (1.0 + fabs(nb.rec[rec_off_vcost(p)])))
                    atomicAdd(&cnt->routed, 1ULL);
            }
        }
        if (tid == 0) depth_seen = (dep > depth_seen) ? dep : depth_seen;
        __syncthreads();
        if (!open) {
            if (tid == 0) {
                ++n_closed;
                atomicSub(&ctl->pending, 1);
            }
            continue;
        }
        if (!can_split) {
            if (tid == 0) {
                trunc = 1;
                atomicSub(&ctl->pending, 1);
            }
            continue;
        }
        // ---- children (the midpoint optimum is in Jm / xmid / nb.g) -----------------------------
        if (mid_status != 0 && tid == 0) {
            atomicAdd(&cnt->errors, 1ULL);
            T.flags[id] |= 16;
        }
        if (tid == 0) s_c0 = atomicAdd(&ctl->n_nodes, 2);
        __syncthreads();
        const int c0 = s_c0;
        if (c0 + 2 > node_cap) {
            if (tid == 0) {
                atomicMax(&ctl->abort, 1);
                atomicSub(&ctl->pending, 1);
            }
            break;
        }
        const int d = T.didx[id];
        struct { double obj; } r = {Jm};
        if (T.grad) {       // the children inherit the vertex gradients, the midpoint's is new
            const int ng = (p + 1) * p;
            const double* gp_ = T.grad + (size_t)id * ng;
            double* g0 = T.grad + (size_t)c0 * ng;
            for (int k = tid; k < ng; k += NT) {
                const double gv = gp_[k];
                K4_WT(g0 + k, (k >= bi * p && k < bi * p + p) ? nb.g[k - bi * p] : gv);
                K4_WT(g0 + ng + k, (k >= bj * p && k < bj * p + p) ? nb.g[k - bj * p] : gv);
            }
        }
        if (T.wit && tid < p + 2) {
            const double* wv = T.wit + (size_t)id * (p + 2);
            double* w0 = T.wit + (size_t)c0 * (p + 2);
            double v0, v1;
            witness_for_children(wv, node + rec_off_vcost(p), bi, bj, tid, v0, v1);
            const bool none = !hand_on || (wv[0] == 0.0 && wv[1 + bi] == 0.0 && wv[1 + bj] == 0.0);
            K4_WT(w0 + tid, none ? 0.0 : v0);
            K4_WT(w0 + (p + 2) + tid, none ? 0.0 : v1);
        }
        double* rec0 = T.rec + (size_t)c0 * T.rec_stride;
        double* rec1 = rec0 + T.rec_stride;
        const int ov = rec_off_vcost(p), ou = rec_off_vinput(p);
        for (int k = tid; k < nrec; k += NT) {
            double v0 = node[k], v1 = node[k];
            if (k < ov) {                       // vertices: row bi / bj replaced
                if (k >= bi * p && k < bi * p + p) v0 = mid[k - bi * p];
                if (k >= bj * p && k < bj * p + p) v1 = mid[k - bj * p];
            } else if (k < ou) {                // vertex costs
                if (k - ov == bi) v0 = r.obj;
                if (k - ov == bj) v1 = r.obj;
            } else {                            // vertex inputs
                const int q = k - ou;
                if (q >= bi * n_u && q < bi * n_u + n_u) v0 = xmid[q - bi * n_u];
                if (q >= bj * n_u && q < bj * n_u + n_u) v1 = xmid[q - bj * n_u];
            }
            K4_WT(rec0 + k, v0);
            K4_WT(rec1 + k, v1);
        }
        if (tid == 0) {
            T.left[id] = c0;
            K4_WT(&T.left[c0], -1);
            K4_WT(&T.left[c0 + 1], -1);
            K4_WT(&T.didx[c0], d);
            K4_WT(&T.didx[c0 + 1], d);
            K4_WT(&T.depth[c0], dep + 1);
            K4_WT(&T.depth[c0 + 1], dep + 1);
            K4_WT(&T.flags[c0], (uint8_t)2);
            K4_WT(&T.flags[c0 + 1], (uint8_t)2);
            K4_WT(&T.tstar[c0], 0.0);
            K4_WT(&T.tstar[c0 + 1], 0.0);
        }
        // every thread's write-through stores have completed before the slots go out
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) {
            ++n_splits;
            const int t = atomicAdd(&ctl->tail, 2);
            if (t + 2 <= n_slots) {
                __hip_atomic_store(&slots[t], c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slots[t + 1], c0 + 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&ctl->pending, 1);        // -1 (this node) + 2 children
            } else {
                atomicMax(&ctl->abort, 1);
                atomicSub(&ctl->pending, 1);
            }
        }
    }
    if (tid0 == 0) {
        atomicAdd(&ctl->closed, n_closed);
        atomicAdd(&ctl->splits, n_splits);
        atomicMax(&ctl->max_depth_seen, depth_seen);
        if (trunc) atomicMax(&ctl->truncated, 1);
        atomicAdd(&cnt->prof[0], (unsigned long long)(wall_clock64() - t_start));
    }
    (void)deal;
}
#undef K4_WT

// ---- vertex solves that seed a node's costs / inputs (lib/oracle.py:416-443) ---------------
template <int NTILE>
EHM4_KERNEL void k4_vertex_solve(
    DevProblem P, DevTree T, const int32_t* __restrict__ nodes, int n_nodes, DevCounters* cnt) {
    K4_PROLOGUE();
    const int p = P.p, n_u = P.n_u;
    K4_USE_BLOCK(0)
    const int total = n_nodes * (p + 1);
    const int per = (total + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per;
    const int hi = (lo + per < total) ? lo + per : total;
    for (int t = lo; t < hi; ++t) {
        tid = pin(tid0);
        const int id = nodes[t / (p + 1)];
        const int v = t % (p + 1);
        double* rec = T.rec + (size_t)id * T.rec_stride;
        if (tid < p) nb.th[tid] = rec[v * p + tid];
        __syncthreads();
        IpmResult r;
        int its = 0;
        for (int attempt = 0; attempt < EHM4_ATTEMPTS; ++attempt) {
            const double b = assemble_point(S, L, nb.th, false, pin(tid));
            r = ipm_solve<NTILE>(S, L, B, b, false, step_fraction(attempt),
                                 T.grad ? nb.g : nullptr);
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;
        count_solve(cnt, r, tid);
        if (r.status != 0 && tid == 0) atomicAdd(&cnt->errors, 1ULL);
        if (T.grad && tid < p) T.grad[((size_t)id * (p + 1) + v) * p + tid] = nb.g[tid];
        if (tid == 0) rec[rec_off_vcost(p) + v] = r.obj;
        if (tid < n_u) rec[rec_off_vinput(p) + v * n_u + tid] = L.xb()[tid];
        __syncthreads();
    }
}

// ---- self test: workgroup reductions, the quad sum and one 16x16x4 tile product -----------
__global__ __launch_bounds__(EHM4_THREADS) void k4_selftest(double* out) {
    Ctx B;
    B.tid = threadIdx.x;
    B.lane = threadIdx.x & 63;
    B.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    B.flip = 0;
    const int tid = B.tid;
    // sum(1 + 0.5 k, k < 64) = 1072 from the first wavefront only
    double mx[2] = {(tid == 137) ? 99.0 : -(double)tid, -1.0 - tid};
    double sm[4] = {(tid < 64) ? 1.0 + 0.5 * tid : 0.0, (tid < 25) ? 1.0 : 0.0, 0.0, 0.0};
    block_reduce<2, 4>(B, mx, sm);
    // D = A B with A[i][k] = i + 1, B[k][j] = (k == 0): D[i][j] = i + 1; trace / 16 = 8.5
    const int li = B.lane & 15, lk = B.lane >> 4;
    double4v C = {0.0, 0.0, 0.0, 0.0};
    C = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(li + 1), (lk == 0) ? 1.0 : 0.0, C, 0, 0, 0);
    double tr = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (lk + 4 * r == li) tr += C[r];
    tr = wave_sum(tr);
    // quad sums: lanes 4q .. 4q+3 hold q + 1 each -> 4 (q + 1); checked on quad 5
    const double qs = quad_sum((double)((B.lane >> 2) + 1));
    const double q5 = readlane_d(qs, 21);
    if (tid == 0) {
        out[0] = sm[0];
        out[1] = mx[0];
        out[2] = sm[1];
        out[3] = frcp(3.0) + (tr / 16.0 - 8.5) + (q5 - 24.0);
        out[4] = mx[1];
    }
}

}  // namespace ehm4

// ---------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------
namespace {

using namespace ehm4;

int ntile_of(const DevProblem& P) { return (P.nd0 + P.p + 1 > 32) ? 3 : 2; }

hipError_t set_lds(int bytes) {
    const void* ks[] = {(const void*)k4_point_batch<2>,   (const void*)k4_point_batch<3>,
                        (const void*)k4_simplex_batch<2>, (const void*)k4_simplex_batch<3>,
                        (const void*)k4_lcss_decide<2>,   (const void*)k4_lcss_decide<3>,
                        (const void*)k4_lcss_expand<2>,   (const void*)k4_lcss_expand<3>,
                        (const void*)k4_vertex_solve<2>,  (const void*)k4_vertex_solve<3>,
                        (const void*)k4_persist<2>,       (const void*)k4_persist<3>};
    for (const void* k : ks) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// LDS doubles of the LP in flight (+ the node buffer); the block is the "shared" part
size_t unit_doubles_for(const DevProblem& P, int /*n_lp*/, int /*ne*/, int /*persist*/) {
    return ((size_t)O_VAR + lp_doubles(P.m, P.p, P.nd0, P.n - P.nd0) + 1) & ~(size_t)1;
}
size_t shared_doubles_for(const DevProblem& P) {
    return blk_doubles(P.m, P.p, P.nd0, P.n - P.nd0, P.LE4);
}
// the family takes a problem when its factorised columns, its rows and its eliminated block fit
int fits(const DevProblem& P, int lds_budget_bytes) {
    const int nE = P.n - P.nd0;
    if (P.nd0 + P.p + 1 > NF || nE > MAXNE) return 0;
    if (((P.m + 3) & ~3) + P.p + 3 > NT) return 0;
    if (P.n_u > P.nd0 || P.p * P.p + 64 > NT || rec_doubles(P.p, P.n_u) > 160) return 0;
    const size_t need = (shared_doubles_for(P) + unit_doubles_for(P, 0, 0, 0)) * sizeof(double);
    return need <= (size_t)lds_budget_bytes ? 1 : 0;
}

void l_point(const K2Launch& L, DevProblem P, long long n_inst, const double* theta,
             const int32_t* seg, int feas, double* J, double* u0, int32_t* status,
             int32_t* iters, DevCounters* cnt, K2Gather G) {
    (void)hipMemsetAsync(&cnt->ticket, 0, sizeof(unsigned int), L.stream);
    if (ntile_of(P) == 3)
        hipLaunchKernelGGL(k4_point_batch<3>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, n_inst, theta, seg, feas, J, u0, status, iters, cnt, G);
    else
        hipLaunchKernelGGL(k4_point_batch<2>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, n_inst, theta, seg, feas, J, u0, status, iters, cnt, G);
}
void l_simplex(const K2Launch& L, DevProblem P, long long n_inst, const double* R,
               const double* Vbar, const int32_t* seg, int mode, double* obj, double* alpha,
               int32_t* status, int32_t* iters, DevCounters* cnt, K2Gather G) {
    (void)hipMemsetAsync(&cnt->ticket, 0, sizeof(unsigned int), L.stream);
    if (ntile_of(P) == 3)
        hipLaunchKernelGGL(k4_simplex_batch<3>, dim3(L.grid), dim3(L.threads), L.lds_bytes,
                           L.stream, P, n_inst, R, Vbar, seg, mode, obj, alpha, status, iters, cnt,
                           G);
    else
        hipLaunchKernelGGL(k4_simplex_batch<2>, dim3(L.grid), dim3(L.threads), L.lds_bytes,
                           L.stream, P, n_inst, R, Vbar, seg, mode, obj, alpha, status, iters, cnt,
                           G);
}
void l_decide(const K2Launch& L, DevProblem P, DevTree T, const int32_t* frontier, int nf,
              int32_t* open_flag, DevCounters* cnt, int sign_only) {
    (void)hipMemsetAsync(&cnt->ticket, 0, sizeof(unsigned int), L.stream);
    if (ntile_of(P) == 3)
        hipLaunchKernelGGL(k4_lcss_decide<3>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, T, frontier, nf, open_flag, cnt, sign_only);
    else
        hipLaunchKernelGGL(k4_lcss_decide<2>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, T, frontier, nf, open_flag, cnt, sign_only);
}
void l_expand(const K2Launch& L, DevProblem P, DevTree T, const int32_t* open_list, int n_open,
              int child_base, int32_t* next_frontier, DevCounters* cnt) {
    (void)hipMemsetAsync(&cnt->ticket, 0, sizeof(unsigned int), L.stream);
    if (ntile_of(P) == 3)
        hipLaunchKernelGGL(k4_lcss_expand<3>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, T, open_list, n_open, child_base, next_frontier, cnt);
    else
        hipLaunchKernelGGL(k4_lcss_expand<2>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, T, open_list, n_open, child_base, next_frontier, cnt);
}
void l_vertex(const K2Launch& L, DevProblem P, DevTree T, const int32_t* nodes, int n_nodes,
              DevCounters* cnt) {
    if (ntile_of(P) == 3)
        hipLaunchKernelGGL(k4_vertex_solve<3>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, T, nodes, n_nodes, cnt);
    else
        hipLaunchKernelGGL(k4_vertex_solve<2>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, T, nodes, n_nodes, cnt);
}
void l_persist(const K2Launch& L, DevProblem P, DevTree T, int32_t* slots, int n_slots,
               PersistCtl* ctl, int node_cap, DevCounters* cnt, int sign_only, int max_depth,
               PersistDeal deal) {
    if (ntile_of(P) == 3)
        hipLaunchKernelGGL(k4_persist<3>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream, P, T,
                           slots, n_slots, ctl, node_cap, cnt, sign_only, max_depth, deal);
    else
        hipLaunchKernelGGL(k4_persist<2>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream, P, T,
                           slots, n_slots, ctl, node_cap, cnt, sign_only, max_depth, deal);
}
void l_selftest(hipStream_t stream, double* out) {
    hipLaunchKernelGGL(k4_selftest, dim3(1), dim3(EHM4_THREADS), O_VAR * sizeof(double), stream,
                       out);
}

// np = 48 factorised columns; slots in units of 64 rows; one LP per workgroup of 512 threads
const K2Api g_api = {NF,       EHM4_THREADS / 64, EHM4_THREADS, EHM4_THREADS, set_lds,
                     unit_doubles_for, shared_doubles_for, l_point, l_simplex, l_decide,
                     l_expand, l_vertex, l_selftest, l_persist, fits};

}  // namespace

extern "C" const ehm::K2Api* ehm_k4_api() { return &g_api; }

// phase timers of an experimental build (-DEHM4_PROFILE; a diagnostic hook outside include/*.h,
// exported by hand); zeros otherwise
extern "C" __attribute__((visibility("default"))) int ehm_k4_profile(unsigned long long* out,
                                                                     int reset) {
#ifdef EHM4_PROFILE
    unsigned long long zero[40] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ehm4::g_prof4), sizeof zero) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ehm4::g_prof4), zero, sizeof zero) != hipSuccess)
        return -1;
    return 1;
#else
    for (int k = 0; k < 40; ++k) out[k] = 0;
    (void)reset;
    return 0;
#endif
}
