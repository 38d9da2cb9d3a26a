// Second-generation kernels of libehmpc (gfx950): one copy of the commutation's constant LP
// block in LDS per workgroup, several wavefronts per workgroup, each solving its own LP.
// Compiled once per (EHM_NP, EHM_SLOTS) pair; ehm_capi.hip picks the instance that fits
// the LP of each launch.  See ehm_ipm2.h for the solver, ehm_k2_asm.h for the assembly of the
// oracle problems and DESIGN.md section 3.
#include <hip/hip_runtime.h>

#include "ehm_k2_asm.h"

using namespace ehm;

namespace EHM2_NS {

#if EHM2_PROF
#define K2_PROF_HOOK(S) S.gprof = cnt ? cnt->phase : nullptr;
#else
#define K2_PROF_HOOK(S)
#endif
#define K2_PROLOGUE()                                                            \
    double* sm = reinterpret_cast<double*>(k2_smem);                             \
    __shared__ int s_ctr;                                                        \
    const int tid = threadIdx.x;                                                 \
    const int lane0 = tid & 63;                                                  \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   /* uniform: SGPR pointers */ \
    Shared S;                                                                    \
    carve_shared(S, sm, P);                                                      \
    K2_PROF_HOOK(S)                                                              \
    NodeBuf nb;                                                                  \
    carve_node(nb, sm + shared_doubles(P) + (size_t)wave * wave_doubles, P.p, P.n_u)

// ---- a2: P_theta_delta batch / its feasibility form; instances sorted by commutation -----
// (compiled once per kind: with the kind a compile-time constant the assembly of the other kinds
// is not in the kernel, which is what keeps it near the register budget)
template <int feas>
__global__ __launch_bounds__(EHM_K2_THREADS) void k2_point_batch(
    DevProblem P, long long n_inst, const double* __restrict__ theta,
    const int32_t* __restrict__ seg, double* __restrict__ J, double* __restrict__ u0,
    int32_t* __restrict__ status, int32_t* __restrict__ iters, DevCounters* cnt,
    int wave_doubles, K2Gather G) {
    K2_PROLOGUE();
    if (G.n_dev) n_inst = *G.n_dev;
    const long long t_start = wall_clock64();
    const long long per = (n_inst + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per;
    const long long hi = (lo + per < n_inst) ? lo + per : n_inst;
    long long pos = lo;
    while (pos < hi) {      // workgroup-uniform
        int d = 0;
        while (d + 1 < P.n_delta && seg[d + 1] <= pos) ++d;
        const long long run_end = (seg[d + 1] < hi) ? seg[d + 1] : hi;
        __syncthreads();
        load_shared(P, d, sm, tid, blockDim.x);
        use_commutation(S, P, d);
        if (tid == 0) s_ctr = (int)(pos - lo);
        __syncthreads();
        for (;;) {
            const long long inst = lo + pull(&s_ctr, lane0);
            if (inst >= run_end) break;
            const int lane = pin(lane0);
            const double* tsrc = G.src ? theta + G.src[inst] : theta + inst * P.p;
            const long long o = G.dst ? (long long)G.dst[inst] : inst;
            if (lane < P.p) nb.th[lane] = tsrc[lane];
            wsync();
            // shared results (K2Gather::pt): the first wavefront to ask for (parameter,
            // commutation, kind) solves and publishes, the others take the entry -- an LP's
            // optimum does not depend on who solves it
            int pt_res = MT_NONE, pt_slot = 0;
            unsigned long long pt_tg = 0ull;
            const unsigned int pt_kind = (unsigned int)((d * 2 + (feas ? 1 : 0)) * 4 + G.sign_mode);
            if (G.pt.state && !G.grad) {
                unsigned int pt_i = 0u;
                pt_tg = pt_tag(nb.th, P.p, pt_kind, G.pt.mask, &pt_i);
                if (lane == 0)
                    pt_res = mt_claim(G.pt, pt_tg, pt_i, t_start, 60LL * 100000000LL, &pt_slot);
                pt_res = __builtin_amdgcn_readfirstlane(pt_res);
                pt_slot = __builtin_amdgcn_readfirstlane(pt_slot);
                if (pt_res == MT_HIT) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    const double* e = G.pt.data + (size_t)pt_slot * MT_DOUBLES;
                    const double ev = lane < MT_DOUBLES ? e[lane] : 0.0;
                    const bool differs =
                        (lane < P.p && __double_as_longlong(ev) != __double_as_longlong(nb.th[lane])) ||
                        (lane == 26 && ev != (double)pt_kind);
                    if (__builtin_amdgcn_ballot_w64(differs) == 0ull) {
                        const double Jv = __shfl(ev, 8);
                        const int word = (int)__shfl(ev, 9);
                        if (lane == 0) {
                            J[o] = Jv;
                            if (status) status[o] = word;
                            if (iters) iters[o] = 0;
                            atomicAdd(&cnt->mid_shared, 1ULL);
                        }
                        if (u0 && lane >= 10 && lane < 10 + P.n_u) u0[o * P.n_u + lane - 10] = ev;
                        wsync();
                        continue;
                    }
                    pt_res = MT_NONE;           // another problem with this tag
                }
            }
            Wave W;
            IpmResult r;
            int its = 0;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {     // see EHM2_STEP_FRAC
                double b[SLOTS];
                const int ln = pin(lane);     // nothing of the assembly outlives the attempt
                assemble_point(S, W, nb.lp, nb.th, feas != 0, b, ln, P, d);
                r = ipm_solve(S, W, b, ln, feas ? G.sign_mode : 0,
                              step_fraction(attempt), (G.grad && !feas) ? nb.F : nullptr);
                its += r.iters;
                if (r.status == 0) break;
            }
            r.iters = its;
            count_solve(cnt, r, lane);
            if (G.grad && !feas) {
#if EHM2_QUAD
                quad_grad_add(W, P, d, nb.th, nb.F, lane);
#endif
                if (lane < P.p) G.grad[o * P.p + lane] = nb.F[lane];
            }
            const double Jout = (feas && G.sign_mode) ? copysign(r.margin, r.obj) : r.obj;
            if (lane == 0) {
                J[o] = Jout;
                if (status) status[o] = ehm_status_word(r.status, r.merit);
                if (iters) iters[o] = r.iters;
            }
            if (u0 && lane < P.n_u) u0[o * P.n_u + lane] = W.xb[lane];
            if (pt_res == MT_OWN) {
                double* e = G.pt.data + (size_t)pt_slot * MT_DOUBLES;
                if (lane < MT_DOUBLES) {
                    double v = 0.0;
                    if (lane < 8) v = lane < P.p ? nb.th[lane] : 0.0;
                    else if (lane == 8) v = Jout;
                    else if (lane == 9) v = (double)r.status;
                    else if (lane < 18) v = lane - 10 < P.n_u ? W.xb[lane - 10] : 0.0;
                    else if (lane == 26) v = (double)pt_kind;
                    __hip_atomic_store(e + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
                if (lane == 0)
                    __hip_atomic_store(&G.pt.state[pt_slot], pt_tg | 3ull, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            wsync();
        }
        pos = run_end;
    }
}

// ---- a5 / a7': problems over a simplex, one commutation per instance (sorted) -------------
template <int mode>
__global__ __launch_bounds__(EHM_K2_THREADS) void k2_simplex_batch(
    DevProblem P, long long n_inst, const double* __restrict__ R,
    const double* __restrict__ Vbar, const int32_t* __restrict__ seg,
    double* __restrict__ obj, double* __restrict__ alpha, int32_t* __restrict__ status,
    int32_t* __restrict__ iters, DevCounters* cnt, int wave_doubles, K2Gather G) {
    K2_PROLOGUE();
    const int p = P.p;
    const int nR = (p + 1) * p;
    if (G.n_dev) n_inst = *G.n_dev;
    const long long per = (n_inst + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per;
    const long long hi = (lo + per < n_inst) ? lo + per : n_inst;
    long long pos = lo;
    while (pos < hi) {
        int d = 0;
        while (d + 1 < P.n_delta && seg[d + 1] <= pos) ++d;
        const long long run_end = (seg[d + 1] < hi) ? seg[d + 1] : hi;
        __syncthreads();
        load_shared(P, d, sm, tid, blockDim.x);
        use_commutation(S, P, d);
        if (tid == 0) s_ctr = (int)(pos - lo);
        __syncthreads();
        for (;;) {
            const long long inst = lo + pull(&s_ctr, lane0);
            if (inst >= run_end) break;
            const int lane = pin(lane0);
            double* Rl = nb.rec;
            double* Vl = nb.rec + nR;
            const double* Rsrc = G.src ? R + G.src[inst] : R + inst * nR;
            const double* Vsrc = G.src ? Rsrc + G.v_off : Vbar + inst * (p + 1);
            const long long o = G.dst ? (long long)G.dst[inst] : inst;
            for (int k = lane; k < nR; k += 64) Rl[k] = Rsrc[k];
            if (mode == SX_SLACK && lane <= p) Vl[lane] = Vsrc[lane];
            wsync();
            Wave W;
            IpmResult r;
            int its = 0;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
                double b[SLOTS];
                const int ln = pin(lane);
                assemble_simplex(S, W, nb, Rl, Vl, mode, P.eps_a, P.eps_r, b, ln, P, d);
                r = ipm_solve(S, W, b, ln, (mode == SX_MIN) ? 0 : G.sign_mode,
                              step_fraction(attempt));
                its += r.iters;
                if (r.status == 0) break;
            }
            r.iters = its;
            count_solve(cnt, r, lane);
            if (lane == 0) {
                const double val = (G.sign_mode && mode != SX_MIN) ? copysign(r.margin, r.obj) : r.obj;
                obj[o] = (mode == SX_SLACK) ? -val : val;         // t* = -(min -t)
                if (status) status[o] = ehm_status_word(r.status, r.merit);
                if (iters) iters[o] = r.iters;
            }
            if (alpha) {
                const double beta = (lane < p) ? W.xb[W.psi0 + lane] : 0.0;
                const double sb = wave_sum(beta);
                if (lane < p) alpha[o * (p + 1) + lane + 1] = beta;
                if (lane == 0) alpha[o * (p + 1)] = 1.0 - sb;
            }
            wsync();
        }
        pos = run_end;
    }
}

// ---- frontier sweep (single commutation): epsilon-suboptimality decision per node --------
// (lib/worker.py:368-375)
__global__ __launch_bounds__(EHM_K2_THREADS) void k2_lcss_decide(
    DevProblem P, DevTree T, const int32_t* __restrict__ frontier, int nf,
    int32_t* __restrict__ open_flag, DevCounters* cnt, int wave_doubles, int sign_only) {
    K2_PROLOGUE();
    const int nrec = rec_doubles(P.p, P.n_u);
    const int per = (nf + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per;
    const int hi = (lo + per < nf) ? lo + per : nf;
    if (lo >= hi) return;
    load_shared(P, 0, sm, tid, blockDim.x);
    if (tid == 0) s_ctr = lo;
    __syncthreads();
    for (;;) {
        const int f = pull(&s_ctr, lane0);
        if (f >= hi) break;
        const int lane = pin(lane0);
        const int id = frontier[f];
        const double* rec = T.rec + (size_t)id * T.rec_stride;
        for (int k = lane; k < nrec; k += 64) nb.rec[k] = rec[k];
        wsync();
        if (T.grad && sign_only) {
            // tangent-plane bound of t* (ehm_dev.h, cut_bound): negative => the leaf is closed,
            // exactly as a negative t* would close it, without solving the LP
            const double thr = -EHM_ROUTE_TOL * (1.0 + fabs(nb.rec[rec_off_vcost(P.p)]));
            const double bnd = cut_bound(nb.rec, T.grad + (size_t)id * (P.p + 1) * P.p, P.p,
                                         P.eps_a, P.eps_r, lane, nb.lp, thr);
            if (bnd < thr) {
                if (lane == 0) {
                    atomicAdd(&cnt->cert_closed, 1ULL);
                    T.tstar[id] = bnd;
                    open_flag[f] = 0;
                    T.flags[id] |= 1;
                    atomicMin(&cnt->min_margin_bits,
                              (unsigned long long)__double_as_longlong(-bnd));
                }
                wsync();
                continue;
            }
        }
        Wave W;
        IpmResult r;
        int its = 0;
        for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
            double b[SLOTS];
            const int ln = pin(lane);
            assemble_simplex(S, W, nb, nb.rec, nb.rec + rec_off_vcost(P.p), SX_SLACK, P.eps_a,
                             P.eps_r, b, ln, P, 0);
            r = ipm_solve(S, W, b, ln, sign_only != 0,
                          step_fraction(attempt));
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;
        count_solve(cnt, r, lane);
        if (lane == 0) {
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
        wsync();
    }
}

// ---- split every open node, solve P_theta_delta at the midpoint, write the children -------
// (lib/worker.py:403-414, 354-365)
__global__ __launch_bounds__(EHM_K2_THREADS) void k2_lcss_expand(
    DevProblem P, DevTree T, const int32_t* __restrict__ open_list, int n_open, int child_base,
    int32_t* __restrict__ next_frontier, DevCounters* cnt, int wave_doubles) {
    K2_PROLOGUE();
    const int p = P.p, n_u = P.n_u;
    const int nrec = rec_doubles(p, n_u);
    const int per = (n_open + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per;
    const int hi = (lo + per < n_open) ? lo + per : n_open;
    if (lo >= hi) return;
    load_shared(P, 0, sm, tid, blockDim.x);
    if (tid == 0) s_ctr = lo;
    __syncthreads();
    for (;;) {
        const int f = pull(&s_ctr, lane0);
        if (f >= hi) break;
        const int lane = pin(lane0);
        const int id = open_list[f];
        const double* rec = T.rec + (size_t)id * T.rec_stride;
        double* node = nb.rec;
        double* mid = nb.th;
        for (int k = lane; k < nrec; k += 64) node[k] = rec[k];
        wsync();
        int bi, bj;
        longest_edge_wave(node, p, lane, bi, bj);
        if (lane < p) {
#pragma clang fp contract(off)
            mid[lane] = (node[bi * p + lane] + node[bj * p + lane]) / 2.0;
        }
        wsync();
        const int d = T.didx[id];
        Wave W;
        IpmResult r;
        int its = 0;
        for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
            double b[SLOTS];
            const int ln = pin(lane);
            assemble_point(S, W, nb.lp, mid, false, b, ln, P, 0);
            r = ipm_solve(S, W, b, ln, false, step_fraction(attempt), T.grad ? nb.F : nullptr);
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;
#if EHM2_QUAD
        if (T.grad) quad_grad_add(W, P, 0, mid, nb.F, lane);
#endif
        count_solve(cnt, r, lane);
        if (r.status != 0 && lane == 0) {
            atomicAdd(&cnt->errors, 1ULL);
            T.flags[id] |= 16;
        }
        const int c0 = child_base + 2 * f;
        if (T.grad) {       // the children inherit the vertex gradients, the midpoint's is new
            const int ng = (p + 1) * p;
            const double* gp_ = T.grad + (size_t)id * ng;
            double* g0 = T.grad + (size_t)c0 * ng;
            for (int k = lane; k < ng; k += 64) {
                const double gv = gp_[k];
                g0[k] = (k >= bi * p && k < bi * p + p) ? nb.F[k - bi * p] : gv;
                g0[ng + k] = (k >= bj * p && k < bj * p + p) ? nb.F[k - bj * p] : gv;
            }
        }
        if (T.wit && lane < 2 * (p + 2))    // the sweeps do not hand witnesses on (DevTree::wit)
            T.wit[(size_t)c0 * (p + 2) + lane] = 0.0;
        double* rec0 = T.rec + (size_t)c0 * T.rec_stride;
        double* rec1 = rec0 + T.rec_stride;
        const int ov = rec_off_vcost(p), ou = rec_off_vinput(p);
        for (int k = lane; k < nrec; k += 64) {
            double v0 = node[k], v1 = node[k];
            // row bi / bj of each block is replaced (range tests instead of k / p)
            if (k < ov) {                       // vertices
                if (k >= bi * p && k < bi * p + p) v0 = mid[k - bi * p];
                if (k >= bj * p && k < bj * p + p) v1 = mid[k - bj * p];
            } else if (k < ou) {                // vertex costs
                if (k - ov == bi) v0 = r.obj;
                if (k - ov == bj) v1 = r.obj;
            } else {                            // vertex inputs
                const int q = k - ou;
                if (q >= bi * n_u && q < bi * n_u + n_u) v0 = W.xb[q - bi * n_u];
                if (q >= bj * n_u && q < bj * n_u + n_u) v1 = W.xb[q - bj * n_u];
            }
            rec0[k] = v0;
            rec1[k] = v1;
        }
        if (lane == 0) {
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
        wsync();
    }
}


// ---- persistent frontier kernel ------------------------------------------------------------
// One launch grows the whole partition: every wavefront pops node ids from a global queue,
// runs the suboptimality test (lib/worker.py:368-375) and, if the node stays open, at once the
// split + midpoint solve + child construction (lib/worker.py:403-414, 354-365), allocates the
// two child records from the node pool and pushes them.  No sweep barriers: subtrees progress
// independently, the narrow top and bottom levels of the tree overlap with the bulk, and the
// workgroup's copy of the constant LP block stays in LDS for the whole run.
//   * queue slot k is written once (-1 = not yet): a consumer that drew a slot ahead of the
//     tail waits for it (s_sleep) or leaves when `pending` (nodes pushed, not yet completed)
//     reaches 0.  Producers never wait for consumers, and a waiting wavefront holds nothing
//     another one needs, so the kernel cannot deadlock whatever the residency of the grid;
//   * MI355X has one L2 per XCD: child records and structure words are written through to the
//     device coherence point (agent-scope atomic stores), the wavefront waits for them to
//     complete, then stores the queue slots; the consumer reads the slot with an agent-scope
//     atomic load and invalidates its L1 / non-local L2 lines (acquire fence) before reading
//     the record.  No L2 write-back anywhere;
//   * node ids follow the allocation order and differ from run to run; the TREE does not
//     (a node's fate depends on its own record only).  ehm_tree_export relabels to the
//     breadth-first order of the level-synchronous engine.
#define EHM_PERSIST_WATCHDOG_TICKS (60LL * 100000000LL)    // 60 s of the 100 MHz wall clock
#ifndef EHM_PERSIST_MIDFIRST
#define EHM_PERSIST_MIDFIRST 0
#endif
__global__ __launch_bounds__(EHM_K2_THREADS) void k2_persist(
    DevProblem P, DevTree T, int32_t* slots, int n_slots, PersistCtl* ctl, int node_cap,
    DevCounters* cnt, int wave_doubles, int sign_only, int max_depth, PersistDeal deal) {
    K2_PROLOGUE();
    const int p = P.p, n_u = P.n_u;
    const int nrec = rec_doubles(p, n_u);
    load_shared(P, 0, sm, tid, blockDim.x);
    __syncthreads();
    const long long t_start = wall_clock64();
    // statistics are kept per wavefront (in LDS: registers are what this kernel is short of) and
    // added to the global counters ONCE, when it leaves -- the level-synchronous kernels pay ~8
    // device atomics per node for them
    unsigned long long* wst = reinterpret_cast<unsigned long long*>(nb.aug);
    double* wmargin = nb.aug + 16;
    enum { W_SOLVES = 0, W_ITERS, W_STALLED, W_ERRORS, W_SLACK, W_SLACK_ITERS, W_CLOSED, W_SPLITS,
           W_DEPTH, W_TRUNC, W_CERT, W_WIT, W_ROUTED, W_RCLOSED, W_RSPLITS, W_RSOLVES,
           W_INH = 17, W_MT = 18, W_MTPARK = 19,           // slot 16 is *wmargin
           W_TQ = 20, W_TMT = 21, W_TMID = 22, W_TSLK = 23, W_NMT = 24,     // DevCounters::prof
           W_WITT = 25, W_TPRE = 26, W_TPOST = 27, W_REQ = 28 };
    if (lane0 < 29 && lane0 != 16) wst[lane0] = 0ULL;
    if (lane0 == 0) *wmargin = 1e300;
    wsync();
    int keep = -1;          // the child this wavefront goes on with (see "work first" below)
    // hot: everything the kept child's visit reads first is still in LDS -- its record in nb.rec,
    // its vertex gradients at nb.lp + K2_HOT_GRAD, its witness in the stash -- and its depth /
    // commutation in registers: the visit starts without a single global round trip (the sibling's
    // consumer pays those; round 3 re-read all of it, four dependent loads per kept child)
    bool hot = false;
    int hot_dep = 0, hot_d = 0;
    const bool keep_child = deal.keep != 0;
    for (;;) {
        int id = -1;
        const bool is_hot = hot && keep >= 0;
        hot = false;
        if (keep >= 0) {
            id = keep;
            keep = -1;
        } else if (lane0 == 0) {
            const int idx = atomicAdd(&ctl->head, 1);
            if (idx < ((deal.pop_limit > 0 && deal.pop_limit < n_slots) ? deal.pop_limit
                                                                        : n_slots)) {
                long long t_q = 0;
                for (;;) {
                    id = __hip_atomic_load(&slots[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (id >= 0) break;
                    if (!t_q) t_q = wall_clock64();
                    if (__hip_atomic_load(&ctl->pending, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) <= 0 ||
                        __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) != 0)
                        break;
                    if (wall_clock64() - t_start > EHM_PERSIST_WATCHDOG_TICKS) {
                        atomicMax(&ctl->abort, 3);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(64);
                }
                if (t_q) wst[W_TQ] += (unsigned long long)(wall_clock64() - t_q);
            }
        }
        id = __builtin_amdgcn_readfirstlane(id);
        if (id < 0) break;
        // a node that was put back once (its midpoint was being solved elsewhere) carries a mark
        const bool came_back = (id & EHM_REQUEUED) != 0;
        id &= ~EHM_REQUEUED;
        const long long t_pre = wall_clock64();
        const int lane = pin(lane0);
        double* node = nb.rec;
        double* hgrad = nb.lp + K2_HOT_GRAD;        // vertex gradients of the node, (p+1) p doubles
        const int ng = (p + 1) * p;
        int dep = hot_dep;
        if (!is_hot) {
            // acquire (L1 / non-local L2 invalidate): the record behind the slot is visible
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const double* rec = T.rec + (size_t)id * T.rec_stride;
            // record, gradients, witness and depth are independent loads: one wait for all of them
            const double g_in = (T.grad && lane < ng) ? T.grad[(size_t)id * ng + lane] : 0.0;
#if EHM_PERSIST_MIDFIRST
            const double w_in = (T.wit && sign_only && lane < p + 2)
                ? T.wit[(size_t)id * (p + 2) + lane] : 0.0;
#endif
            dep = T.depth[id];
            for (int k = lane; k < nrec; k += 64) node[k] = rec[k];
            if (T.grad) {
                if (lane < ng) hgrad[lane] = g_in;
                for (int k = lane + 64; k < ng; k += 64) hgrad[k] = T.grad[(size_t)id * ng + k];
            }
#if EHM_PERSIST_MIDFIRST
            if (T.wit && sign_only && lane < p + 2)
                nb.rec[(size_t)wave_doubles - k2_stash_doubles(p, n_u) + n_u + p + lane] = w_in;
#endif
        }
        wsync();
        // ---- suboptimality test --------------------------------------------------------------
        // the inherited witness first (a dozen instructions): a node it proves open (t* > 0) cannot
        // be closed by the tangent-plane bound (t* < 0), whose 45 pairs of planes are then skipped
        bool inh_open = false;
        double inh_tw = 0.0;
#if EHM_PERSIST_MIDFIRST
        if (T.wit && sign_only) {
            const double* wit_ = nb.rec + (size_t)wave_doubles - k2_stash_doubles(p, n_u) + n_u + p;
            const double* Vc = node + rec_off_vcost(p);
            double vbw = 0.0;
            for (int q = 0; q <= p; ++q) vbw = fma(wit_[1 + q], Vc[q], vbw);
            const double cw = wit_[0];
            const double tw = fmin(vbw - cw - P.eps_a, vbw - (1.0 + P.eps_r) * cw);
            inh_open = tw > EHM_ROUTE_TOL * (1.0 + fabs(vbw));
            inh_tw = tw;
        }
#endif
        if (T.grad && sign_only && inh_open && deal.check) {
            // option "check_witness": the witness says open -- the bound must not say closed
            const double thr = -EHM_ROUTE_TOL * (1.0 + fabs(node[rec_off_vcost(p)]));
            const double bnd = cut_bound(node, hgrad, p, P.eps_a, P.eps_r, lane, nb.lp, thr);
            if (bnd < thr && lane == 0) wst[W_ERRORS] += 1;
            wsync();
        }
        if (T.grad && sign_only && !inh_open) {
            // tangent-plane bound of t* (ehm_dev.h, cut_bound): negative => closed, no LP
            const double thr = -EHM_ROUTE_TOL * (1.0 + fabs(node[rec_off_vcost(p)]));
            const double bnd = cut_bound(node, hgrad, p, P.eps_a, P.eps_r, lane, nb.lp, thr);
            if (bnd < thr) {
                if (lane == 0) {
                    const int dep0 = dep;
                    wst[W_CERT] += 1;
                    wst[W_CLOSED] += 1;
                    if (dep0 < deal.depth) wst[W_RCLOSED] += 1;
                    *wmargin = fmin(*wmargin, -bnd);
                    if ((unsigned long long)dep0 > wst[W_DEPTH]) wst[W_DEPTH] = (unsigned long long)dep0;
                    T.tstar[id] = bnd;
                    // (a kept child's flags are the ones it was created with a moment ago: 2)
                    if (is_hot) T.flags[id] = 3;
                    else T.flags[id] |= 1;
                    atomicSub(&ctl->pending, 1);
                }
                wsync();
                continue;
            }
        }
#if EHM_PERSIST_MIDFIRST
        // Midpoint first (-DEHM_PERSIST_MIDFIRST=1, how every instance is built since round 2;
        // validated on the device: identical tree): after the tangent-plane bound has taken out 97 %
        // of the closed leaves, almost every node that reaches an LP is open and needs its midpoint
        // solve anyway.  Doing that solve FIRST gives a witness: at theta = mid the interpolated
        // cost is (V_bi + V_bj)/2 and the optimal cost is the solve's optimum, so
        //     t_mid = min( Vbar - J_mid - eps_a , Vbar - (1 + eps_r) J_mid ) <= t* ,
        // and t_mid > 0 proves the node open without its suboptimality-test LP (42 % of the open
        // nodes of the bench tree, tools/midpoint_certificate.py).  Otherwise the LP decides as
        // before; a node it closes has paid for a midpoint solve it did not need (3 % of them).
        const bool can_split = !(max_depth > 0 && dep >= max_depth);
        double* mid = nb.th;
        // the last k2_stash_doubles of the wave's LDS: midpoint input, midpoint gradient,
        // the witness handed on to the children (DevTree::wit)
        const int st_g = n_u, st_w = n_u + p;       // layout: k2_stash_doubles (ehm_k2.h)
        double* stash = nb.rec + (size_t)wave_doubles - k2_stash_doubles(p, n_u);
        double* wit = stash + st_w;
        bool have_wit = false;
        int mt_res = MT_NONE;       // table of midpoint optima (ehm_midtable.h)
        int bi = 0, bj = 1;
        int its = 0;
        double Jm = 0.0;
        int mid_status = 1, mid_iters = 0;
        bool mid_conv = false;
        bool open = false;
        double tst = 0.0, margin = 0.0;
        bool decided = false;
        if (T.wit && sign_only) {
            // inherited witness: the point that proved an ancestor open, if it lies in this node
            // (loaded with the record, or left in the stash by the parent's visit; evaluated above)
            if (inh_open) {
                open = true;
                decided = true;
                have_wit = true;
                tst = inh_tw;
                margin = inh_tw;
                if (lane == 0) wst[W_INH] += 1;
            }
        }
        if (can_split) {
            longest_edge_wave(node, p, lane, bi, bj);
            if (lane < p) {
#pragma clang fp contract(off)
                mid[lane] = (node[bi * p + lane] + node[bj * p + lane]) / 2.0;
            }
            wsync();
            if (T.mt.state) {
                // the simplices around an edge all ask for this midpoint: solve it once
                unsigned int mt_i = 0u;
                int mt_slot = 0;
                const unsigned long long mt_tg = mt_tag(mid, p, T.mt.mask, &mt_i);
                if (lane == 0) {
                    wst[W_TPRE] += (unsigned long long)(wall_clock64() - t_pre);
                    long long waited = 0;
                    // A node whose fate is known (inherited witness) needs nothing but this
                    // optimum: if another wavefront is solving it right now, the node goes back
                    // into the queue -- once -- and this wavefront takes another one instead of
                    // sleeping through the solve.  (Not in budgeted launches: their left-over
                    // must stay the contiguous slice behind the pop limit.)
                    const bool may_requeue = decided && !came_back && deal.pop_limit <= 0 &&
                                             id < EHM_REQUEUED;
                    mt_res = mt_claim(T.mt, mt_tg, mt_i, t_start, EHM_PERSIST_WATCHDOG_TICKS,
                                      &mt_slot, &waited, may_requeue);
                    if (mt_res == MT_BUSY) {
                        const int t = atomicAdd(&ctl->tail, 1);
                        if (t < n_slots) {
                            __hip_atomic_store(&slots[t], id | EHM_REQUEUED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                            wst[W_REQ] += 1;
                        } else {            // no room behind the tail: wait after all
                            mt_res = mt_claim(T.mt, mt_tg, mt_i, t_start,
                                              EHM_PERSIST_WATCHDOG_TICKS, &mt_slot, &waited);
                        }
                    }
                    if (waited) {
                        wst[W_TMT] += (unsigned long long)waited;
                        wst[W_NMT] += 1;
                    }
                }
                mt_res = __builtin_amdgcn_readfirstlane(mt_res);
                mt_slot = __builtin_amdgcn_readfirstlane(mt_slot);
                if (mt_res == MT_BUSY) {    // put back: still pending, somebody else's visit
                    wsync();
                    continue;
                }
                if (mt_res == MT_HIT) {
                    bool same = false;
                    const double ev = mt_read(T.mt, mt_slot, lane, mid, p, &same);
                    if (!same) {
                        mt_res = MT_NONE;               // another midpoint with this tag
                    } else {
                        // entry layout: ehm_midtable.h
                        const int word = (int)__shfl(ev, 9);
                        Jm = __shfl(ev, 8);
                        mid_status = word & 0xff;
                        mid_conv = ((word >> 8) & 1) != 0;
                        mid_iters = 0;                  // no iterations were spent here
                        if (lane >= 10 && lane < 10 + n_u) stash[lane - 10] = ev;
                        if (T.grad && lane >= 18 && lane < 18 + p) stash[st_g + lane - 18] = ev;
                        if (lane == 0) wst[W_MT] += 1;
                    }
                }
                // what the code after the solve needs, parked in LDS: registers are what this
                // kernel is short of while it solves
                if (lane == 0) wst[W_MTPARK] = (unsigned long long)mt_res |
                                               ((unsigned long long)mt_slot << 2);
            }
            if (mt_res != MT_HIT)
            {
            const long long t_mid = wall_clock64();
            Wave Wm;
            IpmResult rm;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
                double b[SLOTS];
                const int ln = pin(lane);
                assemble_point(S, Wm, nb.lp, mid, false, b, ln, P, 0);
                rm = ipm_solve(S, Wm, b, ln, false, step_fraction(attempt), T.grad ? nb.F : nullptr);
                its += rm.iters;
                if (rm.status == 0) break;
            }
#if EHM2_QUAD
            if (T.grad) quad_grad_add(Wm, P, 0, mid, nb.F, lane);
#endif
            Jm = rm.obj;
            mid_status = rm.status;
            mid_conv = (rm.status == 0) && (rm.merit <= 1.0);    // not merely "accepted"
            mid_iters = its;
            if (lane < n_u) stash[lane] = Wm.xb[lane];
            if (lane == 0) wst[W_TMID] += (unsigned long long)(wall_clock64() - t_mid);
            }
            if (mt_res != MT_HIT && T.grad && lane < p) stash[st_g + lane] = nb.F[lane];
            wsync();
            if (T.mt.state) {
                const unsigned long long park = wst[W_MTPARK];
                mt_res = (int)(park & 3ull);
                if (mt_res == MT_OWN) {
                    unsigned int mt_i = 0u;
                    const unsigned long long mt_tg = mt_tag(mid, p, T.mt.mask, &mt_i);
                    mt_publish(T.mt, (int)(park >> 2), mt_tg, lane, mid, p, Jm, mid_status,
                               mid_conv ? 1 : 0, mid_iters, stash, n_u,
                               T.grad ? stash + st_g : nullptr);
                }
            }
            if (sign_only && mid_conv && !decided) {
                const double* Vc = node + rec_off_vcost(p);
                const double vb = 0.5 * (Vc[bi] + Vc[bj]);
                const double tw = fmin(vb - Jm - P.eps_a, vb - (1.0 + P.eps_r) * Jm);
                if (tw > EHM_ROUTE_TOL * (1.0 + fabs(vb))) {
                    open = true;
                    decided = true;
                    tst = tw;
                    margin = tw;
                    if (lane == 0) wst[W_WIT] += 1;
                }
            }
        }
#if !EHM2_QUAD
        if (!decided && T.mt.state && sign_only) {
            // The OTHER edges' midpoints.  A neighbour that has bisected one of this simplex's
            // edges left the optimal cost at that edge's midpoint in the table; like the node's
            // own midpoint it is a candidate witness -- V*(mid') known, interpolated cost
            // (V_a + V_b)/2 -- and, unlike it, a point that stays in the interior of an edge of the
            // children, so they inherit it.  Lane e looks edge e up (read-only, nobody waits).
            const double* Vc = node + rec_off_vcost(p);
            double tw_l = -1e300, J_l = 0.0;
            int ea = 0, eb = 1, slot_l = -1;
            const int n_edges = (p + 1) * p / 2;
            // lane e's midpoint: 8 doubles of the wavefront's LP workspace (free between solves)
            double* em = nb.lp + 8 * (lane < n_edges ? lane : 0);
            if (lane < n_edges) {
                int rem = lane;
                while (rem >= p - ea) { rem -= (p - ea); ++ea; }
                eb = ea + 1 + rem;
                if (!(ea == bi && eb == bj)) {
                    {
#pragma clang fp contract(off)
                        for (int k = 0; k < p; ++k) em[k] = (node[ea * p + k] + node[eb * p + k]) / 2.0;
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
                const unsigned long long won = __builtin_amdgcn_ballot_w64(tw_l > -1e299);
                if (won != 0ull) {
                    // first edge (enumeration order) whose midpoint proves the node open
                    const int src = __builtin_ctzll(won);
                    const double tw = __shfl(tw_l, src);
                    const double Jw = __shfl(J_l, src);
                    const int wa = __shfl(ea, src), wb = __shfl(eb, src);
                    open = true;
                    decided = true;
                    tst = tw;
                    margin = tw;
                    if (T.wit) {
                        if (lane == 0) wit[0] = Jw;
                        if (lane <= p) wit[1 + lane] = (lane == wa || lane == wb) ? 0.5 : 0.0;
                        have_wit = true;
                    }
                    if (lane == 0) wst[W_WITT] += 1;
                    wsync();
                }
            }
        }
#endif
        if (!decided) {
            const long long t_slk = wall_clock64();
            Wave W;
            IpmResult r;
            its = 0;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
                double b[SLOTS];
                const int ln = pin(lane);
                assemble_simplex(S, W, nb, node, node + rec_off_vcost(p), SX_SLACK, P.eps_a, P.eps_r,
                                 b, ln, P, 0);
                r = ipm_solve(S, W, b, ln, sign_only != 0, step_fraction(attempt));
                its += r.iters;
                if (r.status == 0) break;
            }
            tst = -r.obj;
            open = (tst >= 0.0);
            margin = r.margin;
            const int slack_status = r.status;
#if !EHM2_QUAD
            if (T.wit && sign_only && open && slack_status == 0) {
                // this node's own witness: the accepted iterate's parameter (barycentric) and the
                // cost of its z, raised by the safety amount (EHM_WIT_REL, ehm_dev.h)
                const int nz = S.n;
                double cz = 0.0, sb = 0.0;
                for (int q = 0; q < nz; ++q) cz = fma(S.cv[q], W.xb[zcol(W, S, q)], cz);
                for (int q = 0; q < p; ++q) sb += W.xb[W.psi0 + q];
                wsync();
                if (lane == 0) {
                    wit[0] = fma(EHM_WIT_REL, margin, cz);
                    wit[1] = 1.0 - sb;
                }
                if (lane < p) wit[2 + lane] = W.xb[W.psi0 + lane];
                have_wit = true;
                wsync();
            }
#endif
            if (lane == 0) {
                wst[W_TSLK] += (unsigned long long)(wall_clock64() - t_slk);
                wst[W_SOLVES] += 1;
                if (dep < deal.depth) wst[W_RSOLVES] += 1;
                wst[W_ITERS] += (unsigned long long)its;
                wst[W_SLACK] += 1;
                wst[W_SLACK_ITERS] += (unsigned long long)its;
                if (slack_status != 0) {
                    wst[W_STALLED] += 1;
                    wst[W_ERRORS] += 1;
                    T.flags[id] |= 8;
                }
            }
        }
        if (lane == 0) {
            if (can_split) {        // the midpoint solve was done, whatever became of the node
                if (mt_res != MT_HIT) {     // ... by this wavefront (else: taken from the table)
                    wst[W_SOLVES] += 1;
                    if (dep < deal.depth) wst[W_RSOLVES] += 1;
                    wst[W_ITERS] += (unsigned long long)mid_iters;
                }
                if (mid_status != 0 && open) {
                    wst[W_STALLED] += 1;
                    wst[W_ERRORS] += 1;
                    T.flags[id] |= 16;
                }
            }
            *wmargin = fmin(*wmargin, margin);
            if (margin < EHM_ROUTE_TOL * (1.0 + fabs(node[rec_off_vcost(p)]))) wst[W_ROUTED] += 1;
            if ((unsigned long long)dep > wst[W_DEPTH]) wst[W_DEPTH] = (unsigned long long)dep;
            T.tstar[id] = tst;
            if (!open) {
                T.flags[id] |= 1;
                wst[W_CLOSED] += 1;
                if (dep < deal.depth) wst[W_RCLOSED] += 1;
            }
        }
        if (!open) {
            if (lane == 0) atomicSub(&ctl->pending, 1);
            wsync();
            continue;
        }
        if (!can_split) {
            if (lane == 0) {
                wst[W_TRUNC] = 1;
                atomicSub(&ctl->pending, 1);
            }
            wsync();
            continue;
        }
        // ---- children (the midpoint solve is in Jm / stash) --------------------------------------
        const long long t_post = wall_clock64();
        // the wavefront goes on with child 1 itself (work first, below) unless the children are
        // dealt over ranks at this depth: then its record, gradients and witness also stay in LDS
        const bool fast = keep_child && !(deal.world > 1 && dep + 1 == deal.depth);
        // the node's own gradients again (the solves have used the workspace they were staged in):
        // the load is under way while the allocation below makes its round trip
        const double gv0 = (T.grad && lane < ng) ? T.grad[(size_t)id * ng + lane] : 0.0;
        const int d = is_hot ? hot_d : T.didx[id];
        int c0 = 0;
        if (lane == 0) c0 = atomicAdd(&ctl->n_nodes, 2);
        c0 = __builtin_amdgcn_readfirstlane(c0);
        if (c0 + 2 > node_cap) {
            if (lane == 0) {
                atomicMax(&ctl->abort, 1);
                atomicSub(&ctl->pending, 1);
            }
            break;
        }
        struct { double obj; } r = {Jm};      // the child-record loop below reads r.obj
        if (T.grad) {       // children's vertex gradients, written through like the records
            const double* gp_ = T.grad + (size_t)id * ng;
            double* g0 = T.grad + (size_t)c0 * ng;
            for (int k = lane; k < ng; k += 64) {
                const double gv = (k < 64) ? gv0 : gp_[k];
                const double a0 = (k >= bi * p && k < bi * p + p) ? stash[st_g + k - bi * p] : gv;
                const double a1 = (k >= bj * p && k < bj * p + p) ? stash[st_g + k - bj * p] : gv;
                __hip_atomic_store(g0 + k, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g0 + ng + k, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fast) hgrad[k] = a1;
            }
        }
        if (T.wit && lane < p + 2) {
            // the witness goes to the child that contains it: child 0 (vertex bi -> midpoint) iff
            // alpha_bj >= alpha_bi; theta_w = 2 a_i mid + (a_j - a_i) v_j + ... there.  The OTHER
            // child gets the point where the segment from theta_w to the parent's vertex on its
            // side meets the shared face, with the cost bound (1 - mu) c_w + mu V_vertex (a convex
            // combination of two feasible decision vectors is feasible, the cost is linear)
            double* w0 = T.wit + (size_t)c0 * (p + 2);
            double v0 = 0.0, v1 = 0.0;
            if (have_wit) {
                witness_for_children(wit, node + rec_off_vcost(p), bi, bj, lane, v0, v1);
            }
            __hip_atomic_store(w0 + lane, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(w0 + (p + 2) + lane, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wsync();                        // every lane has read the parent's witness
            if (fast) wit[lane] = v1;       // child 1's, for its visit by this wavefront
        }
        if (lane == 0) {
            wst[W_SPLITS] += 1;
            if (dep < deal.depth) wst[W_RSPLITS] += 1;
        }
        const double* xmid = stash;
#else
        Wave W;
        IpmResult r;
        int its = 0;
        for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
            double b[SLOTS];
            const int ln = pin(lane);
            assemble_simplex(S, W, nb, node, node + rec_off_vcost(p), SX_SLACK, P.eps_a, P.eps_r,
                             b, ln, P, 0);
            r = ipm_solve(S, W, b, ln, sign_only != 0, step_fraction(attempt));
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;
        const double tst = -r.obj;
        const bool open = (tst >= 0.0);
        if (lane == 0) {
            wst[W_SOLVES] += 1;
            if (dep < deal.depth) wst[W_RSOLVES] += 1;
            wst[W_ITERS] += (unsigned long long)r.iters;
            wst[W_SLACK] += 1;
            wst[W_SLACK_ITERS] += (unsigned long long)r.iters;
            if (r.status != 0) {
                wst[W_STALLED] += 1;
                wst[W_ERRORS] += 1;
                T.flags[id] |= 8;
            }
            *wmargin = fmin(*wmargin, r.margin);
            if (r.margin < EHM_ROUTE_TOL * (1.0 + fabs(node[rec_off_vcost(p)]))) wst[W_ROUTED] += 1;
            if ((unsigned long long)dep > wst[W_DEPTH]) wst[W_DEPTH] = (unsigned long long)dep;
            T.tstar[id] = tst;
            if (!open) {
                T.flags[id] |= 1;
                wst[W_CLOSED] += 1;
                if (dep < deal.depth) wst[W_RCLOSED] += 1;
            }
        }
        if (!open) {
            if (lane == 0) atomicSub(&ctl->pending, 1);
            wsync();
            continue;
        }
        if (max_depth > 0 && dep >= max_depth) {
            if (lane == 0) {
                wst[W_TRUNC] = 1;
                atomicSub(&ctl->pending, 1);
            }
            wsync();
            continue;
        }
        // ---- split, midpoint solve, children -------------------------------------------------
        int c0 = 0;
        if (lane == 0) c0 = atomicAdd(&ctl->n_nodes, 2);
        c0 = __builtin_amdgcn_readfirstlane(c0);
        if (c0 + 2 > node_cap) {
            if (lane == 0) {
                atomicMax(&ctl->abort, 1);
                atomicSub(&ctl->pending, 1);
            }
            break;
        }
        double* mid = nb.th;
        int bi, bj;
        longest_edge_wave(node, p, lane, bi, bj);
        if (lane < p) {
#pragma clang fp contract(off)
            mid[lane] = (node[bi * p + lane] + node[bj * p + lane]) / 2.0;
        }
        wsync();
        const int d = T.didx[id];
        its = 0;
        for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
            double b[SLOTS];
            const int ln = pin(lane);
            assemble_point(S, W, nb.lp, mid, false, b, ln, P, 0);
            r = ipm_solve(S, W, b, ln, false, step_fraction(attempt), T.grad ? nb.F : nullptr);
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;
#if EHM2_QUAD
        if (T.grad) quad_grad_add(W, P, 0, mid, nb.F, lane);
#endif
        if (T.grad) {       // children's vertex gradients, written through like the records
            const int ng = (p + 1) * p;
            const double* gp_ = T.grad + (size_t)id * ng;
            double* g0 = T.grad + (size_t)c0 * ng;
            for (int k = lane; k < ng; k += 64) {
                const double gv = gp_[k];
                const double a0 = (k >= bi * p && k < bi * p + p) ? nb.F[k - bi * p] : gv;
                const double a1 = (k >= bj * p && k < bj * p + p) ? nb.F[k - bj * p] : gv;
                __hip_atomic_store(g0 + k, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g0 + ng + k, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (lane == 0) {
            wst[W_SOLVES] += 1;
            wst[W_ITERS] += (unsigned long long)r.iters;
            wst[W_SPLITS] += 1;
            if (dep < deal.depth) {
                wst[W_RSOLVES] += 1;
                wst[W_RSPLITS] += 1;
            }
            if (r.status != 0) {
                wst[W_STALLED] += 1;
                wst[W_ERRORS] += 1;
                T.flags[id] |= 16;
            }
        }
        const long long t_post = wall_clock64();
        const double* xmid = W.xb;
#endif
        double* rec0 = T.rec + (size_t)c0 * T.rec_stride;
        double* rec1 = rec0 + T.rec_stride;
        const int ov = rec_off_vcost(p), ou = rec_off_vinput(p);
        for (int k = lane; k < nrec; k += 64) {
            double v0 = node[k], v1 = node[k];
            if (k < ov) {
                if (k >= bi * p && k < bi * p + p) v0 = mid[k - bi * p];
                if (k >= bj * p && k < bj * p + p) v1 = mid[k - bj * p];
            } else if (k < ou) {
                if (k - ov == bi) v0 = r.obj;
                if (k - ov == bj) v1 = r.obj;
            } else {
                const int q = k - ou;
                if (q >= bi * n_u && q < bi * n_u + n_u) v0 = xmid[q - bi * n_u];
                if (q >= bj * n_u && q < bj * n_u + n_u) v1 = xmid[q - bj * n_u];
            }
            // everything a child's consumer reads or later overwrites is written THROUGH to the
            // device coherence point (agent-scope atomic stores): visible to the other XCDs
            // without writing this XCD's whole L2 back (a device-scope release fence would --
            // measured: 47 GB of write-backs per partition, mostly register spills), and no
            // dirty copy stays behind that could later clobber the consumer's own writes
            __hip_atomic_store(rec0 + k, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rec1 + k, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if EHM_PERSIST_MIDFIRST
            if (fast) node[k] = v1;         // child 1's record, in place (entry k depends on entry k)
#endif
        }
        // sharded launch: the children created at the deal depth go to rank (path code % world)
        int own0 = 1, own1 = 1;
        if (lane == 0) {
            T.left[id] = c0;
#define EHM_WT(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
            if (T.code) {
                const uint32_t pc = T.code[id];
                const uint32_t code0 = 2u * pc, code1 = 2u * pc + 1u;
                EHM_WT(&T.code[c0], code0);
                EHM_WT(&T.code[c0 + 1], code1);
                if (deal.world > 1 && dep + 1 == deal.depth) {
                    const uint32_t h0 = deal.mix ? ((code0 * 2654435761u) >> 12) : code0;
                    const uint32_t h1 = deal.mix ? ((code1 * 2654435761u) >> 12) : code1;
                    own0 = (int)(h0 % (uint32_t)deal.world) == deal.rank;
                    own1 = (int)(h1 % (uint32_t)deal.world) == deal.rank;
                }
            }
            EHM_WT(&T.left[c0], -1);
            EHM_WT(&T.left[c0 + 1], -1);
            EHM_WT(&T.didx[c0], d);
            EHM_WT(&T.didx[c0 + 1], d);
            EHM_WT(&T.depth[c0], dep + 1);
            EHM_WT(&T.depth[c0 + 1], dep + 1);
            EHM_WT(&T.flags[c0], (uint8_t)(own0 ? 2 : 6));
            EHM_WT(&T.flags[c0 + 1], (uint8_t)(own1 ? 2 : 6));
            EHM_WT(&T.tstar[c0], 0.0);
            EHM_WT(&T.tstar[c0 + 1], 0.0);
#undef EHM_WT
        }
        // the write-through stores above have completed (s_waitcnt) before the slots go out
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        int kept = -1;
        if (lane == 0) {
            // Work first: the wavefront goes on with ONE of the children itself (its record is
            // hot, no queue round trip, and the deep chains that end a partition are followed at
            // once instead of waiting behind the whole frontier at every level); the other child
            // feeds the queue.  Budgeted launches (pop_limit) queue BOTH children by default
            // (PersistDeal::keep = 0: persistent_run sets it unless option "budget_keep" is on):
            // keeping one there was built and measured in round 5 -- the kept chains run depth
            // first, what the launch leaves behind the pop limit is then made of deep small cells
            // and the rebalancing rounds move 4x the nodes (38.2 against 33.4 ms for two ranks) --
            // and stays off; with budget_keep = 1 a kept child never enters the queue, so what the
            // launch leaves is still the contiguous slice behind the pop limit.
            const int nown = own0 + own1;
            int push0 = own0, push1 = own1;
            if (keep_child) {
                if (own1) { kept = c0 + 1; push1 = 0; }
                else if (own0) { kept = c0; push0 = 0; }
            }
            const int npush = push0 + push1;
            const int t = npush ? atomicAdd(&ctl->tail, npush) : 0;
            if (t + npush <= n_slots) {
                int at = t;
                if (push0)
                    __hip_atomic_store(&slots[at++], c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (push1)
                    __hip_atomic_store(&slots[at], c0 + 1, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                // -1 (this node) + its children, the queued one and the kept one alike
                if (nown != 1) atomicAdd(&ctl->pending, nown - 1);
            } else {
                kept = -1;
                atomicMax(&ctl->abort, 1);
                atomicSub(&ctl->pending, 1);
            }
            wst[W_TPOST] += (unsigned long long)(wall_clock64() - t_post);
        }
        keep = __builtin_amdgcn_readfirstlane(kept);
#if EHM_PERSIST_MIDFIRST
        hot = fast && keep >= 0;
        hot_dep = dep + 1;
        hot_d = d;
#endif
        wsync();
    }
    wsync();
    if (lane0 == 0) {
        atomicAdd(&cnt->lp_solves, wst[W_SOLVES]);
        atomicAdd(&cnt->ipm_iters, wst[W_ITERS]);
        if (wst[W_STALLED]) atomicAdd(&cnt->stalled, wst[W_STALLED]);
        if (wst[W_ERRORS]) atomicAdd(&cnt->errors, wst[W_ERRORS]);
        atomicAdd(&cnt->slack_solves, wst[W_SLACK]);
        atomicAdd(&cnt->slack_iters, wst[W_SLACK_ITERS]);
        atomicMin(&cnt->min_margin_bits, (unsigned long long)__double_as_longlong(*wmargin));
        if (wst[W_CERT]) atomicAdd(&cnt->cert_closed, wst[W_CERT]);
        if (wst[W_WIT]) atomicAdd(&cnt->wit_open, wst[W_WIT]);
        if (wst[W_INH]) atomicAdd(&cnt->wit_inherited, wst[W_INH]);
        if (wst[W_MT]) atomicAdd(&cnt->mid_shared, wst[W_MT]);
        if (wst[W_WITT]) atomicAdd(&cnt->wit_table, wst[W_WITT]);
        atomicAdd(&cnt->prof[0], (unsigned long long)(wall_clock64() - t_start));
        if (wst[W_TQ]) atomicAdd(&cnt->prof[1], wst[W_TQ]);
        if (wst[W_TMT]) atomicAdd(&cnt->prof[2], wst[W_TMT]);
        if (wst[W_TMID]) atomicAdd(&cnt->prof[3], wst[W_TMID]);
        if (wst[W_TSLK]) atomicAdd(&cnt->prof[4], wst[W_TSLK]);
        if (wst[W_NMT]) atomicAdd(&cnt->prof[5], wst[W_NMT]);
        if (wst[W_TPRE]) atomicAdd(&cnt->prof[6], wst[W_TPRE]);
        if (wst[W_TPOST]) atomicAdd(&cnt->prof[7], wst[W_TPOST]);
        if (wst[W_REQ]) atomicAdd(&cnt->prof[8], wst[W_REQ]);
        if (wst[W_ROUTED]) atomicAdd(&cnt->routed, wst[W_ROUTED]);
        atomicAdd(&ctl->closed, wst[W_CLOSED]);
        atomicAdd(&ctl->splits, wst[W_SPLITS]);
        if (wst[W_RCLOSED]) atomicAdd(&ctl->repl_closed, wst[W_RCLOSED]);
        if (wst[W_RSPLITS]) atomicAdd(&ctl->repl_splits, wst[W_RSPLITS]);
        if (wst[W_RSOLVES]) atomicAdd(&ctl->repl_solves, wst[W_RSOLVES]);
        atomicMax(&ctl->max_depth_seen, (int)wst[W_DEPTH]);
        if (wst[W_TRUNC]) atomicMax(&ctl->truncated, 1);
    }
}

// ---- vertex solves that seed a node's costs / inputs (lib/oracle.py:416-443) ---------------
__global__ __launch_bounds__(EHM_K2_THREADS) void k2_vertex_solve(
    DevProblem P, DevTree T, const int32_t* __restrict__ nodes, int n_nodes, DevCounters* cnt,
    int wave_doubles) {
    K2_PROLOGUE();
    const int p = P.p, n_u = P.n_u;
    const int total = n_nodes * (p + 1);
    const int per = (total + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per;
    const int hi = (lo + per < total) ? lo + per : total;
    if (lo >= hi) return;
    load_shared(P, 0, sm, tid, blockDim.x);
    if (tid == 0) s_ctr = lo;
    __syncthreads();
    for (;;) {
        const int t = pull(&s_ctr, lane0);
        if (t >= hi) break;
        const int lane = pin(lane0);
        const int id = nodes[t / (p + 1)];
        const int v = t % (p + 1);
        double* rec = T.rec + (size_t)id * T.rec_stride;
        if (lane < p) nb.th[lane] = rec[v * p + lane];
        wsync();
        Wave W;
        IpmResult r;
        int its = 0;
        for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
            double b[SLOTS];
            const int ln = pin(lane);
            assemble_point(S, W, nb.lp, nb.th, false, b, ln, P, 0);
            r = ipm_solve(S, W, b, ln, false, step_fraction(attempt), T.grad ? nb.F : nullptr);
            its += r.iters;
            if (r.status == 0) break;
        }
        r.iters = its;
        count_solve(cnt, r, lane);
        if (r.status != 0 && lane == 0) atomicAdd(&cnt->errors, 1ULL);
#if EHM2_QUAD
        if (T.grad) quad_grad_add(W, P, 0, nb.th, nb.F, lane);
#endif
        if (T.grad && lane < p) T.grad[((size_t)id * (p + 1) + v) * p + lane] = nb.F[lane];
        if (lane == 0) rec[rec_off_vcost(p) + v] = r.obj;
        if (lane < n_u) rec[rec_off_vinput(p) + v * n_u + lane] = W.xb[lane];
        wsync();
    }
}

// ---- self test of the wave primitives (ehm_selftest) --------------------------------------
__global__ __launch_bounds__(64) void k2_selftest(double* out) {
    const int lane = threadIdx.x;
    const double v = 1.0 + 0.5 * lane;                       // sum = 64 + 0.5*2016 = 1072
    out[0] = wave_sum(v);
    out[1] = wave_max((lane == 37) ? 99.0 : -(double)lane);
    out[2] = wave_sum((lane < 25) ? 1.0 : 0.0);
    out[3] = frcp(3.0);
    out[4] = wave_max(-1.0 - lane);
}

}  // namespace EHM2_NS

// ---------------------------------------------------------------------------------------
// host-side launchers (one set per compiled instance)
// ---------------------------------------------------------------------------------------
namespace {

using namespace EHM2_NS;

hipError_t set_lds(int bytes) {
    const void* ks[] = {(const void*)k2_point_batch<0>, (const void*)k2_point_batch<1>,
                        (const void*)k2_simplex_batch<SX_MIN>,
                        (const void*)k2_simplex_batch<SX_SLACK>,
                        (const void*)k2_simplex_batch<SX_FEAS>,
                        (const void*)k2_lcss_decide, (const void*)k2_lcss_expand,
                        (const void*)k2_vertex_solve, (const void*)k2_persist};
    for (const void* k : ks) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t wave_doubles_for(const DevProblem& P, int n_lp, int ne, int persist) {
    // (+ k2_stash_doubles: what the midpoint-first flow of k2_persist parks per node behind the
    // wavefront's workspace; the batch and sweep kernels park nothing)
    return k2_node_doubles(P.p, P.n_u) + wave_lp_doubles(n_lp, ne, P.n - P.nd0, P.p) +
           ((EHM_PERSIST_MIDFIRST && persist) ? k2_stash_doubles(P.p, P.n_u) : 0);
}
size_t shared_doubles_for(const DevProblem& P) { return shared_doubles(P); }

void l_point(const K2Launch& L, DevProblem P, long long n_inst, const double* theta,
             const int32_t* seg, int feas, double* J, double* u0, int32_t* status,
             int32_t* iters, DevCounters* cnt, K2Gather G) {
    P.wc_lds = L.wc_lds;
    if (feas)
        hipLaunchKernelGGL(k2_point_batch<1>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, n_inst, theta, seg, J, u0, status, iters, cnt, L.wave_doubles, G);
    else
        hipLaunchKernelGGL(k2_point_batch<0>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream,
                           P, n_inst, theta, seg, J, u0, status, iters, cnt, L.wave_doubles, G);
}
void l_simplex(const K2Launch& L, DevProblem P, long long n_inst, const double* R,
               const double* Vbar, const int32_t* seg, int mode, double* obj, double* alpha,
               int32_t* status, int32_t* iters, DevCounters* cnt, K2Gather G) {
    P.wc_lds = L.wc_lds;
#define K2_SX_LAUNCH(M)                                                                      \
    hipLaunchKernelGGL(k2_simplex_batch<M>, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream, \
                       P, n_inst, R, Vbar, seg, obj, alpha, status, iters, cnt, L.wave_doubles, G)
    if (mode == SX_SLACK) K2_SX_LAUNCH(SX_SLACK);
    else if (mode == SX_FEAS) K2_SX_LAUNCH(SX_FEAS);
    else K2_SX_LAUNCH(SX_MIN);
#undef K2_SX_LAUNCH
}
void l_decide(const K2Launch& L, DevProblem P, DevTree T, const int32_t* frontier, int nf,
              int32_t* open_flag, DevCounters* cnt, int sign_only) {
    P.wc_lds = L.wc_lds;
    hipLaunchKernelGGL(k2_lcss_decide, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream, P,
                       T, frontier, nf, open_flag, cnt, L.wave_doubles, sign_only);
}
void l_expand(const K2Launch& L, DevProblem P, DevTree T, const int32_t* open_list, int n_open,
              int child_base, int32_t* next_frontier, DevCounters* cnt) {
    P.wc_lds = L.wc_lds;
    hipLaunchKernelGGL(k2_lcss_expand, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream, P,
                       T, open_list, n_open, child_base, next_frontier, cnt, L.wave_doubles);
}
void l_vertex(const K2Launch& L, DevProblem P, DevTree T, const int32_t* nodes, int n_nodes,
              DevCounters* cnt) {
    P.wc_lds = L.wc_lds;
    hipLaunchKernelGGL(k2_vertex_solve, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream, P,
                       T, nodes, n_nodes, cnt, L.wave_doubles);
}
void l_persist(const K2Launch& L, DevProblem P, DevTree T, int32_t* slots, int n_slots,
               PersistCtl* ctl, int node_cap, DevCounters* cnt, int sign_only, int max_depth,
               PersistDeal deal) {
    P.wc_lds = L.wc_lds;
    hipLaunchKernelGGL(k2_persist, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream, P, T,
                       slots, n_slots, ctl, node_cap, cnt, L.wave_doubles, sign_only, max_depth,
                       deal);
}
void l_selftest(hipStream_t stream, double* out) {
    hipLaunchKernelGGL(k2_selftest, dim3(1), dim3(64), 0, stream, out);
}

const K2Api g_api = {EHM_NP,   EHM_SLOTS,        EHM_K2_THREADS,     64,      set_lds,
                     wave_doubles_for, shared_doubles_for, l_point, l_simplex, l_decide,
                     l_expand, l_vertex,         l_selftest,     l_persist};

}  // namespace

#define K2_CAT2(a, b, c) a##b##_##c
#define K2_CAT(a, b, c) K2_CAT2(a, b, c)
#if EHM2_QUAD
extern "C" const ehm::K2Api* K2_CAT(ehm_k2q_api_, EHM_NP, EHM_SLOTS)() { return &g_api; }
#else
extern "C" const ehm::K2Api* K2_CAT(ehm_k2_api_, EHM_NP, EHM_SLOTS)() { return &g_api; }
#endif
