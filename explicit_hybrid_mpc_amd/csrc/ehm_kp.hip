// Persistent frontier kernel at two solver widths (gfx950).  Same kernel as k2_persist in
// ehm_k2.hip (see there and DESIGN.md section 4 for the queue protocol), but the suboptimality-
// test LPs (n + p + 1 columns) and the midpoint LPs (n columns) each run in the instance of
// ehm_ipm2.h that fits them: a row of the normal matrix lives in registers, so the column
// capacity is a compile-time size and every elimination step costs that many FMAs per lane.
// Compiled per (EHM_NPD, EHM_NPE, EHM_SLOTS); ehm_capi.hip uses it when a matching pair exists.
#include <hip/hip_runtime.h>

#ifndef EHM_NPD
#error "EHM_NPD / EHM_NPE (column capacities of the two LP kinds) must be defined"
#endif

#define EHM_NP EHM_NPD
#include "ehm_k2_asm.h"
namespace kd = EHM2_NS;
#undef EHM_NP
#define EHM_NP EHM_NPE
#include "ehm_k2_asm.h"
namespace ke = EHM2_NS;
#undef EHM_NP

using namespace ehm;

// one named namespace per instance (kernels of different objects must not share a symbol)
#define KP_CAT2(a, b, c, d) a##b##_##c##_##d
#define KP_CAT(a, b, c, d) KP_CAT2(a, b, c, d)
#if EHM2_QUAD
#define KP_NS KP_CAT(ehm_kpq_, EHM_NPD, EHM_NPE, EHM_SLOTS)
#else
#define KP_NS KP_CAT(ehm_kp_, EHM_NPD, EHM_NPE, EHM_SLOTS)
#endif

namespace KP_NS {

#define EHM_PERSIST_WATCHDOG_TICKS (60LL * 100000000LL)    // 60 s of the 100 MHz wall clock
constexpr int SLOTS = EHM_SLOTS;

__global__ __launch_bounds__(EHM_K2_THREADS) void kp_persist(
    DevProblem P, DevTree T, int32_t* slots, int n_slots, PersistCtl* ctl, int node_cap,
    DevCounters* cnt, int wave_doubles, int sign_only, int max_depth) {
    double* sm = reinterpret_cast<double*>(k2_smem);
    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    kd::Shared S;           // the two namespaces describe the same LDS image
    kd::carve_shared(S, sm, P);
    ke::Shared Se;
    ke::carve_shared(Se, sm, P);
    kd::NodeBuf nb;
    kd::carve_node(nb, sm + kd::shared_doubles(P) + (size_t)wave * wave_doubles, P.p, P.n_u);
    ke::NodeBuf nbe;
    ke::carve_node(nbe, sm + kd::shared_doubles(P) + (size_t)wave * wave_doubles, P.p, P.n_u);
    const int p = P.p, n_u = P.n_u;
    const int nrec = rec_doubles(p, n_u);
    kd::load_shared(P, 0, sm, tid, blockDim.x);
    __syncthreads();
    const long long t_start = wall_clock64();
    // statistics are kept per wavefront (in LDS: registers are what this kernel is short of) and
    // added to the global counters ONCE, when it leaves -- the level-synchronous kernels pay ~8
    // device atomics per node for them
    unsigned long long* wst = reinterpret_cast<unsigned long long*>(nb.aug);
    double* wmargin = nb.aug + 12;
    enum { W_SOLVES = 0, W_ITERS, W_STALLED, W_ERRORS, W_SLACK, W_SLACK_ITERS, W_CLOSED, W_SPLITS,
           W_DEPTH, W_TRUNC };
    if (lane0 < 12) wst[lane0] = 0ULL;
    if (lane0 == 0) *wmargin = 1e300;
    kd::wsync();
    for (;;) {
        int id = -1;
        if (lane0 == 0) {
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
                    if (wall_clock64() - t_start > EHM_PERSIST_WATCHDOG_TICKS) {
                        atomicMax(&ctl->abort, 3);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(64);
                }
            }
        }
        id = __builtin_amdgcn_readfirstlane(id);
        if (id < 0) break;
        // acquire (L1 / non-local L2 invalidate): the record behind the slot is visible
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int lane = kd::pin(lane0);
        const double* rec = T.rec + (size_t)id * T.rec_stride;
        double* node = nb.rec;
        for (int k = lane; k < nrec; k += 64) node[k] = rec[k];
        kd::wsync();
        // ---- suboptimality test --------------------------------------------------------------
        kd::IpmResult r;
        int its = 0;
        {
            kd::Wave W;
            for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
                double b[kd::SLOTS];
                const int ln = kd::pin(lane);
                kd::assemble_simplex(S, W, nb, node, node + rec_off_vcost(p), SX_SLACK, P.eps_a,
                                     P.eps_r, b, ln, P, 0);
                r = kd::ipm_solve(S, W, b, ln, sign_only != 0, kd::step_fraction(attempt));
                its += r.iters;
                if (r.status == 0) break;
            }
        }
        r.iters = its;
        const double tst = -r.obj;
        const bool open = (tst >= 0.0);
        const int dep = T.depth[id];
        if (lane == 0) {
            wst[W_SOLVES] += 1;
            wst[W_ITERS] += (unsigned long long)r.iters;
            wst[W_SLACK] += 1;
            wst[W_SLACK_ITERS] += (unsigned long long)r.iters;
            if (r.status != 0) {
                wst[W_STALLED] += 1;
                wst[W_ERRORS] += 1;
                T.flags[id] |= 8;
            }
            *wmargin = fmin(*wmargin, r.margin);
            if ((unsigned long long)dep > wst[W_DEPTH]) wst[W_DEPTH] = (unsigned long long)dep;
            T.tstar[id] = tst;
            if (!open) {
                T.flags[id] |= 1;
                wst[W_CLOSED] += 1;
            }
        }
        if (!open) {
            if (lane == 0) atomicSub(&ctl->pending, 1);
            kd::wsync();
            continue;
        }
        if (max_depth > 0 && dep >= max_depth) {
            if (lane == 0) {
                wst[W_TRUNC] = 1;
                atomicSub(&ctl->pending, 1);
            }
            kd::wsync();
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
        longest_edge(node, p, bi, bj);
        if (lane < p) {
#pragma clang fp contract(off)
            mid[lane] = (node[bi * p + lane] + node[bj * p + lane]) / 2.0;
        }
        kd::wsync();
        const int d = T.didx[id];
        // the midpoint LP has p + 1 columns less: it runs in the narrower instance
        ke::Wave W;
        ke::IpmResult re;
        its = 0;
        for (int attempt = 0; attempt < EHM2_ATTEMPTS; ++attempt) {
            double b[ke::SLOTS];
            const int ln = kd::pin(lane);
            ke::assemble_point(Se, W, nbe.lp, mid, false, b, ln, P, 0);
            re = ke::ipm_solve(Se, W, b, ln, false, ke::step_fraction(attempt));
            its += re.iters;
            if (re.status == 0) break;
        }
        r.iters = its;
        r.status = re.status;
        r.obj = re.obj;
        if (lane == 0) {
            wst[W_SOLVES] += 1;
            wst[W_ITERS] += (unsigned long long)r.iters;
            wst[W_SPLITS] += 1;
            if (r.status != 0) {
                wst[W_STALLED] += 1;
                wst[W_ERRORS] += 1;
                T.flags[id] |= 16;
            }
        }
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
                if (q >= bi * n_u && q < bi * n_u + n_u) v0 = W.xb[q - bi * n_u];
                if (q >= bj * n_u && q < bj * n_u + n_u) v1 = W.xb[q - bj * n_u];
            }
            // everything a child's consumer reads or later overwrites is written THROUGH to the
            // device coherence point (agent-scope atomic stores): visible to the other XCDs
            // without writing this XCD's whole L2 back (a device-scope release fence would --
            // measured: 47 GB of write-backs per partition, mostly register spills), and no
            // dirty copy stays behind that could later clobber the consumer's own writes
            __hip_atomic_store(rec0 + k, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(rec1 + k, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            T.left[id] = c0;
#define EHM_WT(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
            EHM_WT(&T.left[c0], -1);
            EHM_WT(&T.left[c0 + 1], -1);
            EHM_WT(&T.didx[c0], d);
            EHM_WT(&T.didx[c0 + 1], d);
            EHM_WT(&T.depth[c0], dep + 1);
            EHM_WT(&T.depth[c0 + 1], dep + 1);
            EHM_WT(&T.flags[c0], (uint8_t)2);
            EHM_WT(&T.flags[c0 + 1], (uint8_t)2);
            EHM_WT(&T.tstar[c0], 0.0);
            EHM_WT(&T.tstar[c0 + 1], 0.0);
#undef EHM_WT
        }
        // the write-through stores above have completed (s_waitcnt) before the slots go out
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) {
            const int t = atomicAdd(&ctl->tail, 2);
            if (t + 2 <= n_slots) {
                __hip_atomic_store(&slots[t], c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&slots[t + 1], c0 + 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&ctl->pending, 1);       // -1 (this node) + 2 (its children)
            } else {
                atomicMax(&ctl->abort, 1);
                atomicSub(&ctl->pending, 1);
            }
        }
        kd::wsync();
    }
    kd::wsync();
    if (lane0 == 0) {
        atomicAdd(&cnt->lp_solves, wst[W_SOLVES]);
        atomicAdd(&cnt->ipm_iters, wst[W_ITERS]);
        if (wst[W_STALLED]) atomicAdd(&cnt->stalled, wst[W_STALLED]);
        if (wst[W_ERRORS]) atomicAdd(&cnt->errors, wst[W_ERRORS]);
        atomicAdd(&cnt->slack_solves, wst[W_SLACK]);
        atomicAdd(&cnt->slack_iters, wst[W_SLACK_ITERS]);
        atomicMin(&cnt->min_margin_bits, (unsigned long long)__double_as_longlong(*wmargin));
        atomicAdd(&ctl->closed, wst[W_CLOSED]);
        atomicAdd(&ctl->splits, wst[W_SPLITS]);
        atomicMax(&ctl->max_depth_seen, (int)wst[W_DEPTH]);
        if (wst[W_TRUNC]) atomicMax(&ctl->truncated, 1);
    }
}


hipError_t set_lds(int bytes) {
    return hipFuncSetAttribute((const void*)kp_persist,
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
size_t wave_doubles_for(const DevProblem& P, int n_lp_d, int ne_d, int n_lp_e) {
    const size_t a = kd::wave_lp_doubles(n_lp_d, ne_d), b = ke::wave_lp_doubles(n_lp_e, 0);
    return k2_node_doubles(P.p, P.n_u) + (a > b ? a : b);
}
size_t shared_doubles_for(const DevProblem& P) { return kd::shared_doubles(P); }
void l_persist(const K2Launch& L, DevProblem P, DevTree T, int32_t* slots, int n_slots,
               PersistCtl* ctl, int node_cap, DevCounters* cnt, int sign_only, int max_depth) {
    hipLaunchKernelGGL(kp_persist, dim3(L.grid), dim3(L.threads), L.lds_bytes, L.stream, P, T,
                       slots, n_slots, ctl, node_cap, cnt, L.wave_doubles, sign_only, max_depth);
}

const KpApi g_api = {EHM_NPD, EHM_NPE, EHM_SLOTS, EHM_K2_THREADS, set_lds, wave_doubles_for,
                     shared_doubles_for, l_persist};

}  // namespace KP_NS

#if EHM2_QUAD
extern "C" const ehm::KpApi* KP_CAT(ehm_kpq_api_, EHM_NPD, EHM_NPE, EHM_SLOTS)() {
    return &KP_NS::g_api;
}
#else
extern "C" const ehm::KpApi* KP_CAT(ehm_kp_api_, EHM_NPD, EHM_NPE, EHM_SLOTS)() {
    return &KP_NS::g_api;
}
#endif
