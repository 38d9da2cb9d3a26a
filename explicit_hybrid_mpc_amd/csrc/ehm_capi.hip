// libehmpc.so -- kernels and C-ABI (include/ehmpc.h) of the MI355X partitioning hot path.
// gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../../include/ehmpc.h"

#define EHM_MAX_P_DEV 8
#include "ehm_kernels.h"
#include "ehm_k2.h"

using namespace ehm;

// second-generation kernel instances (ehm_k2.hip compiled per column capacity / row slots)
#define EHM_K2_NPS(X, S) X(8, S) X(12, S) X(16, S) X(20, S) X(24, S) X(28, S) X(32, S)
#define EHM_K2_ALL(X) EHM_K2_NPS(X, 1) EHM_K2_NPS(X, 2) EHM_K2_NPS(X, 3) EHM_K2_NPS(X, 4)
#define EHM_K2_DECL(NPV, SL) extern "C" const ehm::K2Api* ehm_k2_api_##NPV##_##SL();
EHM_K2_ALL(EHM_K2_DECL)
#define EHM_K2_ENTRY(NPV, SL) ehm_k2_api_##NPV##_##SL,
// wide instances (ehm_k3.hip compiled per row capacity): LPs with 33..64 columns or > 256 rows
extern "C" const ehm::K2Api* ehm_k3_api_2();
extern "C" const ehm::K2Api* ehm_k3_api_4();
// the LDS-resident wide family (ehm_k4.hip): taken instead of ehm_k3 where the reduced block fits
extern "C" const ehm::K2Api* ehm_k4_api();
#define EHM_LDS_BUDGET (160 * 1024 - 256)   // dynamic LDS a workgroup may ask for
typedef const ehm::K2Api* (*k2_getter)();
static const k2_getter g_k2_getters[] = {EHM_K2_ALL(EHM_K2_ENTRY) ehm_k3_api_2, ehm_k3_api_4,
                                         ehm_k4_api};
// instances of ehm_k2.hip with the quadratic block (-DEHM2_QUAD=1): convex QP / QCQP
#define EHM_K2Q_NPS(X, S) X(8, S) X(16, S) X(24, S) X(32, S)
#define EHM_K2Q_ALL(X) EHM_K2Q_NPS(X, 1) EHM_K2Q_NPS(X, 2) EHM_K2Q_NPS(X, 3) EHM_K2Q_NPS(X, 4)
#define EHM_K2Q_DECL(NPV, SL) extern "C" const ehm::K2Api* ehm_k2q_api_##NPV##_##SL();
EHM_K2Q_ALL(EHM_K2Q_DECL)
#define EHM_K2Q_ENTRY(NPV, SL) ehm_k2q_api_##NPV##_##SL,
static const k2_getter g_k2q_getters[] = {EHM_K2Q_ALL(EHM_K2Q_ENTRY)};
// persistent frontier kernel at two solver widths (ehm_kp.hip): (decide NP, expand NP, slots)
// (the widths are those of the FACTORISED columns: with the eliminated block of ehm_ipm2.h the
// suboptimality-test LP of config 2 runs at 16 and its midpoint LP at 12)
#define EHM_KP_SL(X, D, E) X(D, E, 2) X(D, E, 3)
#define EHM_KP_ALL(X) EHM_KP_SL(X, 12, 8) EHM_KP_SL(X, 16, 8) EHM_KP_SL(X, 16, 12) \
    EHM_KP_SL(X, 20, 12) EHM_KP_SL(X, 20, 16) EHM_KP_SL(X, 24, 16) EHM_KP_SL(X, 24, 20) \
    EHM_KP_SL(X, 28, 20) EHM_KP_SL(X, 28, 24) EHM_KP_SL(X, 32, 24) EHM_KP_SL(X, 32, 28) \
    X(16, 12, 4) X(28, 20, 4) X(32, 24, 4)
#define EHM_KP_DECL(D, E, SL) extern "C" const ehm::KpApi* ehm_kp_api_##D##_##E##_##SL();
EHM_KP_ALL(EHM_KP_DECL)
#define EHM_KP_ENTRY(D, E, SL) ehm_kp_api_##D##_##E##_##SL,
typedef const ehm::KpApi* (*kp_getter)();
static const kp_getter g_kp_getters[] = {EHM_KP_ALL(EHM_KP_ENTRY)};
// the same with the midpoint solve BEFORE the suboptimality test (-DEHM_PERSIST_MIDFIRST=1;
// option "mid_first", the default: identical tree, 13 % faster on the bench workload)
#define EHM_KPM_ALL(X) EHM_KP_ALL(X)
#define EHM_KPM_DECL(D, E, SL) extern "C" const ehm::KpApi* ehm_kpm_api_##D##_##E##_##SL();
EHM_KPM_ALL(EHM_KPM_DECL)
#define EHM_KPM_ENTRY(D, E, SL) ehm_kpm_api_##D##_##E##_##SL,
static const kp_getter g_kpm_getters[] = {EHM_KPM_ALL(EHM_KPM_ENTRY)};
#define EHM_V1_MAX_N 32     // limits of the generation-1 and wave-local kernels
#define EHM_V1_MAX_M 256

// =========================================================================================
// kernels: one 64-lane workgroup (= one wavefront) per LP / per node
// =========================================================================================
#define NODE_LDS_DOUBLES 160   // >= (p+1)*p + (p+1) + (p+1)*n_u for p <= 8, n_u <= 8 (checked)

extern __shared__ __attribute__((aligned(16))) char ehm_smem[];

__device__ __forceinline__ void count_solve(DevCounters* cnt, const IpmResult& r, int lane) {
    if (lane == 0 && cnt) {
        atomicAdd(&cnt->lp_solves, 1ULL);
        atomicAdd(&cnt->ipm_iters, (unsigned long long)r.iters);
        if (r.status != 0) atomicAdd(&cnt->stalled, 1ULL);
    }
}

// a2: P_theta_delta batch / its feasibility form
__global__ __launch_bounds__(64) void k_point_batch(DevProblem P, long long n_inst,
                                                    const double* __restrict__ theta,
                                                    const int32_t* __restrict__ didx,
                                                    int feas, double* __restrict__ J,
                                                    double* __restrict__ u0,
                                                    int32_t* __restrict__ status,
                                                    int32_t* __restrict__ iters,
                                                    DevCounters* cnt, K2Gather G) {
    double* smem = reinterpret_cast<double*>(ehm_smem);
    double* th = smem;                 // p doubles
    double* lp_base = smem + 16;
    const int lane = threadIdx.x;
    if (G.n_dev) n_inst = *G.n_dev;
    for (long long inst0 = blockIdx.x; inst0 < n_inst; inst0 += gridDim.x) {
        wave_sync();
        const double* tsrc = G.src ? theta + G.src[inst0] : theta + inst0 * P.p;
        if (lane < P.p) th[lane] = tsrc[lane];
        wave_sync();
        const int d = didx ? didx[inst0] : 0;
        const long long inst = G.dst ? (long long)G.dst[inst0] : inst0;   // results go here
        LpWork w;
        double b[EHM_SLOTS];
        assemble_point(w, lp_base, P, d, th, feas != 0, b, lane);
        const IpmResult r = ipm_solve(w, b, lane);
        count_solve(cnt, r, lane);
        if (lane == 0) {
            J[inst] = r.obj;
            if (status) status[inst] = ehm_status_word(r.status, r.merit);     // decade of the merit: ehm_dev.h
            if (iters) iters[inst] = r.iters;
        }
        if (u0 && lane < P.n_u) u0[inst * P.n_u + lane] = w.xb[lane];
    }
}

// a5 / a7': slack of the suboptimality test, or min over the simplex, one commutation each
__global__ __launch_bounds__(64) void k_simplex_batch(DevProblem P, long long n_inst,
                                                      const double* __restrict__ R,
                                                      const double* __restrict__ Vbar,
                                                      const int32_t* __restrict__ didx,
                                                      int slack, double* __restrict__ obj,
                                                      double* __restrict__ alpha,
                                                      int32_t* __restrict__ status,
                                                      int32_t* __restrict__ iters,
                                                      DevCounters* cnt, K2Gather G) {
    double* smem = reinterpret_cast<double*>(ehm_smem);
    double* Rl = smem;                          // (p+1)*p
    double* Vl = smem + (P.p + 1) * P.p;        // p+1
    double* lp_base = smem + NODE_LDS_DOUBLES;
    const int lane = threadIdx.x;
    const int nR = (P.p + 1) * P.p;
    if (G.n_dev) n_inst = *G.n_dev;
    for (long long inst0 = blockIdx.x; inst0 < n_inst; inst0 += gridDim.x) {
        wave_sync();
        const double* Rsrc = G.src ? R + G.src[inst0] : R + inst0 * nR;
        const double* Vsrc = G.src ? Rsrc + G.v_off : Vbar + inst0 * (P.p + 1);
        for (int k = lane; k < nR; k += 64) Rl[k] = Rsrc[k];
        if (slack == SX_SLACK && lane <= P.p) Vl[lane] = Vsrc[lane];
        wave_sync();
        const int d = didx ? didx[inst0] : 0;
        const long long inst = G.dst ? (long long)G.dst[inst0] : inst0;   // results go here
        LpWork w;
        double b[EHM_SLOTS];
        assemble_simplex(w, lp_base, P, d, Rl, Vl, slack, b, lane);
        const IpmResult r = ipm_solve(w, b, lane);
        count_solve(cnt, r, lane);
        if (lane == 0) {
            obj[inst] = (slack == SX_SLACK) ? -r.obj : r.obj;     // t* = -(min -t)
            if (status) status[inst] = ehm_status_word(r.status, r.merit);     // decade of the merit: ehm_dev.h
            if (iters) iters[inst] = r.iters;
        }
        if (alpha) {
            double beta = (lane < P.p) ? w.xb[P.n + lane] : 0.0;
            const double sb = wave_sum(beta);
            if (lane < P.p) alpha[inst * (P.p + 1) + lane + 1] = beta;
            if (lane == 0) alpha[inst * (P.p + 1)] = 1.0 - sb;
        }
    }
}

// a12: split_along_longest_edge, one simplex per thread
__global__ void k_split_batch(long long n, int p, const double* __restrict__ R,
                              double* __restrict__ S1, double* __restrict__ S2,
                              int32_t* __restrict__ ij) {
#pragma clang fp contract(off)
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int nR = (p + 1) * p;
    const double* r = R + k * nR;
    int bi, bj;
    longest_edge(r, p, bi, bj);
    double* s1 = S1 + k * nR;
    double* s2 = S2 + k * nR;
    for (int q = 0; q < nR; ++q) {
        s1[q] = r[q];
        s2[q] = r[q];
    }
    for (int c = 0; c < p; ++c) {
        const double mid = (r[bi * p + c] + r[bj * p + c]) / 2.0;
        s1[bi * p + c] = mid;
        s2[bj * p + c] = mid;
    }
    ij[2 * k] = bi;
    ij[2 * k + 1] = bj;
}

// a14: simplex_volume = |det([v_i - v_0])| / p!, one simplex per thread
__device__ inline double simplex_volume_dev(const double* r, int p) {
    double M[EHM_MAX_P_DEV * EHM_MAX_P_DEV];
    for (int i = 0; i < p; ++i)
        for (int c = 0; c < p; ++c) M[c * p + i] = r[(i + 1) * p + c] - r[c];   // column i
    double det = 1.0;
    for (int k = 0; k < p; ++k) {
        int piv = k;
        double best = fabs(M[k * p + k]);
        for (int i = k + 1; i < p; ++i)
            if (fabs(M[i * p + k]) > best) {
                best = fabs(M[i * p + k]);
                piv = i;
            }
        if (best == 0.0) return 0.0;
        if (piv != k) {
            for (int c = 0; c < p; ++c) {
                const double t = M[k * p + c];
                M[k * p + c] = M[piv * p + c];
                M[piv * p + c] = t;
            }
            det = -det;
        }
        det *= M[k * p + k];
        for (int i = k + 1; i < p; ++i) {
            const double l = M[i * p + k] / M[k * p + k];
            for (int c = k + 1; c < p; ++c) M[i * p + c] -= l * M[k * p + c];
        }
    }
    double fact = 1.0;
    for (int k = 2; k <= p; ++k) fact *= k;
    return fabs(1.0 / fact * det);
}

__global__ void k_volume_batch(long long n, int p, const double* __restrict__ R,
                               double* __restrict__ vol) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    vol[k] = simplex_volume_dev(R + k * (p + 1) * p, p);
}

// Progress of a run (lib/worker.py:374-375 "volume_filled_increment"): volume of the closed
// leaves of pool nodes [0, n).  Thread per node, fixed-shape tree reduction per workgroup,
// one partial sum per workgroup (the host adds them in order: reproducible).
__global__ __launch_bounds__(256) void k_closed_volume(DevTree T, long long first, long long n,
                                                       double* __restrict__ partial) {
    __shared__ double red[256];
    const long long k = first + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    if (k < n && (T.flags[k] & 1)) v = simplex_volume_dev(T.rec + (size_t)k * T.rec_stride, T.p);
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---- frontier sweep, level-synchronous engine ---------------------------------------------
// K1: epsilon-suboptimality decision for every frontier node (lib/worker.py:368-375).
__global__ __launch_bounds__(64) void k_lcss_decide(DevProblem P, DevTree T,
                                                    const int32_t* __restrict__ frontier,
                                                    int nf, int32_t* __restrict__ open_flag,
                                                    DevCounters* cnt) {
    double* smem = reinterpret_cast<double*>(ehm_smem);
    double* node = smem;
    double* lp_base = smem + NODE_LDS_DOUBLES;
    const int lane = threadIdx.x;
    const int nrec = rec_doubles(P.p, P.n_u);
    for (int f = blockIdx.x; f < nf; f += gridDim.x) {
        wave_sync();
        const int id = frontier[f];
        const double* rec = T.rec + (size_t)id * T.rec_stride;
        for (int k = lane; k < nrec; k += 64) node[k] = rec[k];
        wave_sync();
        LpWork w;
        double b[EHM_SLOTS];
        assemble_simplex(w, lp_base, P, T.didx[id], node, node + rec_off_vcost(P.p), SX_SLACK,
                         b, lane);
        const IpmResult r = ipm_solve(w, b, lane);
        count_solve(cnt, r, lane);
        if (r.status != 0 && lane == 0) {
            atomicAdd(&cnt->errors, 1ULL);
            T.flags[id] |= 8;
        }
        if (lane == 0) {
            atomicAdd(&cnt->slack_solves, 1ULL);
            atomicAdd(&cnt->slack_iters, (unsigned long long)r.iters);
            const double t = -r.obj;
            const bool open = (t >= 0.0);
            T.tstar[id] = t;
            open_flag[f] = open ? 1 : 0;
            if (!open) T.flags[id] |= 1;
            const double a = fabs(t);
            atomicMin(&cnt->min_margin_bits, (unsigned long long)__double_as_longlong(a));
        }
    }
}

// K2: exclusive scan of the open flags -> compacted list of open nodes (single workgroup).
__global__ __launch_bounds__(1024) void k_scan_open(const int32_t* __restrict__ open_flag,
                                                    const int32_t* __restrict__ frontier,
                                                    int nf, int32_t* __restrict__ open_list,
                                                    int32_t* __restrict__ count) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int chunk = (nf + 1023) / 1024;
    const int lo = tid * chunk, hi = min(nf, lo + chunk);
    int s = 0;
    for (int k = lo; k < hi; ++k) s += open_flag[k];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int base = part[tid] - s;
    for (int k = lo; k < hi; ++k)
        if (open_flag[k]) open_list[base++] = frontier[k];
    if (tid == 1023) *count = part[1023];
}

// Multi-GPU sharding: keep frontier position k iff k % world == rank; flag the others.
__global__ void k_shard_filter(DevTree T, const int32_t* __restrict__ frontier, int nf, int rank,
                               int world, int32_t* __restrict__ kept) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nf) return;
    const int id = frontier[k];
    if (k % world == rank)
        kept[k / world] = id;
    else
        T.flags[id] |= 4;
}

// K3: split every open node along its longest edge, solve P_theta_delta at the midpoint
// and write the two child records (lib/worker.py:403-414, 354-365).
__global__ __launch_bounds__(64) void k_lcss_expand(DevProblem P, DevTree T,
                                                    const int32_t* __restrict__ open_list,
                                                    int n_open, int child_base,
                                                    int32_t* __restrict__ next_frontier,
                                                    DevCounters* cnt) {
    double* smem = reinterpret_cast<double*>(ehm_smem);
    double* node = smem;
    double* mid = smem + NODE_LDS_DOUBLES - 8;    // p doubles
    double* lp_base = smem + NODE_LDS_DOUBLES;
    const int lane = threadIdx.x;
    const int p = P.p, n_u = P.n_u;
    const int nrec = rec_doubles(p, n_u);
    for (int f = blockIdx.x; f < n_open; f += gridDim.x) {
        wave_sync();
        const int id = open_list[f];
        const double* rec = T.rec + (size_t)id * T.rec_stride;
        for (int k = lane; k < nrec; k += 64) node[k] = rec[k];
        wave_sync();
        int bi, bj;
        longest_edge(node, p, bi, bj);
        if (lane < p) {
#pragma clang fp contract(off)
            mid[lane] = (node[bi * p + lane] + node[bj * p + lane]) / 2.0;
        }
        wave_sync();
        const int d = T.didx[id];
        LpWork w;
        double b[EHM_SLOTS];
        assemble_point(w, lp_base, P, d, mid, false, b, lane);
        const IpmResult r = ipm_solve(w, b, lane);
        count_solve(cnt, r, lane);
        if (r.status != 0 && lane == 0) {
            atomicAdd(&cnt->errors, 1ULL);
            T.flags[id] |= 16;
        }
        const int c0 = child_base + 2 * f;
        double* rec0 = T.rec + (size_t)c0 * T.rec_stride;
        double* rec1 = rec0 + T.rec_stride;
        const int ov = rec_off_vcost(p), ou = rec_off_vinput(p);
        for (int k = lane; k < nrec; k += 64) {
            double v0 = node[k], v1 = node[k];
            if (k < ov) {                       // vertices
                const int row = k / p, col = k - row * p;
                if (row == bi) v0 = mid[col];
                if (row == bj) v1 = mid[col];
            } else if (k < ou) {                // vertex costs
                const int row = k - ov;
                if (row == bi) v0 = r.obj;
                if (row == bj) v1 = r.obj;
            } else {                            // vertex inputs
                const int row = (k - ou) / n_u, col = (k - ou) - row * n_u;
                if (row == bi) v0 = w.xb[col];
                if (row == bj) v1 = w.xb[col];
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
    }
}

// vertex solves that seed a node's costs / inputs (lib/oracle.py:416-443): one LP per
// (node, vertex) pair, results written straight into the node record.
__global__ __launch_bounds__(64) void k_vertex_solve(DevProblem P, DevTree T,
                                                     const int32_t* __restrict__ nodes,
                                                     int n_nodes, DevCounters* cnt) {
    double* smem = reinterpret_cast<double*>(ehm_smem);
    double* th = smem;
    double* lp_base = smem + 16;
    const int lane = threadIdx.x;
    const int p = P.p, n_u = P.n_u;
    const int total = n_nodes * (p + 1);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        wave_sync();
        const int id = nodes[t / (p + 1)];
        const int v = t % (p + 1);
        double* rec = T.rec + (size_t)id * T.rec_stride;
        if (lane < p) th[lane] = rec[v * p + lane];
        wave_sync();
        LpWork w;
        double b[EHM_SLOTS];
        assemble_point(w, lp_base, P, T.didx[id], th, false, b, lane);
        const IpmResult r = ipm_solve(w, b, lane);
        count_solve(cnt, r, lane);
        if (r.status != 0 && lane == 0) atomicAdd(&cnt->errors, 1ULL);
        if (lane == 0) rec[rec_off_vcost(p) + v] = r.obj;
        if (lane < n_u) rec[rec_off_vinput(p) + v * n_u + lane] = w.xb[lane];
    }
}

// =========================================================================================
// host side
// =========================================================================================
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr, code)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(code, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                        __FILE__, __LINE__);                                             \
    } while (0)

struct DevBuf {
    void* ptr = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return EHM_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        size_t want = std::max(bytes, (size_t)4096);
        HIP_TRY(hipMalloc(&ptr, want), EHM_E_HIP);
        cap = want;
        return EHM_OK;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() { return reinterpret_cast<T*>(ptr); }
};

struct ehm_problem {
    int device = 0;
    hipStream_t stream = nullptr;
    DevProblem dp{};
    int delta_len = 0;
    std::vector<uint8_t> deltas;
    DevBuf consts;           // Gt | St | w | c
    DevBuf wc2;              // [n_delta][n+p+2][m|1]  LDS image, all columns (wide kernels; the
                             // shared-block kernels when no column is eliminated)
    DevBuf wc4;              // [n_delta][tot4]  LDS image without the eliminated columns + their
                             // tables (DevProblem::Wc4; unused when nd0 == n: Wc4 aliases Wc2)
    DevBuf wr3;              // [n_delta][mpad][64]    row-major image for the wide kernels
    DevBuf quad;             // H | F^T | f0 | C | c1 | c0  (ehm_problem_set_quadratic)
    bool quadratic = false;
    bool k2q_ok = false;     // a quadratic shared-block instance fits (ehm_k2.hip, EHM2_QUAD)
    bool v1_ok = false;      // the generation-1 kernels fit this problem
    int decide_full = 0;     // 1 = the suboptimality test solves to full accuracy (no sign-only stop)
    int mid_first = 1;       // 1 = persistent kernel with the midpoint solve first (default)
    int inherit_wit = 1;     // 1 = open nodes hand the point that proved them open to the child
                             // that contains it (DevTree::wit; option "inherit_witness")
    bool budget_keep = false;   // budgeted launches keep one child too (option "budget_keep")
    bool check_witness = false; // PersistDeal::check (option "check_witness")
    int work_first = 1;      // 1 = a wavefront of the persistent kernel that splits a node goes on
                             // with one child itself and queues the other (option "work_first")
    int share_mid = 1;       // 1 = the persistent kernel keeps a table of midpoint optima: the
                             // simplices around an edge solve its midpoint once (DevTree::mt;
                             // option "share_midpoints")
    int any_admissible = 0;  // multi-commutation engine: seed + 1 = V_R and bar_D return a hashed
                             // draw among the admissible commutations instead of the canonical
                             // one (ehm_hybrid.h, hy_draw; option "any_admissible"), 0 = canonical
    int hy_timing = 0;       // 1 = event pairs + counter snapshots around every batch of the
                             // multi-commutation engine (kernel seconds, solves by kind)
    int solver_gen = 2;      // 1 = one wavefront per workgroup (ehm_kernels.h), 2 = ehm_k2.hip
    DevBuf seg;              // commutation segment offsets of a sorted batch
    std::set<const K2Api*> k2_ready;
    std::set<const KpApi*> kp_ready;
    // device memory kept between partition runs (hipMalloc/hipFree of a GB-sized node pool
    // cost milliseconds): the pool of the last destroyed tree and the frontier scratch
    struct PoolCache {
        long long cap = 0;
        DevBuf rec, left, didx, depth, flags, tstar, grad, wit, mt_state, mt_data;
    } pool_cache;
    DevBuf fr_a, fr_b, open_flag, open_list, d_count;
    DevBuf pq_slots, pq_ctl;   // persistent engine: queue slots, control block
    DevBuf in0, in1, in2, out0, out1, out2, out3;
    DevBuf ex_stage, ex_tmp;      // ehm_tree_export: staging of the gathered records, scan scratch
    DevCounters* d_cnt = nullptr;
    struct ehm_tree* active_run = nullptr;   // the partition run that owns the scratch above
    // every tree made from this handle that is still alive: ehm_problem_destroy detaches them, so
    // a tree destroyed (or asked about) AFTER its problem never touches freed memory
    std::vector<struct ehm_tree*> trees;
    // batched oracles (point_batch / simplex_batch): device time of their kernels, by HIP events
    // on the handle's stream around every launch ([0] point problems, [1] problems over a simplex)
    hipEvent_t bev[2] = {nullptr, nullptr};
    double batch_seconds[2] = {0.0, 0.0};
    long long batch_launches[2] = {0, 0};
    long long launches = 0;
    long long fallbacks = 0;   // LPs handed from the generation-2 to the generation-1 kernels
    long long slivers = 0;     // (simplex, commutation) pairs dropped as interior-free (slack_all)
    int num_cu = 256;
    size_t lds_point = 0, lds_simplex = 0, lds_expand = 0;
};

struct HyState;      // multi-commutation engine (ehm_hybrid.h)

struct ehm_tree {
    ehm_problem* prob = nullptr;     // nullptr once the problem handle has been destroyed
    int device = 0;
    HyState* hy = nullptr;
    DevTree dt{};
    long long cap = 0;       // allocated node records (a cached pool may be larger than asked for)
    long long limit = 0;     // max_nodes of this run: the capacity the caller agreed to
    DevBuf rec, left, didx, depth, flags, tstar, grad, code, wit, mt_state, mt_data;
    ehm_tree_info info{};
    int skip_volume = 0;
    // persistent engine: node ids follow the allocation order; the export relabels them to the
    // breadth-first order of the level-synchronous engine (perm[new id] = device id)
    bool unordered = false;
    bool keep_ids = false;   // runs that exchanged nodes with other ranks: the transfer logs name
                             // device ids, the export must not renumber
    DevBuf d_perm, d_inv;    // the numbering, once computed (k_bfs_*)
    bool have_perm = false;
    // state of a run in progress (ehm_partition_begin .. ehm_partition_finish)
    struct Run {
        bool active = false;
        int max_depth = 0, action = 0, shard_world = 1, shard_rank = 0, engine = 0;
        int deal_depth = 0;     // > 0: sharded persistent launch dealt at this tree depth
        long long shard_min = 0;
        bool sharded = true, cur_is_a = true;
        long long n_roots = 0, n_nodes = 0, nf = 0, n_closed = 0, ref_solves = 0;
        long long pre_closed = 0, pre_nodes = 0, pre_solves = 0, given = 0, received = 0;
        long long pre_first = 0;    // first node id of the frontier that was dealt
        int sweeps = 0, depth = 0, truncated = 0;
        DevCounters c0{};
        hipEvent_t ev0 = nullptr;
        std::vector<hipEvent_t> evs;    // (start, stop) pairs
        std::vector<int> ev_kind;       // 0 decide, 1 expand, per pair
    } run;
};

static int map_deltas(ehm_problem* P, int64_t n_inst, const uint8_t* delta,
                      std::vector<int32_t>& out) {
    out.resize((size_t)n_inst);
    const int L = P->delta_len;
    for (int64_t k = 0; k < n_inst; ++k) {
        int found = -1;
        if (!delta) {
            found = 0;
        } else {
            for (int d = 0; d < P->dp.n_delta; ++d) {
                bool eq = true;
                for (int q = 0; q < L && eq; ++q)
                    eq = ((delta[k * L + q] != 0) == (P->deltas[(size_t)d * L + q] != 0));
                if (eq) {
                    found = d;
                    break;
                }
            }
        }
        if (found < 0)
            return fail(EHM_E_INVALID, "instance %lld: not an admissible commutation",
                        (long long)k);
        out[(size_t)k] = found;
    }
    return EHM_OK;
}

static size_t lds_bytes_for(const DevProblem& dp, int kind, size_t prefix_doubles) {
    return (prefix_doubles + lp_lds_doubles(lp_cols(dp, kind), lp_rows(dp, kind))) * sizeof(double);
}

static int grid_for(ehm_problem* P, long long n) {
    long long cap = (long long)P->num_cu * 64;
    return (int)std::max(1LL, std::min(n, cap));
}

// ---- second-generation launch configuration --------------------------------------------------

static void kind_dims(const DevProblem& dp, int kind, int& n_lp, int& ne) {
    switch (kind) {
        case LP_POINT: n_lp = dp.n; ne = 0; break;
        case LP_FEAS: n_lp = dp.n + 1; ne = 1; break;
        case LP_MIN_SIMPLEX: n_lp = dp.n + dp.p; ne = dp.p + 1; break;
        case LP_FEAS_SIMPLEX: n_lp = dp.n + dp.p + 1; ne = dp.p + 2; break;
        default: n_lp = dp.n + dp.p + 1; ne = dp.p + 3; break;
    }
}

// smallest compiled instance that holds n_lp columns and `slots` row slots (dp: the problem, for
// families that do not take every problem of their class -- K2Api::fits; null = those are skipped)
static const K2Api* k2_pick(int n_lp, int slots, bool quad, const DevProblem* dp = nullptr) {
    const K2Api* best = nullptr;
    static const int no_k4 = [] {       // EHM_K4=0: the streaming wide kernels everywhere (A/B runs)
        const char* e = getenv("EHM_K4");
        return (e && atoi(e) == 0) ? 1 : 0;
    }();
    auto consider = [&](const K2Api* a) {
        if (a->np < n_lp || a->slots < slots) return;
        if (a->fits && (no_k4 || !dp || !a->fits(*dp, EHM_LDS_BUDGET))) return;
        if (!best || a->np < best->np || (a->np == best->np && a->slots < best->slots)) best = a;
    };
    if (quad) {
        for (k2_getter g : g_k2q_getters) consider(g());
    } else {
        for (k2_getter g : g_k2_getters) consider(g());
        // Where both fit, the LDS-resident wide family takes the LPs at the upper end of the
        // shared-block instances: at >= EHM_K4_MIN_NP factorised columns and 4 row slots one
        // wavefront per LP keeps its rows in 4 x 10 register pairs and the forced-inline solver
        // spills (configs[4], horizon-5 table: 3.0 us per LP against the wide family's 1.7).
        static const int k4_min_np = [] {
            const char* e = getenv("EHM_K4_MIN_NP");
            return e ? atoi(e) : 24;
        }();
        const K2Api* k4 = ehm_k4_api();
        if (best && best != k4 && best->threads_per_lp == 64 && n_lp >= k4_min_np && slots >= 4 &&
            !no_k4 && dp && k4->np >= n_lp && k4->slots >= slots && k4->fits(*dp, EHM_LDS_BUDGET))
            best = k4;
    }
    return best;
}

struct K2Cfg {
    const K2Api* api = nullptr;
    K2Launch L{};
};

// kind_a / kind_b: the LP kinds the launch may assemble (workspace sized for the larger)
static int k2_config(ehm_problem* P, int kind_a, int kind_b, long long n_items, K2Cfg& cfg,
                     bool persist = false) {
    int n_lp, ne, n_lp2, ne2;
    kind_dims(P->dp, kind_a, n_lp, ne);
    kind_dims(P->dp, kind_b, n_lp2, ne2);
    const int slots = std::max(lp_slots(P->dp.m, ne), lp_slots(P->dp.m, ne2));
    n_lp = std::max(n_lp, n_lp2);
    ne = std::max(ne, ne2);
    // (the instances are sized by the FACTORISED columns: ehm_ipm2.h eliminates [nd0, n))
    const K2Api* api = k2_pick(n_lp - (P->dp.n - P->dp.nd0), slots, P->quadratic, &P->dp);
    if (!api)
        return fail(EHM_E_INVALID, "no kernel instance for an LP with %d columns, %d row slots",
                    n_lp, slots);
    const size_t wave = api->wave_doubles(P->dp, n_lp, ne, persist ? 1 : 0);
    const size_t budget = EHM_LDS_BUDGET / sizeof(double);
    // w and c of the commutation behind the constant block in LDS -- unless that costs the
    // workgroup a wavefront (DevProblem::wc_lds; the wide kernels keep nothing in LDS)
    long long max_w = api->max_threads / api->threads_per_lp;
    if (const char* e = getenv("EHM_K2_MAX_WAVES"))     // experiments: cap the LPs per workgroup
        if (atoi(e) > 0 && api->threads_per_lp == 64) max_w = std::min<long long>(max_w, atoi(e));
    DevProblem dp = P->dp;
    dp.wc_lds = 1;
    const size_t shared1 = api->shared_doubles(dp);
    dp.wc_lds = 0;
    const size_t shared0 = api->shared_doubles(dp);
    if (shared0 + wave > budget)
        return fail(EHM_E_INVALID, "LP does not fit in LDS (%zu + %zu doubles)", shared0, wave);
    const long long nw1 = (shared1 + wave <= budget)
        ? std::min<long long>(max_w, (long long)((budget - shared1) / wave)) : 0;
    const long long nw0 = std::min<long long>(max_w, (long long)((budget - shared0) / wave));
    const int wc_lds = (nw1 >= nw0 && nw1 >= 1) ? 1 : 0;
    const size_t shared = wc_lds ? shared1 : shared0;
    if (api->threads_per_lp > 64 && !api->fits && !P->dp.Wr3)
        return fail(EHM_E_INVALID, "wide kernels selected but their constant image is missing");
    // LPs per workgroup: as many wavefronts as fit (wave-local kernels); exactly one (wide)
    long long nw = wc_lds ? nw1 : nw0;
    nw = std::max(1LL, std::min(nw, n_items));
    const size_t lds = (shared + (size_t)nw * wave) * sizeof(double);
    if (!P->k2_ready.count(api)) {
        HIP_TRY(api->set_lds(EHM_LDS_BUDGET), EHM_E_HIP);
        P->k2_ready.insert(api);
    }
    // residency: LDS, and the wavefronts per CU the instances' register budget admits
    // (wide kernels: 256 threads with up to 256 VGPRs -> two workgroups per CU)
    const long long reg_wg = (api->threads_per_lp > 256) ? 1
        : (api->threads_per_lp > 64) ? 2 : (api->max_threads / 64) / nw;
    long long wg_per_cu = std::max<long long>(1, std::min<long long>(EHM_LDS_BUDGET / lds, reg_wg));
    long long grid = std::min<long long>((long long)P->num_cu * wg_per_cu, (n_items + nw - 1) / nw);
    cfg.api = api;
    cfg.L.grid = (int)std::max(1LL, grid);
    cfg.L.threads = (int)(api->threads_per_lp * nw);
    cfg.L.lds_bytes = lds;
    cfg.L.wave_doubles = (int)wave;
    cfg.L.stream = P->stream;
    cfg.L.wc_lds = wc_lds;
    return EHM_OK;
}

// stable counting sort of a batch by commutation index: order[k] = original position of the
// k-th instance of the sorted batch, seg[d] = first sorted position of commutation d
static void sort_by_commutation(int nd, int64_t n, const int32_t* didx, std::vector<int64_t>& order,
                                std::vector<int32_t>& seg) {
    seg.assign((size_t)nd + 1, 0);
    for (int64_t k = 0; k < n; ++k) seg[(size_t)didx[k] + 1]++;
    for (int d = 0; d < nd; ++d) seg[(size_t)d + 1] += seg[(size_t)d];
    std::vector<int32_t> pos(seg.begin(), seg.end() - 1);
    order.resize((size_t)n);
    for (int64_t k = 0; k < n; ++k) order[(size_t)pos[(size_t)didx[k]]++] = k;
}

// ---- eliminated columns (ehm_ipm2.h) ---------------------------------------
#define EHM_ELIM_MAX_ROWS 16    // rows one eliminated column may have (Shared::LE)

// Largest trailing range [nd0, n) of the z-columns of which, in EVERY commutation, every row holds
// at most one entry and every column 1..EHM_ELIM_MAX_ROWS entries -- the epigraph variables of an
// infinity-norm cost (lib/mpc_library.py:530-560).  G: [nd][m][n] row-major.  nd0 = n: none.
static void elim_detect(const double* G, int nd, int m, int n, int p, int n_u, int& nd0, int& LE) {
    nd0 = n;
    LE = 0;
    std::vector<int> rowcnt((size_t)nd * m, 0);
    int le = 0;
    for (int c = n - 1; c >= n_u; --c) {
        bool ok = true;
        int colmax = 0;
        for (int k = 0; k < nd && ok; ++k) {
            int nnz = 0;
            for (int i = 0; i < m; ++i)
                if (G[((size_t)k * m + i) * n + c] != 0.0) {
                    ++nnz;
                    if (rowcnt[(size_t)k * m + i] >= 1) ok = false;
                }
            if (nnz < 1 || nnz > EHM_ELIM_MAX_ROWS) ok = false;
            colmax = std::max(colmax, nnz);
        }
        if (!ok) break;
        for (int k = 0; k < nd; ++k)
            for (int i = 0; i < m; ++i)
                if (G[((size_t)k * m + i) * n + c] != 0.0) rowcnt[(size_t)k * m + i]++;
        le = std::max(le, colmax);
        // the transform of the block's weight rows runs as (column, weight) tasks of one wavefront
        if ((n - c) * p > 256) break;
        nd0 = c;
        LE = le;
    }
    if (n - nd0 < 2) {      // not worth a second code path
        nd0 = n;
        LE = 0;
    }
    LE = (LE + 3) & ~3;      // the kernels gather the rows of a column four at a time
}

static size_t elim_tot4(int m, int p, int lda, int nd0, int nE, int LE) {
    const size_t tab = (size_t)nE * LE + ((size_t)nE * LE + m + 1) / 2;
    return (((size_t)(nd0 + p + 3) * lda + tab) + 1) & ~(size_t)1;
}

// One commutation's image without the columns [nd0, n) (layout: DevProblem::Wc4).  False when the
// block is not of the expected shape (a row with two entries, a column with too many).
static bool elim_image(const double* Gk, const double* Sk, int m, int n, int p, int lda, int nd0,
                       int LE, double* base, size_t tot4) {
    const int nE = n - nd0;
    std::fill(base, base + tot4, 0.0);
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < nd0; ++j) base[(size_t)j * lda + i] = Gk[(size_t)i * n + j];
        for (int q = 0; q < p; ++q) base[(size_t)(nd0 + q) * lda + i] = -Sk[(size_t)i * p + q];
        base[(size_t)(nd0 + p) * lda + i] = -1.0;
    }
    double* aE = base + (size_t)(nd0 + p + 2) * lda;
    double* eval = base + (size_t)(nd0 + p + 3) * lda;
    int32_t* erow = reinterpret_cast<int32_t*>(eval + (size_t)nE * LE);
    int32_t* eidx = erow + (size_t)nE * LE;
    std::vector<int> fill((size_t)nE, 0);
    for (int i = 0; i < m; ++i) {
        int found = -1;
        for (int e = 0; e < nE; ++e) {
            const double v = Gk[(size_t)i * n + nd0 + e];
            if (v == 0.0) continue;
            if (found >= 0 || fill[(size_t)e] >= LE) return false;
            found = e;
            aE[i] = v;
            eidx[i] = e;
            eval[(size_t)e * LE + fill[(size_t)e]] = v;
            erow[(size_t)e * LE + fill[(size_t)e]] = i;
            fill[(size_t)e]++;
        }
    }
    for (int e = 0; e < nE; ++e)
        if (fill[(size_t)e] == 0) return false;     // an empty column: Delta_e would be 0
    return true;
}

// no eliminated columns: the shared-block kernels read the full image
static void elim_off(ehm_problem* P) {
    P->dp.nd0 = P->dp.n;
    P->dp.LE4 = 0;
    P->dp.Wc4 = P->dp.Wc2;
    P->dp.lda4 = P->dp.lda2;
    P->dp.ncw4 = P->dp.ncw2;
    P->dp.tot4 = P->dp.ncw2 * P->dp.lda2;
}

extern "C" {

const char* ehm_last_error(void) { return g_err.c_str(); }
const char* ehm_version(void) { return "ehmpc 0.1 (gfx950)"; }

int ehm_problem_create(const ehm_problem_desc* d, int device, ehm_problem** out) {
    if (!d || !out) return fail(EHM_E_INVALID, "null argument");
    *out = nullptr;
    if (d->n < 1 || d->m < 1 || d->p < 1 || d->n_u < 1 || d->n_delta < 1)
        return fail(EHM_E_INVALID, "non-positive dimension");
    if (d->p > EHM_MAX_P || d->n + d->p + 1 > EHM_MAX_N || d->m + d->p + 3 > EHM_MAX_M)
        return fail(EHM_E_INVALID,
                    "unsupported size: need n+p+1 <= %d, m+p+3 <= %d, p <= %d (got n=%d m=%d p=%d)",
                    EHM_MAX_N, EHM_MAX_M, EHM_MAX_P, d->n, d->m, d->p);
    if (d->n_u > d->n || rec_doubles(d->p, d->n_u) > NODE_LDS_DOUBLES - 8)
        return fail(EHM_E_INVALID, "unsupported n_u=%d", d->n_u);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(EHM_E_NO_DEVICE, "no HIP device available (libehmpc has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return fail(EHM_E_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device), EHM_E_NO_DEVICE);
    ehm_problem* P = new ehm_problem();
    P->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) P->num_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreate(&P->stream), EHM_E_NO_DEVICE);
    (void)hipEventCreate(&P->bev[0]);
    (void)hipEventCreate(&P->bev[1]);
    const int n = d->n, m = d->m, p = d->p, nd = d->n_delta;
    // transpose to column-major per commutation
    const size_t nG = (size_t)nd * n * m, nS = (size_t)nd * p * m, nw = (size_t)nd * m;
    std::vector<double> host(nG + nS + nw + n);
    double* Gt = host.data();
    double* St = Gt + nG;
    double* w = St + nS;
    double* c = w + nw;
    for (int k = 0; k < nd; ++k)
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < n; ++j)
                Gt[((size_t)k * n + j) * m + i] = d->G[((size_t)k * m + i) * n + j];
            for (int q = 0; q < p; ++q)
                St[((size_t)k * p + q) * m + i] = d->S[((size_t)k * m + i) * p + q];
            w[(size_t)k * m + i] = d->w[(size_t)k * m + i];
        }
    for (int j = 0; j < n; ++j) c[j] = d->c[j];
    int rc = P->consts.ensure(host.size() * sizeof(double));
    if (rc) { delete P; return rc; }
    HIP_TRY(hipMemcpy(P->consts.ptr, host.data(), host.size() * sizeof(double),
                      hipMemcpyHostToDevice), EHM_E_HIP);
    P->dp.n = n; P->dp.m = m; P->dp.p = p; P->dp.n_u = d->n_u; P->dp.n_delta = nd;
    P->dp.Gt = P->consts.as<double>();
    P->dp.St = P->dp.Gt + nG;
    P->dp.w = P->dp.St + nS;
    P->dp.c = P->dp.w + nw;
    P->dp.eps_a = d->eps_a;
    P->dp.eps_r = d->eps_r;
    {
        // LDS image of the constant LP block per commutation: [G | -S | -1 | 0], column-major
        const int lda = m | 1, ncw = n + p + 2;
        std::vector<double> img((size_t)nd * ncw * lda, 0.0);
        for (int k = 0; k < nd; ++k) {
            double* base = img.data() + (size_t)k * ncw * lda;
            for (int i = 0; i < m; ++i) {
                for (int j = 0; j < n; ++j)
                    base[(size_t)j * lda + i] = d->G[((size_t)k * m + i) * n + j];
                for (int q = 0; q < p; ++q)
                    base[(size_t)(n + q) * lda + i] = -d->S[((size_t)k * m + i) * p + q];
                base[(size_t)(n + p) * lda + i] = -1.0;
            }
        }
        rc = P->wc2.ensure(img.size() * sizeof(double));
        if (rc) { delete P; return rc; }
        HIP_TRY(hipMemcpy(P->wc2.ptr, img.data(), img.size() * sizeof(double),
                          hipMemcpyHostToDevice), EHM_E_HIP);
        P->dp.Wc2 = P->wc2.as<double>();
        P->dp.lda2 = lda;
        P->dp.ncw2 = ncw;
        // the same image without the eliminated columns (EHM_SPARSE=0: keep every column)
        elim_off(P);
        int nd0 = n, LE = 0;
        const char* es = getenv("EHM_SPARSE");
        if (!(es && atoi(es) == 0)) elim_detect(d->G, nd, m, n, p, d->n_u, nd0, LE);
        if (nd0 < n) {
            const int nE = n - nd0;
            // Column stride of the reduced image: = 2 (mod 8) doubles.  The LDS serves a
            // ds_read_b64 in two groups of 32 lanes over 32 bank pairs (MI355X_MICROARCH.md):
            // the matrix-core operands (lane -> column l & 15, row l >> 4) and the column
            // products (lane -> column block, K-slice) both need (column stride) x (small integer)
            // to spread over the bank pairs, which an ODD stride (round 1: m | 1) does not do --
            // 18 % of the LDS-active cycles of the headline kernel were bank conflicts.  The row
            // products (lane = row) are conflict free at any stride.  EHM_LDA4=odd: the old stride.
            int lda4 = lda;
            {
                const char* e = getenv("EHM_LDA4");
                if (!(e && !strcmp(e, "odd"))) {
                    lda4 = m;
                    while ((lda4 & 7) != 2) ++lda4;
                }
            }
            const size_t tot4 = elim_tot4(m, p, lda4, nd0, nE, LE);
            std::vector<double> img4((size_t)nd * tot4);
            bool ok = true;
            for (int k = 0; k < nd && ok; ++k)
                ok = elim_image(d->G + (size_t)k * m * n, d->S + (size_t)k * m * p, m, n, p, lda4,
                                nd0, LE, img4.data() + (size_t)k * tot4, tot4);
            if (ok) {
                rc = P->wc4.ensure(img4.size() * sizeof(double));
                if (rc) { delete P; return rc; }
                HIP_TRY(hipMemcpy(P->wc4.ptr, img4.data(), img4.size() * sizeof(double),
                                  hipMemcpyHostToDevice), EHM_E_HIP);
                P->dp.Wc4 = P->wc4.as<double>();
                P->dp.lda4 = lda4;
                P->dp.ncw4 = nd0 + p + 3;
                P->dp.tot4 = (int)tot4;
                P->dp.nd0 = nd0;
                P->dp.LE4 = LE;
            }
        }
    }
    P->dp.Wr3 = nullptr;
    P->dp.mpad3 = 0;
    if (n + p + 1 > EHM_V1_MAX_N || m + p + 3 > EHM_V1_MAX_M) {
        // row-major image for the wide kernels: [G | -S | -1 | 0 ...], 64 columns, zero rows
        // up to a multiple of 64
        const int mpad = (m + 63) & ~63;
        std::vector<double> img((size_t)nd * mpad * 64, 0.0);
        for (int k = 0; k < nd; ++k) {
            double* base = img.data() + (size_t)k * mpad * 64;
            for (int i = 0; i < m; ++i) {
                double* rowp = base + (size_t)i * 64;
                for (int j = 0; j < n; ++j) rowp[j] = d->G[((size_t)k * m + i) * n + j];
                for (int q = 0; q < p; ++q) rowp[n + q] = -d->S[((size_t)k * m + i) * p + q];
                rowp[n + p] = -1.0;
            }
        }
        rc = P->wr3.ensure(img.size() * sizeof(double));
        if (rc) { delete P; return rc; }
        HIP_TRY(hipMemcpy(P->wr3.ptr, img.data(), img.size() * sizeof(double),
                          hipMemcpyHostToDevice), EHM_E_HIP);
        P->dp.Wr3 = P->wr3.as<double>();
        P->dp.mpad3 = mpad;
    }
    if (const char* e = getenv("EHM_SOLVER")) P->solver_gen = (atoi(e) == 1) ? 1 : 2;
    if (const char* e = getenv("EHM_DECIDE_FULL")) P->decide_full = atoi(e) ? 1 : 0;
    if (const char* e = getenv("EHM_MID_FIRST")) P->mid_first = atoi(e) ? 1 : 0;
    P->delta_len = d->delta_len;
    if (d->deltas && d->delta_len > 0)
        P->deltas.assign(d->deltas, d->deltas + (size_t)nd * d->delta_len);
    else
        P->deltas.assign((size_t)nd * std::max(1, d->delta_len), 1);
    HIP_TRY(hipMalloc((void**)&P->d_cnt, sizeof(DevCounters)), EHM_E_HIP);
    DevCounters zero{};
    zero.min_margin_bits = 0x7FF0000000000000ULL;   // +inf
    HIP_TRY(hipMemcpy(P->d_cnt, &zero, sizeof zero, hipMemcpyHostToDevice), EHM_E_HIP);
    P->lds_point = lds_bytes_for(P->dp, LP_FEAS, 16);
    P->lds_simplex = lds_bytes_for(P->dp, LP_SLACK, NODE_LDS_DOUBLES);
    P->lds_expand = lds_bytes_for(P->dp, LP_POINT, NODE_LDS_DOUBLES);
    const size_t lds_max = std::max(P->lds_point, P->lds_simplex);
    if (lds_max > 160 * 1024 || n + p + 1 > EHM_V1_MAX_N || m + p + 3 > EHM_V1_MAX_M) {
        if (P->solver_gen == 1) {
            ehm_problem_destroy(P);
            return fail(EHM_E_INVALID, "LP too large for the generation-1 kernels (%zu bytes)",
                        lds_max);
        }
    } else {
        const void* kernels[] = {(const void*)k_point_batch, (const void*)k_simplex_batch,
                                 (const void*)k_lcss_decide, (const void*)k_lcss_expand,
                                 (const void*)k_vertex_solve};
        for (const void* k : kernels)
            HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_max), EHM_E_HIP);
        P->v1_ok = true;
    }
    *out = P;
    return EHM_OK;
}

// Replaces the constant blocks of commutation slots [first, first + count): G [count][m][n],
// w [count][m], S [count][m][p] (row-major, as in ehm_problem_desc).  Every image the kernels
// read is rebuilt for those slots.  This is what lets a caller use the commutation table as a
// CACHE of problems it generates on the fly -- the prefix relaxations of a branch-and-bound over
// mode sequences (explicit_hybrid_mpc_amd/sequences.py) -- instead of a fixed enumeration.
int ehm_problem_update_blocks(ehm_problem* P, int32_t first, int32_t count, const double* G,
                              const double* w, const double* S) {
    if (!P || !G || !w || !S || first < 0 || count < 0 || first + count > P->dp.n_delta)
        return fail(EHM_E_INVALID, "bad argument");
    if (P->active_run) return fail(EHM_E_INVALID, "a partition run is active on this handle");
    if (count == 0) return EHM_OK;
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    const int n = P->dp.n, m = P->dp.m, p = P->dp.p, nd = P->dp.n_delta;
    const size_t nG = (size_t)nd * n * m, nS = (size_t)nd * p * m;
    double* dGt = P->consts.as<double>();
    double* dSt = dGt + nG;
    double* dw = dSt + nS;
    const int lda = P->dp.lda2, ncw = P->dp.ncw2, mpad = P->dp.mpad3;
    std::vector<double> gt((size_t)count * n * m), st((size_t)count * p * m),
        img2((size_t)count * ncw * lda, 0.0), img3(P->dp.Wr3 ? (size_t)count * mpad * 64 : 0, 0.0);
    for (int k = 0; k < count; ++k) {
        double* b2 = img2.data() + (size_t)k * ncw * lda;
        double* b3 = img3.empty() ? nullptr : img3.data() + (size_t)k * mpad * 64;
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < n; ++j) {
                const double v = G[((size_t)k * m + i) * n + j];
                gt[((size_t)k * n + j) * m + i] = v;
                b2[(size_t)j * lda + i] = v;
                if (b3) b3[(size_t)i * 64 + j] = v;
            }
            for (int q = 0; q < p; ++q) {
                const double v = S[((size_t)k * m + i) * p + q];
                st[((size_t)k * p + q) * m + i] = v;
                b2[(size_t)(n + q) * lda + i] = -v;
                if (b3) b3[(size_t)i * 64 + n + q] = -v;
            }
            b2[(size_t)(n + p) * lda + i] = -1.0;
            if (b3) b3[(size_t)i * 64 + n + p] = -1.0;
        }
    }
    HIP_TRY(hipMemcpy(dGt + (size_t)first * n * m, gt.data(), gt.size() * 8, hipMemcpyHostToDevice),
            EHM_E_HIP);
    HIP_TRY(hipMemcpy(dSt + (size_t)first * p * m, st.data(), st.size() * 8, hipMemcpyHostToDevice),
            EHM_E_HIP);
    HIP_TRY(hipMemcpy(dw + (size_t)first * m, w, (size_t)count * m * 8, hipMemcpyHostToDevice),
            EHM_E_HIP);
    HIP_TRY(hipMemcpy(P->wc2.as<double>() + (size_t)first * ncw * lda, img2.data(), img2.size() * 8,
                      hipMemcpyHostToDevice), EHM_E_HIP);
    if (!img3.empty())
        HIP_TRY(hipMemcpy(P->wr3.as<double>() + (size_t)first * mpad * 64, img3.data(),
                          img3.size() * 8, hipMemcpyHostToDevice), EHM_E_HIP);
    if (P->dp.nd0 < n) {
        // the image without the eliminated columns; a block that does not have the shape the
        // handle was created with (a row with two of those columns) switches the elimination off
        // for the whole handle -- every kernel then reads the full image again
        const size_t tot4 = (size_t)P->dp.tot4;
        std::vector<double> img4((size_t)count * tot4);
        bool ok = true;
        for (int k = 0; k < count && ok; ++k)
            ok = elim_image(G + (size_t)k * m * n, S + (size_t)k * m * p, m, n, p, P->dp.lda4, P->dp.nd0,
                            P->dp.LE4, img4.data() + (size_t)k * tot4, tot4);
        if (ok)
            HIP_TRY(hipMemcpy(P->wc4.as<double>() + (size_t)first * tot4, img4.data(),
                              img4.size() * 8, hipMemcpyHostToDevice), EHM_E_HIP);
        else
            elim_off(P);
    }
    return EHM_OK;
}

int ehm_problem_destroy(ehm_problem* P) {
    if (!P) return EHM_OK;
    (void)hipSetDevice(P->device);
    if (P->stream) (void)hipStreamSynchronize(P->stream);
    // an unfinished run, or a finished tree the caller still holds, outlives the handle as an
    // orphan: its own buffers stay valid until ehm_tree_destroy, everything else is refused
    for (ehm_tree* T : P->trees) {
        T->prob = nullptr;
        T->run.active = false;
    }
    P->trees.clear();
    P->active_run = nullptr;
    P->consts.release();
    P->wc2.release();
    P->wc4.release();
    P->wr3.release();
    P->quad.release();
    P->seg.release();
    P->pool_cache.rec.release(); P->pool_cache.left.release(); P->pool_cache.didx.release();
    P->pool_cache.depth.release(); P->pool_cache.flags.release(); P->pool_cache.tstar.release();
    P->pool_cache.grad.release(); P->pool_cache.wit.release();
    P->fr_a.release(); P->fr_b.release(); P->open_flag.release(); P->open_list.release();
    P->d_count.release();
    P->pq_slots.release(); P->pq_ctl.release();
    P->in0.release(); P->in1.release(); P->in2.release();
    P->ex_stage.release(); P->ex_tmp.release();
    P->out0.release(); P->out1.release(); P->out2.release(); P->out3.release();
    if (P->d_cnt) (void)hipFree(P->d_cnt);
    for (hipEvent_t e : P->bev)
        if (e) (void)hipEventDestroy(e);
    if (P->stream) (void)hipStreamDestroy(P->stream);
    delete P;
    return EHM_OK;
}

int ehm_problem_set_eps(ehm_problem* P, double eps_a, double eps_r) {
    if (!P) return fail(EHM_E_INVALID, "null problem");
    P->dp.eps_a = eps_a;
    P->dp.eps_r = eps_r;
    return EHM_OK;
}

int ehm_problem_set_quadratic(ehm_problem* P, const double* H, const double* F, const double* f0,
                              const double* C, const double* c1, const double* c0) {
    if (!P || !H || !F || !f0 || !C || !c1 || !c0) return fail(EHM_E_INVALID, "null argument");
    if (!P->v1_ok)
        return fail(EHM_E_INVALID,
                    "quadratic costs need the one-wavefront kernels: n+p+1 <= %d, m+p+3 <= %d",
                    EHM_V1_MAX_N, EHM_V1_MAX_M);
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    const int n = P->dp.n, p = P->dp.p, nd = P->dp.n_delta;
    const size_t nH = (size_t)nd * n * n, nF = (size_t)nd * p * n, nf = (size_t)nd * n;
    const size_t nC = (size_t)nd * p * p, n1 = (size_t)nd * p;
    std::vector<double> host(nH + nF + nf + nC + n1 + nd);
    double* h = host.data();
    // symmetrise H and C (the kernels read one triangle's worth through row j, column k)
    for (int k = 0; k < nd; ++k)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                h[((size_t)k * n + i) * n + j] =
                    0.5 * (H[((size_t)k * n + i) * n + j] + H[((size_t)k * n + j) * n + i]);
    double* Ft = h + nH;
    for (int k = 0; k < nd; ++k)
        for (int j = 0; j < n; ++j)
            for (int q = 0; q < p; ++q)
                Ft[((size_t)k * p + q) * n + j] = F[((size_t)k * n + j) * p + q];
    double* f0d = Ft + nF;
    for (size_t i = 0; i < nf; ++i) f0d[i] = f0[i];
    double* Cd = f0d + nf;
    for (int k = 0; k < nd; ++k)
        for (int i = 0; i < p; ++i)
            for (int j = 0; j < p; ++j)
                Cd[((size_t)k * p + i) * p + j] =
                    0.5 * (C[((size_t)k * p + i) * p + j] + C[((size_t)k * p + j) * p + i]);
    double* c1d = Cd + nC;
    for (size_t i = 0; i < n1; ++i) c1d[i] = c1[i];
    double* c0d = c1d + n1;
    for (int k = 0; k < nd; ++k) c0d[k] = c0[k];
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    int rc = P->quad.ensure(host.size() * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipMemcpy(P->quad.ptr, host.data(), host.size() * sizeof(double),
                      hipMemcpyHostToDevice), EHM_E_HIP);
    P->dp.Hq = P->quad.as<double>();
    P->dp.Fq = P->dp.Hq + nH;
    P->dp.f0q = P->dp.Fq + nF;
    P->dp.Cq = P->dp.f0q + nf;
    P->dp.c1q = P->dp.Cq + nC;
    P->dp.c0q = P->dp.c1q + n1;
    P->quadratic = true;
    elim_off(P);        // the Hessian couples the columns: nothing is eliminated
    // shared-block kernels with the quadratic block when an instance holds the largest problem
    // (the suboptimality test), else the one-wavefront kernels
    P->k2q_ok = k2_pick(n + p + 1, lp_slots(P->dp.m, p + 3), true) != nullptr;
    if (!P->k2q_ok || getenv("EHM_SOLVER")) {
        const char* e = getenv("EHM_SOLVER");
        P->solver_gen = (P->k2q_ok && e && atoi(e) == 2) ? 2 : 1;
    } else {
        P->solver_gen = 2;
    }
    return EHM_OK;
}

int ehm_problem_set_solver(ehm_problem* P, int generation) {
    if (!P) return fail(EHM_E_INVALID, "null problem");
    if (generation != 1 && generation != 2) return fail(EHM_E_INVALID, "solver generation 1 or 2");
    if (generation == 2 && P->quadratic && !P->k2q_ok)
        return fail(EHM_E_INVALID,
                    "no shared-block instance with the quadratic block holds this problem");
    if (generation == 1 && !P->v1_ok)
        return fail(EHM_E_INVALID, "the generation-1 kernels do not fit this problem in LDS");
    P->solver_gen = generation;
    return EHM_OK;
}

// wave-primitive self test of the k2 instances: out[5] per instance (see k2_selftest)
int ehm_abi_sizes(int64_t* sizes, int32_t n) {
    const int64_t all[] = {(int64_t)sizeof(ehm_problem_desc), (int64_t)sizeof(ehm_run_opts),
                           (int64_t)sizeof(ehm_node_init),    (int64_t)sizeof(ehm_progress),
                           (int64_t)sizeof(ehm_tree_info),    (int64_t)sizeof(ehm_counters)};
    const int32_t known = (int32_t)(sizeof all / sizeof all[0]);
    for (int32_t k = 0; sizes && k < n && k < known; ++k) sizes[k] = all[k];
    return known;
}

int ehm_selftest(int device, double* out, int32_t max_instances, int32_t* n_instances) {
    if (!out || !n_instances) return fail(EHM_E_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(EHM_E_NO_DEVICE, "no HIP device %d (libehmpc has no CPU fallback)", device);
    HIP_TRY(hipSetDevice(device), EHM_E_NO_DEVICE);
    double* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_out, 5 * sizeof(double)), EHM_E_HIP);
    int k = 0;
    for (k2_getter g : g_k2_getters) {
        if (k >= max_instances) break;
        g()->selftest(nullptr, d_out);
        HIP_TRY(hipDeviceSynchronize(), EHM_E_HIP);
        HIP_TRY(hipMemcpy(out + 5 * k, d_out, 5 * sizeof(double), hipMemcpyDeviceToHost), EHM_E_HIP);
        ++k;
    }
    *n_instances = k;
    (void)hipFree(d_out);
    return EHM_OK;
}

int ehm_problem_set_option(ehm_problem* P, const char* name, double value) {
    if (!P || !name) return fail(EHM_E_INVALID, "null argument");
    if (!strcmp(name, "solver")) return ehm_problem_set_solver(P, (int)value);
    if (!strcmp(name, "decide_full")) {
        P->decide_full = value != 0.0;
        return EHM_OK;
    }
    if (!strcmp(name, "timing")) {
        P->hy_timing = value != 0.0;
        return EHM_OK;
    }
    if (!strcmp(name, "mid_first")) {
        P->mid_first = value != 0.0;
        return EHM_OK;
    }
    if (!strcmp(name, "inherit_witness")) {
        P->inherit_wit = value != 0.0;
        return EHM_OK;
    }
    if (!strcmp(name, "work_first")) {
        P->work_first = value != 0.0;
        return EHM_OK;
    }
    if (!strcmp(name, "check_witness")) {
        P->check_witness = value != 0.0;
        return EHM_OK;
    }
    if (!strcmp(name, "budget_keep")) {
        P->budget_keep = value != 0.0;
        return EHM_OK;
    }
    if (!strcmp(name, "share_midpoints")) {
        P->share_mid = value != 0.0;
        return EHM_OK;
    }
    if (!strcmp(name, "any_admissible")) {
        if (value < 0.0 || value > 2e9) return fail(EHM_E_INVALID, "any_admissible: seed + 1, or 0");
        // the path codes the hashed draws read are allocated when a run begins (hy_begin)
        if (P->active_run)
            return fail(EHM_E_INVALID, "any_admissible: not while a partition run is active");
        P->any_admissible = (int)value;
        return EHM_OK;
    }
    return fail(EHM_E_INVALID, "unknown option '%s'", name);
}

int ehm_sync(ehm_problem* P) {
    if (!P) return fail(EHM_E_INVALID, "null problem");
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    return EHM_OK;
}

void* ehm_stream(ehm_problem* P) { return P ? (void*)P->stream : nullptr; }

int ehm_stats(ehm_problem* P, ehm_counters* out) {
    if (!P || !out) return fail(EHM_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    DevCounters c;
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    HIP_TRY(hipMemcpy(&c, P->d_cnt, sizeof c, hipMemcpyDeviceToHost), EHM_E_HIP);
    out->lp_solves = (int64_t)c.lp_solves;
    out->ipm_iters = (int64_t)c.ipm_iters;
    out->stalled = (int64_t)c.stalled;
    out->kernel_launches = P->launches;
    out->fallbacks = P->fallbacks;
    out->slivers = P->slivers;
    for (int k = 0; k < 2; ++k) {
        out->batch_seconds[k] = P->batch_seconds[k];
        out->batch_launches[k] = P->batch_launches[k];
    }
    return EHM_OK;
}

// ---- batched oracles --------------------------------------------------------------------
static int point_batch(ehm_problem* P, int64_t n_inst, const double* theta,
                       const int32_t* didx_host, int feas, double* J, double* u0,
                       int32_t* status, int32_t* iters) {
    if (n_inst == 0) return EHM_OK;
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    const int p = P->dp.p, n_u = P->dp.n_u;
    int rc;
    if ((rc = P->in0.ensure((size_t)n_inst * p * sizeof(double)))) return rc;
    if ((rc = P->in1.ensure((size_t)n_inst * sizeof(int32_t)))) return rc;
    if ((rc = P->out0.ensure((size_t)n_inst * sizeof(double)))) return rc;
    if ((rc = P->out1.ensure((size_t)n_inst * n_u * sizeof(double)))) return rc;
    if ((rc = P->out2.ensure((size_t)n_inst * 2 * sizeof(int32_t)))) return rc;
    HIP_TRY(hipMemcpyAsync(P->in0.ptr, theta, (size_t)n_inst * p * sizeof(double),
                           hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(P->in1.ptr, didx_host, (size_t)n_inst * sizeof(int32_t),
                           hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
    int32_t* d_status = P->out2.as<int32_t>();
    int32_t* d_iters = d_status + n_inst;
    if (P->solver_gen == 2) {
        // instances travel sorted by commutation (one constant block in LDS per run)
        const int nd = P->dp.n_delta;
        std::vector<int64_t> order;
        std::vector<int32_t> seg;
        sort_by_commutation(nd, n_inst, didx_host, order, seg);
        std::vector<double> th((size_t)n_inst * p);
        for (int64_t k = 0; k < n_inst; ++k)
            std::memcpy(&th[(size_t)k * p], theta + (size_t)order[(size_t)k] * p, p * sizeof(double));
        if ((rc = P->seg.ensure(seg.size() * sizeof(int32_t)))) return rc;
        HIP_TRY(hipMemcpyAsync(P->in0.ptr, th.data(), th.size() * sizeof(double),
                               hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(P->seg.ptr, seg.data(), seg.size() * sizeof(int32_t),
                               hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
        K2Cfg cfg;
        if ((rc = k2_config(P, feas ? LP_FEAS : LP_POINT, feas ? LP_FEAS : LP_POINT, n_inst, cfg)))
            return rc;
        (void)hipEventRecord(P->bev[0], P->stream);
        cfg.api->point(cfg.L, P->dp, (long long)n_inst, P->in0.as<double>(),
                       P->seg.as<int32_t>(), feas, P->out0.as<double>(), P->out1.as<double>(),
                       d_status, d_iters, P->d_cnt, K2Gather{});
        (void)hipEventRecord(P->bev[1], P->stream);
        P->launches++;
        HIP_TRY(hipGetLastError(), EHM_E_HIP);
        std::vector<double> Js((size_t)n_inst), us(u0 ? (size_t)n_inst * n_u : 0);
        std::vector<int32_t> sts((size_t)n_inst), its((size_t)n_inst);
        HIP_TRY(hipMemcpyAsync(Js.data(), P->out0.ptr, (size_t)n_inst * sizeof(double),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
        if (u0)
            HIP_TRY(hipMemcpyAsync(us.data(), P->out1.ptr, (size_t)n_inst * n_u * sizeof(double),
                                   hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(sts.data(), d_status, (size_t)n_inst * sizeof(int32_t),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(its.data(), d_iters, (size_t)n_inst * sizeof(int32_t),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
        {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, P->bev[0], P->bev[1]) == hipSuccess) {
                P->batch_seconds[0] += 1e-3 * ms;
                P->batch_launches[0]++;
            }
        }
        std::vector<int64_t> bad;
        for (int64_t k = 0; k < n_inst; ++k) {
            const int64_t o = order[(size_t)k];
            J[o] = Js[(size_t)k];
            if (u0) std::memcpy(u0 + (size_t)o * n_u, &us[(size_t)k * n_u], n_u * sizeof(double));
            if (status) status[o] = sts[(size_t)k];
            if (iters) iters[o] = its[(size_t)k];
            if (sts[(size_t)k] != 0) bad.push_back(o);
        }
        // same safety net as in simplex_batch (an infeasible instance stays "stalled")
        if (!bad.empty() && P->v1_ok && (int64_t)bad.size() * 4 <= n_inst) {
            const int64_t nb = (int64_t)bad.size();
            std::vector<double> t2((size_t)nb * p), J2((size_t)nb), u2((size_t)nb * n_u);
            std::vector<int32_t> d2((size_t)nb), s2((size_t)nb), i2((size_t)nb);
            for (int64_t k = 0; k < nb; ++k) {
                std::memcpy(&t2[(size_t)k * p], theta + (size_t)bad[(size_t)k] * p, p * sizeof(double));
                d2[(size_t)k] = didx_host[bad[(size_t)k]];
            }
            P->solver_gen = 1;
            rc = point_batch(P, nb, t2.data(), d2.data(), feas, J2.data(), u2.data(), s2.data(),
                             i2.data());
            P->solver_gen = 2;
            if (rc) return rc;
            for (int64_t k = 0; k < nb; ++k) {
                if (s2[(size_t)k] != 0) continue;
                const int64_t o = bad[(size_t)k];
                J[o] = J2[(size_t)k];
                if (u0) std::memcpy(u0 + (size_t)o * n_u, &u2[(size_t)k * n_u], n_u * sizeof(double));
                if (status) status[o] = 0;
                if (iters) iters[o] = i2[(size_t)k];
                P->fallbacks++;
            }
        }
        return EHM_OK;
    }
    hipLaunchKernelGGL(k_point_batch, dim3(grid_for(P, n_inst)), dim3(64), P->lds_point,
                       P->stream, P->dp, (long long)n_inst, P->in0.as<double>(),
                       P->in1.as<int32_t>(), feas, P->out0.as<double>(), P->out1.as<double>(),
                       d_status, d_iters, P->d_cnt, K2Gather{});
    P->launches++;
    HIP_TRY(hipGetLastError(), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(J, P->out0.ptr, (size_t)n_inst * sizeof(double),
                           hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
    if (u0)
        HIP_TRY(hipMemcpyAsync(u0, P->out1.ptr, (size_t)n_inst * n_u * sizeof(double),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
    if (status)
        HIP_TRY(hipMemcpyAsync(status, d_status, (size_t)n_inst * sizeof(int32_t),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
    if (iters)
        HIP_TRY(hipMemcpyAsync(iters, d_iters, (size_t)n_inst * sizeof(int32_t),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    return EHM_OK;
}

int ehm_solve_ptd_batch(ehm_problem* P, int64_t n_inst, const double* theta,
                        const uint8_t* delta, double* J, double* u0, int32_t* status,
                        int32_t* iters) {
    if (!P || !theta || !J || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    std::vector<int32_t> didx;
    int rc = map_deltas(P, n_inst, delta, didx);
    if (rc) return rc;
    return point_batch(P, n_inst, theta, didx.data(), 0, J, u0, status, iters);
}

// feasibility margin: the phase-one optimum is compared against a tolerance that is far
// above the solver accuracy (1e-10 relative) and far below any constraint scale
#define EHM_FEAS_TOL 1e-8
#define EHM_SLIVER_TOL 1e-7
// optimal values of different commutations closer than this (relative to 1+|value|) are
// ties, broken by enumeration order (DESIGN.md "canonical commutation rule")
#define EHM_TIE_TOL 1e-6

int ehm_feas_ptd_batch(ehm_problem* P, int64_t n_inst, const double* theta,
                       const uint8_t* delta, uint8_t* feasible, double* tau) {
    if (!P || !theta || !feasible || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    std::vector<int32_t> didx;
    int rc = map_deltas(P, n_inst, delta, didx);
    if (rc) return rc;
    std::vector<double> t((size_t)n_inst);
    rc = point_batch(P, n_inst, theta, didx.data(), 1, t.data(), nullptr, nullptr, nullptr);
    if (rc) return rc;
    for (int64_t k = 0; k < n_inst; ++k) {
        feasible[k] = (t[(size_t)k] <= EHM_FEAS_TOL) ? 1 : 0;
        if (tau) tau[k] = t[(size_t)k];
    }
    return EHM_OK;
}

static int simplex_batch(ehm_problem* P, int64_t n_inst, const double* R, const double* Vbar,
                         const int32_t* didx_host, int slack, double* obj, double* alpha,
                         int32_t* status) {
    if (n_inst == 0) return EHM_OK;
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    const int p = P->dp.p;
    const size_t nR = (size_t)(p + 1) * p;
    int rc;
    if ((rc = P->in0.ensure((size_t)n_inst * nR * sizeof(double)))) return rc;
    if ((rc = P->in1.ensure((size_t)n_inst * sizeof(int32_t)))) return rc;
    if ((rc = P->in2.ensure((size_t)n_inst * (p + 1) * sizeof(double)))) return rc;
    if ((rc = P->out0.ensure((size_t)n_inst * sizeof(double)))) return rc;
    if ((rc = P->out1.ensure((size_t)n_inst * (p + 1) * sizeof(double)))) return rc;
    if ((rc = P->out2.ensure((size_t)n_inst * 2 * sizeof(int32_t)))) return rc;
    HIP_TRY(hipMemcpyAsync(P->in0.ptr, R, (size_t)n_inst * nR * sizeof(double),
                           hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(P->in1.ptr, didx_host, (size_t)n_inst * sizeof(int32_t),
                           hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
    if (slack == SX_SLACK)
        HIP_TRY(hipMemcpyAsync(P->in2.ptr, Vbar, (size_t)n_inst * (p + 1) * sizeof(double),
                               hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
    int32_t* d_status = P->out2.as<int32_t>();
    if (P->solver_gen == 2) {
        const int nd = P->dp.n_delta, nv = p + 1;
        std::vector<int64_t> order;
        std::vector<int32_t> seg;
        sort_by_commutation(nd, n_inst, didx_host, order, seg);
        std::vector<double> Rs((size_t)n_inst * nR), Vs(slack == SX_SLACK ? (size_t)n_inst * nv : 0);
        for (int64_t k = 0; k < n_inst; ++k) {
            const int64_t o = order[(size_t)k];
            std::memcpy(&Rs[(size_t)k * nR], R + (size_t)o * nR, nR * sizeof(double));
            if (slack == SX_SLACK)
                std::memcpy(&Vs[(size_t)k * nv], Vbar + (size_t)o * nv, nv * sizeof(double));
        }
        if ((rc = P->seg.ensure(seg.size() * sizeof(int32_t)))) return rc;
        HIP_TRY(hipMemcpyAsync(P->in0.ptr, Rs.data(), Rs.size() * sizeof(double),
                               hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
        if (slack == SX_SLACK)
            HIP_TRY(hipMemcpyAsync(P->in2.ptr, Vs.data(), Vs.size() * sizeof(double),
                                   hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(P->seg.ptr, seg.data(), seg.size() * sizeof(int32_t),
                               hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
        const int kind = (slack == SX_SLACK) ? LP_SLACK
                         : (slack == SX_FEAS) ? LP_FEAS_SIMPLEX : LP_MIN_SIMPLEX;
        K2Cfg cfg;
        if ((rc = k2_config(P, kind, kind, n_inst, cfg))) return rc;
        (void)hipEventRecord(P->bev[0], P->stream);
        cfg.api->simplex(cfg.L, P->dp, (long long)n_inst, P->in0.as<double>(),
                         P->in2.as<double>(), P->seg.as<int32_t>(), slack, P->out0.as<double>(),
                         alpha ? P->out1.as<double>() : (double*)nullptr, d_status,
                         d_status + n_inst, P->d_cnt, K2Gather{});
        (void)hipEventRecord(P->bev[1], P->stream);
        P->launches++;
        HIP_TRY(hipGetLastError(), EHM_E_HIP);
        std::vector<double> objs((size_t)n_inst), als(alpha ? (size_t)n_inst * nv : 0);
        std::vector<int32_t> sts((size_t)n_inst);
        HIP_TRY(hipMemcpyAsync(objs.data(), P->out0.ptr, (size_t)n_inst * sizeof(double),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
        if (alpha)
            HIP_TRY(hipMemcpyAsync(als.data(), P->out1.ptr, (size_t)n_inst * nv * sizeof(double),
                                   hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(sts.data(), d_status, (size_t)n_inst * sizeof(int32_t),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
        {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, P->bev[0], P->bev[1]) == hipSuccess) {
                P->batch_seconds[1] += 1e-3 * ms;
                P->batch_launches[1]++;
            }
        }
        std::vector<int64_t> bad;
        for (int64_t k = 0; k < n_inst; ++k) {
            const int64_t o = order[(size_t)k];
            obj[o] = objs[(size_t)k];
            if (alpha) std::memcpy(alpha + (size_t)o * nv, &als[(size_t)k * nv], nv * sizeof(double));
            if (status) status[o] = sts[(size_t)k];
            if (sts[(size_t)k] != 0) bad.push_back(o);
        }
        // numerical safety net: an LP the shared-block kernels (psi-coordinates) could not bring
        // to tolerance is repeated by the generation-1 kernel (barycentric coordinates, private
        // copy of the LP) -- a few per ten million on hybrid instances
        if (!bad.empty() && P->v1_ok) {
            const int64_t nb = (int64_t)bad.size();
            std::vector<double> R2((size_t)nb * nR), V2((size_t)nb * nv, 0.0), o2((size_t)nb),
                a2((size_t)nb * nv);
            std::vector<int32_t> d2((size_t)nb), s2((size_t)nb);
            for (int64_t k = 0; k < nb; ++k) {
                std::memcpy(&R2[(size_t)k * nR], R + (size_t)bad[(size_t)k] * nR, nR * sizeof(double));
                if (slack == SX_SLACK)
                    std::memcpy(&V2[(size_t)k * nv], Vbar + (size_t)bad[(size_t)k] * nv,
                                nv * sizeof(double));
                d2[(size_t)k] = didx_host[bad[(size_t)k]];
            }
            P->solver_gen = 1;
            rc = simplex_batch(P, nb, R2.data(), V2.data(), d2.data(), slack, o2.data(),
                               a2.data(), s2.data());
            P->solver_gen = 2;
            if (rc) return rc;
            for (int64_t k = 0; k < nb; ++k) {
                if (s2[(size_t)k] != 0) continue;
                const int64_t o = bad[(size_t)k];
                obj[o] = o2[(size_t)k];
                if (alpha)
                    std::memcpy(alpha + (size_t)o * nv, &a2[(size_t)k * nv], nv * sizeof(double));
                if (status) status[o] = 0;
                P->fallbacks++;
            }
        }
        return EHM_OK;
    }
    hipLaunchKernelGGL(k_simplex_batch, dim3(grid_for(P, n_inst)), dim3(64), P->lds_simplex,
                       P->stream, P->dp, (long long)n_inst, P->in0.as<double>(),
                       P->in2.as<double>(), P->in1.as<int32_t>(), slack, P->out0.as<double>(),
                       alpha ? P->out1.as<double>() : (double*)nullptr, d_status,
                       d_status + n_inst, P->d_cnt, K2Gather{});
    P->launches++;
    HIP_TRY(hipGetLastError(), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(obj, P->out0.ptr, (size_t)n_inst * sizeof(double),
                           hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
    if (alpha)
        HIP_TRY(hipMemcpyAsync(alpha, P->out1.ptr, (size_t)n_inst * (p + 1) * sizeof(double),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
    if (status)
        HIP_TRY(hipMemcpyAsync(status, d_status, (size_t)n_inst * sizeof(int32_t),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    return EHM_OK;
}

int ehm_slack_batch(ehm_problem* P, int64_t n_inst, const double* R, const double* Vbar,
                    const uint8_t* delta, double* tstar, double* alpha, int32_t* status) {
    if (!P || !R || !Vbar || !tstar || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    std::vector<int32_t> didx;
    int rc = map_deltas(P, n_inst, delta, didx);
    if (rc) return rc;
    return simplex_batch(P, n_inst, R, Vbar, didx.data(), 1, tstar, alpha, status);
}

// The batched problems with the commutation given as a SLOT INDEX of the table instead of a 0/1
// vector (callers that manage the table themselves, ehm_problem_update_blocks).
// mode: 0 = min over the simplex, 1 = suboptimality-test slack, 2 = phase one over the simplex.
int ehm_simplex_idx_batch(ehm_problem* P, int64_t n_inst, const double* R, const double* Vbar,
                          const int32_t* slot, int32_t mode, double* obj, double* alpha,
                          int32_t* status) {
    if (!P || !R || !slot || !obj || n_inst < 0 || mode < 0 || mode > 2 ||
        (mode == SX_SLACK && !Vbar))
        return fail(EHM_E_INVALID, "bad argument");
    for (int64_t k = 0; k < n_inst; ++k)
        if (slot[k] < 0 || slot[k] >= P->dp.n_delta)
            return fail(EHM_E_INVALID, "instance %lld: slot %d out of range", (long long)k, slot[k]);
    return simplex_batch(P, n_inst, R, Vbar, slot, mode, obj, alpha, status);
}
// feas = 1: phase-one form (tau out in J).
int ehm_point_idx_batch(ehm_problem* P, int64_t n_inst, const double* theta, const int32_t* slot,
                        int32_t feas, double* J, double* u0, int32_t* status) {
    if (!P || !theta || !slot || !J || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    for (int64_t k = 0; k < n_inst; ++k)
        if (slot[k] < 0 || slot[k] >= P->dp.n_delta)
            return fail(EHM_E_INVALID, "instance %lld: slot %d out of range", (long long)k, slot[k]);
    return point_batch(P, n_inst, theta, slot, feas ? 1 : 0, J, u0, status, nullptr);
}

int ehm_min_simplex_batch(ehm_problem* P, int64_t n_inst, const double* R,
                          const uint8_t* delta, double* Jmin, int32_t* status) {
    if (!P || !R || !Jmin || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    std::vector<int32_t> didx;
    int rc = map_deltas(P, n_inst, delta, didx);
    if (rc) return rc;
    return simplex_batch(P, n_inst, R, nullptr, didx.data(), 0, Jmin, nullptr, status);
}

// ---- multi-commutation oracles (host-orchestrated over the batched kernels) ---------------
// Every mixed-integer problem of the reference is "enumerate the admissible commutations x
// one LP each" (SURVEY.md section 8a); these entry points expand the (instance,
// commutation[, vertex]) pairs on the host and run them as ONE batched launch per stage.
// Canonical choice rules: DESIGN.md "canonical commutation rule".

// feasibility of (theta_k, d_k) pairs through the phase-one LP
static int feas_pairs(ehm_problem* P, int64_t K, const double* theta, const int32_t* didx,
                      std::vector<uint8_t>& ok) {
    std::vector<double> tau((size_t)K);
    int rc = point_batch(P, K, theta, didx, 1, tau.data(), nullptr, nullptr, nullptr);
    if (rc) return rc;
    ok.resize((size_t)K);
    for (int64_t k = 0; k < K; ++k) ok[(size_t)k] = tau[(size_t)k] <= EHM_FEAS_TOL;
    return EHM_OK;
}

int ehm_solve_pt_batch(ehm_problem* P, int64_t n_inst, const double* theta, double* J,
                       double* u0, int32_t* delta_idx) {
    if (!P || !theta || !J || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    const int nd = P->dp.n_delta, p = P->dp.p, n_u = P->dp.n_u;
    const int64_t K = n_inst * nd;
    std::vector<double> th((size_t)K * p);
    std::vector<int32_t> di((size_t)K);
    for (int64_t k = 0; k < n_inst; ++k)
        for (int d = 0; d < nd; ++d) {
            std::memcpy(&th[(size_t)(k * nd + d) * p], theta + (size_t)k * p, p * sizeof(double));
            di[(size_t)(k * nd + d)] = d;
        }
    std::vector<uint8_t> ok;
    int rc = feas_pairs(P, K, th.data(), di.data(), ok);
    if (rc) return rc;
    // solve only the feasible pairs
    std::vector<int64_t> sel;
    for (int64_t q = 0; q < K; ++q)
        if (ok[(size_t)q]) sel.push_back(q);
    const int64_t F = (int64_t)sel.size();
    std::vector<double> th2((size_t)F * p), J2((size_t)F), u2((size_t)F * n_u);
    std::vector<int32_t> di2((size_t)F), st2((size_t)F);
    for (int64_t f = 0; f < F; ++f) {
        std::memcpy(&th2[(size_t)f * p], &th[(size_t)sel[(size_t)f] * p], p * sizeof(double));
        di2[(size_t)f] = di[(size_t)sel[(size_t)f]];
    }
    rc = point_batch(P, F, th2.data(), di2.data(), 0, J2.data(), u2.data(), st2.data(), nullptr);
    if (rc) return rc;
    for (int64_t k = 0; k < n_inst; ++k) {
        J[k] = INFINITY;
        if (delta_idx) delta_idx[k] = -1;
        if (u0)
            for (int c = 0; c < n_u; ++c) u0[k * n_u + c] = NAN;
    }
    std::vector<double> Jmin((size_t)n_inst, INFINITY);
    for (int64_t f = 0; f < F; ++f) {
        if (st2[(size_t)f] != 0) continue;
        const int64_t k = sel[(size_t)f] / nd;
        Jmin[(size_t)k] = std::min(Jmin[(size_t)k], J2[(size_t)f]);
    }
    std::vector<uint8_t> done((size_t)n_inst, 0);
    for (int64_t f = 0; f < F; ++f) {      // f ascends with the commutation index
        if (st2[(size_t)f] != 0) continue;
        const int64_t k = sel[(size_t)f] / nd;
        if (done[(size_t)k]) continue;
        const double jm = Jmin[(size_t)k];
        if (J2[(size_t)f] <= jm + EHM_TIE_TOL * (1.0 + std::fabs(jm))) {
            done[(size_t)k] = 1;
            J[k] = J2[(size_t)f];
            if (delta_idx) delta_idx[k] = di2[(size_t)f];
            if (u0) std::memcpy(u0 + k * n_u, &u2[(size_t)f * n_u], n_u * sizeof(double));
        }
    }
    return EHM_OK;
}

// which commutations are feasible at EVERY vertex of each simplex:  mask[k*nd + d]
static int vertex_feasible_mask(ehm_problem* P, int64_t n_inst, const double* R,
                                std::vector<uint8_t>& mask) {
    const int nd = P->dp.n_delta, p = P->dp.p, nv = p + 1;
    const int64_t K = n_inst * nd * nv;
    std::vector<double> th((size_t)K * p);
    std::vector<int32_t> di((size_t)K);
    for (int64_t k = 0; k < n_inst; ++k)
        for (int d = 0; d < nd; ++d)
            for (int v = 0; v < nv; ++v) {
                const int64_t q = (k * nd + d) * nv + v;
                std::memcpy(&th[(size_t)q * p], R + ((size_t)k * nv + v) * p, p * sizeof(double));
                di[(size_t)q] = d;
            }
    std::vector<uint8_t> ok;
    int rc = feas_pairs(P, K, th.data(), di.data(), ok);
    if (rc) return rc;
    mask.assign((size_t)(n_inst * nd), 1);
    for (int64_t q = 0; q < K; ++q)
        if (!ok[(size_t)q]) mask[(size_t)(q / nv)] = 0;
    return EHM_OK;
}

// P_theta_delta at every vertex of simplex k for commutation dsel[k] (>= 0)
static int vertex_solves(ehm_problem* P, int64_t n_inst, const double* R, const int32_t* dsel,
                         double* vJ, double* vu0, bool& all_ok) {
    const int p = P->dp.p, nv = p + 1, n_u = P->dp.n_u;
    std::vector<int64_t> sel;
    for (int64_t k = 0; k < n_inst; ++k)
        if (dsel[k] >= 0) sel.push_back(k);
    const int64_t F = (int64_t)sel.size() * nv;
    std::vector<double> th((size_t)F * p), J((size_t)F), u((size_t)F * n_u);
    std::vector<int32_t> di((size_t)F), st((size_t)F);
    for (size_t f = 0; f < sel.size(); ++f)
        for (int v = 0; v < nv; ++v) {
            std::memcpy(&th[(f * nv + v) * p], R + ((size_t)sel[f] * nv + v) * p,
                        p * sizeof(double));
            di[f * nv + v] = dsel[sel[f]];
        }
    int rc = point_batch(P, F, th.data(), di.data(), 0, J.data(), u.data(), st.data(), nullptr);
    if (rc) return rc;
    all_ok = true;
    for (size_t f = 0; f < sel.size(); ++f)
        for (int v = 0; v < nv; ++v) {
            const size_t q = f * nv + v;
            if (st[q] != 0) all_ok = false;
            if (vJ) vJ[(size_t)sel[f] * nv + v] = J[q];
            if (vu0)
                std::memcpy(vu0 + ((size_t)sel[f] * nv + v) * n_u, &u[q * n_u],
                            n_u * sizeof(double));
        }
    return EHM_OK;
}

int ehm_vr_batch(ehm_problem* P, int64_t n_inst, const double* R, int32_t* delta_idx,
                 double* vJ, double* vu0) {
    if (!P || !R || !delta_idx || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    const int nd = P->dp.n_delta;
    std::vector<uint8_t> mask;
    int rc = vertex_feasible_mask(P, n_inst, R, mask);
    if (rc) return rc;
    for (int64_t k = 0; k < n_inst; ++k) {
        delta_idx[k] = -1;
        for (int d = 0; d < nd; ++d)
            if (mask[(size_t)(k * nd + d)]) {
                delta_idx[k] = d;
                break;
            }
    }
    bool all_ok = true;
    rc = vertex_solves(P, n_inst, R, delta_idx, vJ, vu0, all_ok);
    if (rc) return rc;
    if (!all_ok) return fail(EHM_E_NUMERIC, "a vertex solve of V_R did not converge");
    return EHM_OK;
}

// slack t*(d) for every (instance, commutation) pair that is feasible somewhere in the
// simplex; -inf for the others.  tall[k*nd + d], optional alpha_all [k*nd+d][p+1].
static int slack_all(ehm_problem* P, int64_t n_inst, const double* R, const double* Vbar,
                     const std::vector<uint8_t>* restrict_mask, std::vector<double>& tall,
                     std::vector<double>* alpha_all, const uint8_t* known = nullptr,
                     std::vector<uint8_t>* feas_out = nullptr) {
    // known[k*nd + d] (optional): 0 = run the phase-one problem, 1 = the commutation is known to
    // be feasible somewhere on the simplex (e.g. at a vertex), 2 = known to be infeasible on it
    // (e.g. on the parent simplex).  feas_out: the verdicts (1 = feasible with an interior).
    const int nd = P->dp.n_delta, p = P->dp.p, nv = p + 1;
    const size_t nR = (size_t)nv * p;
    std::vector<int64_t> sel;
    for (int64_t q = 0; q < n_inst * nd; ++q)
        if ((!restrict_mask || (*restrict_mask)[(size_t)q]) && !(known && known[q] == 2))
            sel.push_back(q);
    int64_t F = (int64_t)sel.size();
    std::vector<double> R2((size_t)F * nR), V2((size_t)F * nv), obj((size_t)F);
    std::vector<int32_t> di((size_t)F), st((size_t)F);
    for (int64_t f = 0; f < F; ++f) {
        const int64_t k = sel[(size_t)f] / nd;
        std::memcpy(&R2[(size_t)f * nR], R + (size_t)k * nR, nR * sizeof(double));
        std::memcpy(&V2[(size_t)f * nv], Vbar + (size_t)k * nv, nv * sizeof(double));
        di[(size_t)f] = (int32_t)(sel[(size_t)f] % nd);
    }
    tall.assign((size_t)(n_inst * nd), -INFINITY);
    if (alpha_all) alpha_all->assign((size_t)(n_inst * nd) * nv, 0.0);
    if (feas_out) feas_out->assign((size_t)(n_inst * nd), 0);
    std::vector<double> tau_sel;      // phase-one optimum of every pair that stays selected
    if (nd > 1) {
        // drop pairs whose commutation is infeasible on the whole simplex: phase one for the
        // pairs nothing is known about
        std::vector<int64_t> ask;
        for (int64_t f = 0; f < F; ++f)
            if (!(known && known[sel[(size_t)f]] == 1)) ask.push_back(f);
        const int64_t A = (int64_t)ask.size();
        std::vector<double> Ra((size_t)A * nR), tau_a((size_t)A);
        std::vector<int32_t> da((size_t)A), sa((size_t)A);
        for (int64_t a = 0; a < A; ++a) {
            const int64_t f = ask[(size_t)a];
            std::memcpy(&Ra[(size_t)a * nR], &R2[(size_t)f * nR], nR * sizeof(double));
            da[(size_t)a] = di[(size_t)f];
        }
        if (A > 0) {
            int rc = simplex_batch(P, A, Ra.data(), nullptr, da.data(), SX_FEAS, tau_a.data(),
                                   nullptr, sa.data());
            if (rc) return rc;
        }
        // tau of a pair known to be feasible: far from the sliver band
        std::fill(obj.begin(), obj.end(), -1.0);
        for (int64_t a = 0; a < A; ++a) obj[(size_t)ask[(size_t)a]] = tau_a[(size_t)a];
        std::vector<int64_t> sel2;
        for (int64_t f = 0; f < F; ++f)
            if (obj[(size_t)f] <= EHM_FEAS_TOL) sel2.push_back(f);
        const int64_t F2 = (int64_t)sel2.size();
        std::vector<double> R3((size_t)F2 * nR), V3((size_t)F2 * nv);
        std::vector<int32_t> di3((size_t)F2);
        std::vector<int64_t> sel3((size_t)F2);
        for (int64_t g = 0; g < F2; ++g) {
            const int64_t f = sel2[(size_t)g];
            std::memcpy(&R3[(size_t)g * nR], &R2[(size_t)f * nR], nR * sizeof(double));
            std::memcpy(&V3[(size_t)g * nv], &V2[(size_t)f * nv], nv * sizeof(double));
            di3[(size_t)g] = di[(size_t)f];
            sel3[(size_t)g] = sel[(size_t)f];
            tau_sel.push_back(obj[(size_t)f]);
        }
        R2.swap(R3); V2.swap(V3); di.swap(di3); sel.swap(sel3);
        F = F2;
        obj.resize((size_t)F);
        st.resize((size_t)F);
    }
    std::vector<double> al((size_t)F * nv);
    int rc = simplex_batch(P, F, R2.data(), V2.data(), di.data(), SX_SLACK, obj.data(),
                           al.data(), st.data());
    if (rc) return rc;
    if (known && !tau_sel.empty()) {
        // a pair taken as feasible on the caller's word (feasible at a vertex WITHIN the phase-one
        // tolerance) whose slack problem stalls: get its phase-one optimum now, so that the
        // sliver rule below judges it exactly as it would have without the hint
        std::vector<int64_t> redo;
        for (int64_t f = 0; f < F; ++f)
            if (st[(size_t)f] != 0 && known[sel[(size_t)f]] == 1) redo.push_back(f);
        const int64_t A = (int64_t)redo.size();
        if (A > 0) {
            std::vector<double> Ra((size_t)A * nR), tau_a((size_t)A);
            std::vector<int32_t> da((size_t)A), sa((size_t)A);
            for (int64_t a = 0; a < A; ++a) {
                const int64_t f = redo[(size_t)a];
                std::memcpy(&Ra[(size_t)a * nR], &R2[(size_t)f * nR], nR * sizeof(double));
                da[(size_t)a] = di[(size_t)f];
            }
            rc = simplex_batch(P, A, Ra.data(), nullptr, da.data(), SX_FEAS, tau_a.data(), nullptr,
                               sa.data());
            if (rc) return rc;
            for (int64_t a = 0; a < A; ++a) tau_sel[(size_t)redo[(size_t)a]] = tau_a[(size_t)a];
        }
    }
    for (int64_t f = 0; f < F; ++f) {
        if (st[(size_t)f] != 0 && !tau_sel.empty() && tau_sel[(size_t)f] > -EHM_SLIVER_TOL) {
            // the commutation is feasible on the simplex only within the accuracy of the
            // phase-one optimum (|tau*| <= 1e-8 .. 1e-7 in row units): its feasible set has no
            // interior an interior-point method could work in.  It is treated as infeasible
            // there, which is also the verdict of a simplex-type solver for tau* > 0
            // (the CPU oracle: HiGHS at 1e-10); counted in ehm_counters.slivers
            P->slivers++;
            continue;
        }
        if (st[(size_t)f] != 0) {
            if (const char* path = getenv("EHM_DUMP_FAIL")) {     // debugging aid: the instance
                if (FILE* fp = fopen(path, "w")) {
                    fprintf(fp, "%d %d %.17g\n", (int)(sel[(size_t)f] % nd), p, obj[(size_t)f]);
                    for (size_t q = 0; q < nR; ++q) fprintf(fp, "%.17g ", R2[(size_t)f * nR + q]);
                    fprintf(fp, "\n");
                    for (int q = 0; q < nv; ++q) fprintf(fp, "%.17g ", V2[(size_t)f * nv + q]);
                    fprintf(fp, "\n%.17g %.17g\n", P->dp.eps_a, P->dp.eps_r);
                    fclose(fp);
                }
            }
            return fail(EHM_E_NUMERIC, "slack LP (instance %lld, commutation %d) did not converge",
                        (long long)(sel[(size_t)f] / nd), (int)(sel[(size_t)f] % nd));
        }
        tall[(size_t)sel[(size_t)f]] = obj[(size_t)f];
        if (feas_out) (*feas_out)[(size_t)sel[(size_t)f]] = 1;
        if (alpha_all)
            std::memcpy(&(*alpha_all)[(size_t)sel[(size_t)f] * nv], &al[(size_t)f * nv],
                        nv * sizeof(double));
    }
    return EHM_OK;
}

int ehm_bar_e_batch(ehm_problem* P, int64_t n_inst, const double* R, const double* Vbar,
                    uint8_t* closed, double* tbest) {
    if (!P || !R || !Vbar || !closed || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    const int nd = P->dp.n_delta;
    std::vector<double> tall;
    int rc = slack_all(P, n_inst, R, Vbar, nullptr, tall, nullptr);
    if (rc) return rc;
    for (int64_t k = 0; k < n_inst; ++k) {
        double tb = -INFINITY;
        for (int d = 0; d < nd; ++d) tb = std::max(tb, tall[(size_t)(k * nd + d)]);
        closed[k] = (tb >= 0.0) ? 0 : 1;
        if (tbest) tbest[k] = tb;
    }
    return EHM_OK;
}

// Second half of bar_D_delta_R, given the slacks of the commutations feasible at every vertex
// (mask) -- shared by ehm_bar_d_batch and ehm_lcss_batch.
static int bar_d_finish(ehm_problem* P, int64_t n_inst, const double* R, const double* Vbar,
                        const std::vector<int32_t>& dref, const std::vector<uint8_t>& mask,
                        const std::vector<double>& tall, const std::vector<double>& alpha_all,
                        int32_t* delta_idx, double* theta_star, double* vJ, double* vu0,
                        uint8_t* var_small) {
    const int nd = P->dp.n_delta, p = P->dp.p, nv = p + 1;
    std::vector<double> ths((size_t)n_inst * p, 0.0);
    for (int64_t k = 0; k < n_inst; ++k) {
        int best = -1;
        double tb = -INFINITY;
        for (int d = 0; d < nd; ++d) {
            const double t = tall[(size_t)(k * nd + d)];
            if (mask[(size_t)(k * nd + d)] && t >= 0.0) tb = std::max(tb, t);
        }
        for (int d = 0; d < nd && best < 0; ++d) {
            const double t = tall[(size_t)(k * nd + d)];
            if (mask[(size_t)(k * nd + d)] && t >= 0.0 &&
                t >= tb - EHM_TIE_TOL * (1.0 + std::fabs(tb)))
                best = d;
        }
        if (best >= 0 && best == dref[(size_t)k]) best = -1;   // lib/oracle.py:384-394
        delta_idx[k] = best;
        if (best >= 0) {
            const double* al = &alpha_all[(size_t)(k * nd + best) * nv];
            for (int c = 0; c < p; ++c) {
                double acc = 0.0;
                for (int v = 0; v < nv; ++v) acc += al[v] * R[((size_t)k * nv + v) * p + c];
                ths[(size_t)k * p + c] = acc;
            }
        }
        if (theta_star)
            std::memcpy(theta_star + (size_t)k * p, &ths[(size_t)k * p], p * sizeof(double));
        if (var_small) var_small[k] = 0;
    }
    bool all_ok = true;
    int rc = vertex_solves(P, n_inst, R, delta_idx, vJ, vu0, all_ok);
    if (rc) return rc;
    if (!all_ok) return fail(EHM_E_NUMERIC, "a vertex solve of bar_D did not converge");
    // in_variability_ball (lib/oracle.py:220-283) for the instances with a better commutation
    std::vector<int64_t> sel;
    for (int64_t k = 0; k < n_inst; ++k)
        if (delta_idx[k] >= 0) sel.push_back(k);
    const int64_t F = (int64_t)sel.size();
    if (F > 0 && var_small) {
        const size_t nR = (size_t)nv * p;
        std::vector<double> R2((size_t)F * nR), Jmin((size_t)F), th2((size_t)F * p), Jth((size_t)F);
        std::vector<int32_t> d_ref2((size_t)F), d_star2((size_t)F), st((size_t)F), st2((size_t)F);
        for (int64_t f = 0; f < F; ++f) {
            const int64_t k = sel[(size_t)f];
            std::memcpy(&R2[(size_t)f * nR], R + (size_t)k * nR, nR * sizeof(double));
            std::memcpy(&th2[(size_t)f * p], &ths[(size_t)k * p], p * sizeof(double));
            d_ref2[(size_t)f] = dref[(size_t)k];
            d_star2[(size_t)f] = delta_idx[k];
        }
        rc = simplex_batch(P, F, R2.data(), nullptr, d_ref2.data(), SX_MIN, Jmin.data(), nullptr,
                           st.data());
        if (rc) return rc;
        rc = point_batch(P, F, th2.data(), d_star2.data(), 0, Jth.data(), nullptr, st2.data(),
                         nullptr);
        if (rc) return rc;
        for (int64_t f = 0; f < F; ++f) {
            if (st[(size_t)f] != 0 || st2[(size_t)f] != 0)
                return fail(EHM_E_NUMERIC, "in_variability_ball solve did not converge");
            const int64_t k = sel[(size_t)f];
            double vmax = -INFINITY;
            for (int v = 0; v < nv; ++v) vmax = std::max(vmax, Vbar[(size_t)k * nv + v]);
            const double rhs = std::max(P->dp.eps_a, P->dp.eps_r * Jth[(size_t)f]);
            var_small[k] = (vmax - Jmin[(size_t)f] < rhs) ? 1 : 0;
        }
    }
    return EHM_OK;
}

int ehm_bar_d_batch(ehm_problem* P, int64_t n_inst, const double* R, const double* Vbar,
                    const uint8_t* delta_ref, int32_t* delta_idx, double* theta_star,
                    double* vJ, double* vu0, uint8_t* var_small) {
    if (!P || !R || !Vbar || !delta_idx || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    std::vector<int32_t> dref;
    int rc = map_deltas(P, n_inst, delta_ref, dref);
    if (rc) return rc;
    std::vector<uint8_t> mask;
    rc = vertex_feasible_mask(P, n_inst, R, mask);
    if (rc) return rc;
    std::vector<double> tall, alpha_all;
    rc = slack_all(P, n_inst, R, Vbar, &mask, tall, &alpha_all);
    if (rc) return rc;
    return bar_d_finish(P, n_inst, R, Vbar, dref, mask, tall, alpha_all, delta_idx, theta_star,
                        vJ, vu0, var_small);
}

// P_theta_delta(theta, d, check_feasibility=True) for EVERY commutation d: feasible[k*nd + d].
int ehm_feas_all_batch(ehm_problem* P, int64_t n_inst, const double* theta, uint8_t* feasible) {
    if (!P || !theta || !feasible || n_inst < 0) return fail(EHM_E_INVALID, "bad argument");
    const int nd = P->dp.n_delta, p = P->dp.p;
    const int64_t K = n_inst * nd;
    std::vector<double> th((size_t)K * p);
    std::vector<int32_t> di((size_t)K);
    for (int64_t k = 0; k < n_inst; ++k)
        for (int d = 0; d < nd; ++d) {
            std::memcpy(&th[(size_t)(k * nd + d) * p], theta + (size_t)k * p, p * sizeof(double));
            di[(size_t)(k * nd + d)] = d;
        }
    std::vector<uint8_t> ok;
    int rc = feas_pairs(P, K, th.data(), di.data(), ok);
    if (rc) return rc;
    for (int64_t q = 0; q < K; ++q) feasible[q] = ok[(size_t)q];
    return EHM_OK;
}

// One visit of Worker.lcss for a batch of nodes (lib/worker.py:367-401): bar_E_delta_R, and for
// the nodes that stay open bar_D_delta_R, sharing what the two oracles have in common and what
// the caller already knows about the node:
//   vfeas  [n][p+1][nd]  P_theta_delta feasibility of every commutation at every vertex
//                        (children inherit p of their p+1 vertices: ehm_feas_all_batch on the
//                        new midpoints only);
//   cand   [n][nd] or NULL  commutations that were feasible somewhere on the PARENT simplex.
// A commutation feasible at a vertex is feasible on the simplex, one infeasible on the parent
// is infeasible on the child; only the rest needs a phase-one problem, and the slacks bar_E
// computes are the ones bar_D ranks.  Same verdicts and optima as ehm_bar_e_batch followed by
// ehm_bar_d_batch, about 2.5 times fewer sub-problems.
int ehm_lcss_batch(ehm_problem* P, int64_t n_inst, const double* R, const double* Vbar,
                   const uint8_t* delta_ref, const uint8_t* vfeas, const uint8_t* cand,
                   uint8_t* closed, double* tbest, uint8_t* cand_out, int32_t* delta_idx,
                   double* theta_star, double* vJ, double* vu0, uint8_t* var_small) {
    if (!P || !R || !Vbar || !delta_ref || !vfeas || !closed || !delta_idx || n_inst < 0)
        return fail(EHM_E_INVALID, "bad argument");
    const int nd = P->dp.n_delta, p = P->dp.p, nv = p + 1;
    std::vector<int32_t> dref;
    int rc = map_deltas(P, n_inst, delta_ref, dref);
    if (rc) return rc;
    std::vector<uint8_t> known((size_t)(n_inst * nd), 0), vall((size_t)(n_inst * nd), 1);
    for (int64_t k = 0; k < n_inst; ++k)
        for (int d = 0; d < nd; ++d) {
            bool any = false, all = true;
            for (int v = 0; v < nv; ++v) {
                const bool f = vfeas[((size_t)k * nv + v) * nd + d] != 0;
                any = any || f;
                all = all && f;
            }
            vall[(size_t)(k * nd + d)] = all ? 1 : 0;
            known[(size_t)(k * nd + d)] =
                any ? 1 : ((cand && !cand[(size_t)(k * nd + d)]) ? 2 : 0);
        }
    std::vector<double> tall, alpha_all;
    std::vector<uint8_t> feas;
    rc = slack_all(P, n_inst, R, Vbar, nullptr, tall, &alpha_all, known.data(), &feas);
    if (rc) return rc;
    std::vector<int64_t> open;
    for (int64_t k = 0; k < n_inst; ++k) {
        double tb = -INFINITY;
        for (int d = 0; d < nd; ++d) tb = std::max(tb, tall[(size_t)(k * nd + d)]);
        closed[k] = (tb >= 0.0) ? 0 : 1;
        if (tbest) tbest[k] = tb;
        if (cand_out)
            std::memcpy(cand_out + (size_t)k * nd, &feas[(size_t)k * nd], (size_t)nd);
        delta_idx[k] = -1;
        if (var_small) var_small[k] = 0;
        if (!closed[k]) open.push_back(k);
    }
    const int64_t M = (int64_t)open.size();
    if (M == 0) return EHM_OK;
    // bar_D for the open nodes, on compacted copies
    const size_t nR = (size_t)nv * p;
    const int n_u = P->dp.n_u;
    std::vector<double> R2((size_t)M * nR), V2((size_t)M * nv), t2((size_t)(M * nd)),
        a2((size_t)(M * nd) * nv), ths((size_t)M * p), vJ2((size_t)M * nv),
        vu2((size_t)M * nv * n_u);
    std::vector<uint8_t> m2((size_t)(M * nd)), vs2((size_t)M);
    std::vector<int32_t> dr2((size_t)M), di2((size_t)M);
    for (int64_t q = 0; q < M; ++q) {
        const int64_t k = open[(size_t)q];
        std::memcpy(&R2[(size_t)q * nR], R + (size_t)k * nR, nR * sizeof(double));
        std::memcpy(&V2[(size_t)q * nv], Vbar + (size_t)k * nv, nv * sizeof(double));
        std::memcpy(&t2[(size_t)q * nd], &tall[(size_t)k * nd], nd * sizeof(double));
        std::memcpy(&a2[(size_t)q * nd * nv], &alpha_all[(size_t)k * nd * nv],
                    (size_t)nd * nv * sizeof(double));
        // commutations feasible at every vertex AND with an interior on the simplex
        for (int d = 0; d < nd; ++d)
            m2[(size_t)(q * nd + d)] = vall[(size_t)(k * nd + d)] && feas[(size_t)(k * nd + d)];
        dr2[(size_t)q] = dref[(size_t)k];
    }
    rc = bar_d_finish(P, M, R2.data(), V2.data(), dr2, m2, t2, a2, di2.data(), ths.data(),
                      vJ2.data(), vu2.data(), vs2.data());
    if (rc) return rc;
    for (int64_t q = 0; q < M; ++q) {
        const int64_t k = open[(size_t)q];
        delta_idx[k] = di2[(size_t)q];
        if (theta_star)
            std::memcpy(theta_star + (size_t)k * p, &ths[(size_t)q * p], p * sizeof(double));
        if (vJ) std::memcpy(vJ + (size_t)k * nv, &vJ2[(size_t)q * nv], nv * sizeof(double));
        if (vu0)
            std::memcpy(vu0 + (size_t)k * nv * n_u, &vu2[(size_t)q * nv * n_u],
                        (size_t)nv * n_u * sizeof(double));
        if (var_small) var_small[k] = vs2[(size_t)q];
    }
    return EHM_OK;
}

// ---- geometry -------------------------------------------------------------------------------
int ehm_split_batch(int device, int64_t n, int32_t p, const double* R, double* S1, double* S2,
                    int32_t* ij) {
    if (!R || !S1 || !S2 || !ij || n < 0 || p < 1 || p > EHM_MAX_P)
        return fail(EHM_E_INVALID, "bad argument");
    if (n == 0) return EHM_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev || device < 0)
        return fail(EHM_E_NO_DEVICE, "no HIP device %d (libehmpc has no CPU fallback)", device);
    HIP_TRY(hipSetDevice(device), EHM_E_NO_DEVICE);
    const size_t bytes = (size_t)n * (p + 1) * p * sizeof(double);
    double *dR = nullptr, *d1 = nullptr, *d2 = nullptr;
    int32_t* dij = nullptr;
    HIP_TRY(hipMalloc((void**)&dR, bytes), EHM_E_HIP);
    HIP_TRY(hipMalloc((void**)&d1, bytes), EHM_E_HIP);
    HIP_TRY(hipMalloc((void**)&d2, bytes), EHM_E_HIP);
    HIP_TRY(hipMalloc((void**)&dij, (size_t)n * 2 * sizeof(int32_t)), EHM_E_HIP);
    HIP_TRY(hipMemcpy(dR, R, bytes, hipMemcpyHostToDevice), EHM_E_HIP);
    hipLaunchKernelGGL(k_split_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0,
                       (long long)n, (int)p, dR, d1, d2, dij);
    HIP_TRY(hipGetLastError(), EHM_E_HIP);
    HIP_TRY(hipMemcpy(S1, d1, bytes, hipMemcpyDeviceToHost), EHM_E_HIP);
    HIP_TRY(hipMemcpy(S2, d2, bytes, hipMemcpyDeviceToHost), EHM_E_HIP);
    HIP_TRY(hipMemcpy(ij, dij, (size_t)n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost),
            EHM_E_HIP);
    (void)hipFree(dR); (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(dij);
    return EHM_OK;
}

int ehm_volume_batch(int device, int64_t n, int32_t p, const double* R, double* vol) {
    if (!R || !vol || n < 0 || p < 1 || p > EHM_MAX_P) return fail(EHM_E_INVALID, "bad argument");
    if (n == 0) return EHM_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev || device < 0)
        return fail(EHM_E_NO_DEVICE, "no HIP device %d (libehmpc has no CPU fallback)", device);
    HIP_TRY(hipSetDevice(device), EHM_E_NO_DEVICE);
    const size_t bytes = (size_t)n * (p + 1) * p * sizeof(double);
    double *dR = nullptr, *dv = nullptr;
    HIP_TRY(hipMalloc((void**)&dR, bytes), EHM_E_HIP);
    HIP_TRY(hipMalloc((void**)&dv, (size_t)n * sizeof(double)), EHM_E_HIP);
    HIP_TRY(hipMemcpy(dR, R, bytes, hipMemcpyHostToDevice), EHM_E_HIP);
    hipLaunchKernelGGL(k_volume_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0,
                       (long long)n, (int)p, dR, dv);
    HIP_TRY(hipGetLastError(), EHM_E_HIP);
    HIP_TRY(hipMemcpy(vol, dv, (size_t)n * sizeof(double), hipMemcpyDeviceToHost), EHM_E_HIP);
    (void)hipFree(dR); (void)hipFree(dv);
    return EHM_OK;
}

// ---- partition engine -------------------------------------------------------------------------
static int read_counters(ehm_problem* P, DevCounters& c);
#include "ehm_hybrid.h"

int ehm_tree_destroy(ehm_tree* T) {
    if (!T) return EHM_OK;
    (void)hipSetDevice(T->device);
    if (T->prob && T->prob->active_run == T) T->prob->active_run = nullptr;
    if (T->prob) {
        auto& v = T->prob->trees;
        v.erase(std::remove(v.begin(), v.end(), T), v.end());
    }
    if (T->hy) {
        T->hy->release();
        delete T->hy;
        T->hy = nullptr;
    }
    for (hipEvent_t e : T->run.evs) (void)hipEventDestroy(e);
    T->run.evs.clear();
    if (T->run.ev0) (void)hipEventDestroy(T->run.ev0);
    T->run.ev0 = nullptr;
    if (T->prob && T->cap > T->prob->pool_cache.cap) {
        // keep the larger pool for the next run of this problem
        auto& c = T->prob->pool_cache;
        std::swap(c.rec, T->rec); std::swap(c.left, T->left); std::swap(c.didx, T->didx);
        std::swap(c.depth, T->depth); std::swap(c.flags, T->flags); std::swap(c.tstar, T->tstar);
        std::swap(c.grad, T->grad); std::swap(c.wit, T->wit);
        std::swap(c.mt_state, T->mt_state); std::swap(c.mt_data, T->mt_data);
        c.cap = T->cap;
    }
    T->rec.release(); T->left.release(); T->didx.release(); T->depth.release();
    T->flags.release(); T->tstar.release(); T->grad.release(); T->code.release();
    T->wit.release(); T->mt_state.release(); T->mt_data.release();
    T->d_perm.release(); T->d_inv.release();
    delete T;
    return EHM_OK;
}

static int tree_alloc(ehm_tree* T, ehm_problem* P, long long cap) {
    const int p = P->dp.p, n_u = P->dp.n_u;
    const int stride = ((rec_doubles(p, n_u) + 7) / 8) * 8;
    int rc;
    if (P->pool_cache.cap >= cap) {
        auto& c = P->pool_cache;
        std::swap(c.rec, T->rec); std::swap(c.left, T->left); std::swap(c.didx, T->didx);
        std::swap(c.depth, T->depth); std::swap(c.flags, T->flags); std::swap(c.tstar, T->tstar);
        std::swap(c.grad, T->grad); std::swap(c.wit, T->wit);
        std::swap(c.mt_state, T->mt_state); std::swap(c.mt_data, T->mt_data);
        cap = c.cap;
        c.cap = 0;
    }
    if ((rc = T->rec.ensure((size_t)cap * stride * sizeof(double)))) return rc;
    if ((rc = T->left.ensure((size_t)cap * sizeof(int32_t)))) return rc;
    if ((rc = T->didx.ensure((size_t)cap * sizeof(int32_t)))) return rc;
    if ((rc = T->depth.ensure((size_t)cap * sizeof(int32_t)))) return rc;
    if ((rc = T->flags.ensure((size_t)cap))) return rc;
    if ((rc = T->tstar.ensure((size_t)cap * sizeof(double)))) return rc;
    // vertex gradients of the optimal cost (cutting-plane closure of leaves, ehm_dev.h): kept by
    // the shared-block kernels of a single-commutation handle (linear or quadratic cost)
    const bool grads = P->solver_gen == 2 && P->dp.n_delta == 1 && !getenv("EHM_NO_CUTS");
    if (grads && (rc = T->grad.ensure((size_t)cap * (p + 1) * p * sizeof(double)))) return rc;
    T->dt.grad = grads ? T->grad.as<double>() : nullptr;
    // witnesses handed from an open node to its child (DevTree::wit): linear costs, the
    // midpoint-first persistent kernel; every kernel that creates nodes clears or writes them
    const bool wits = grads && !P->quadratic && P->mid_first && P->inherit_wit &&
                      !getenv("EHM_NO_WITNESS");
    if (wits && (rc = T->wit.ensure((size_t)cap * (p + 2) * sizeof(double)))) return rc;
    T->dt.wit = wits ? T->wit.as<double>() : nullptr;
    // table of midpoint optima (ehm_midtable.h): single commutation, linear cost, the
    // midpoint-first persistent kernel.  One slot per two node records (a power of two): a
    // partition has at most one distinct midpoint per split = per two nodes, and 7-8 splits
    // share one in practice, so the table stays below 1/8 full
    const bool mids = grads && !P->quadratic && P->mid_first && P->share_mid &&
                      !getenv("EHM_NO_MIDTABLE");
    T->dt.mt = MidTable{nullptr, nullptr, 0u};
    if (mids) {
        unsigned long long slots = 1024;
        while (slots < (unsigned long long)cap / 2 && slots < (1ull << 30)) slots <<= 1;
        if ((rc = T->mt_state.ensure((size_t)slots * sizeof(unsigned long long)))) return rc;
        if ((rc = T->mt_data.ensure((size_t)slots * MT_DOUBLES * sizeof(double)))) return rc;
        T->dt.mt = MidTable{T->mt_state.as<unsigned long long>(), T->mt_data.as<double>(),
                            (unsigned int)(slots - 1)};
    }
    T->prob = P;
    T->device = P->device;
    if (std::find(P->trees.begin(), P->trees.end(), T) == P->trees.end()) P->trees.push_back(T);
    T->cap = cap;
    T->dt.rec = T->rec.as<double>();
    T->dt.left = T->left.as<int32_t>();
    T->dt.didx = T->didx.as<int32_t>();
    T->dt.depth = T->depth.as<int32_t>();
    T->dt.flags = T->flags.as<uint8_t>();
    T->dt.tstar = T->tstar.as<double>();
    T->dt.rec_stride = stride;
    T->dt.p = p;
    T->dt.n_u = n_u;
    return EHM_OK;
}

static int read_counters(ehm_problem* P, DevCounters& c) {
    HIP_TRY(hipMemcpyAsync(&c, P->d_cnt, sizeof c, hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    return EHM_OK;
}

int ehm_problem_layout(ehm_problem* P, int32_t out[4]) {
    if (!P || !out) return fail(EHM_E_INVALID, "null argument");
    out[0] = P->dp.nd0;
    out[1] = P->dp.n - P->dp.nd0;
    out[2] = P->dp.LE4;
    out[3] = P->dp.lda4;
    return EHM_OK;
}

int ehm_solver_phase_ticks(ehm_problem* P, int64_t out[24]) {
    if (!P || !out) return fail(EHM_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    DevCounters c;
    int rc = read_counters(P, c);
    if (rc) return rc;
    for (int k = 0; k < 24; ++k) out[k] = (int64_t)c.phase[k];
    return EHM_OK;
}

// ---- resumable engine: begin / step / take / give / finish ---------------------------------
// ehm_partition_run = begin + step(until the frontier is empty) + finish.  The pieces exist for
// the multi-GPU driver (explicit_hybrid_mpc_amd/distributed.py): every rank runs a few sweeps,
// the ranks all-gather their frontier sizes and move node records from the longest frontiers
// to the shortest.

// flat export (ehm_tree_export): node k of the output is device node perm[k] (or k); its record
// is split into the caller's three arrays, its child index renumbered through inv
__global__ void k_export_gather(DevTree T, const int32_t* __restrict__ perm,
                                const int32_t* __restrict__ inv, long long k0, long long nk,
                                int n_u, double* __restrict__ out_v, double* __restrict__ out_c,
                                double* __restrict__ out_u, double* __restrict__ out_t,
                                int32_t* __restrict__ out_l, int32_t* __restrict__ out_r,
                                int32_t* __restrict__ out_d, uint8_t* __restrict__ out_f) {
    const int p = T.p, nR = (p + 1) * p, nc = p + 1, nu = (p + 1) * n_u, nrec = nR + nc + nu;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nk * nrec) return;
    const long long k = t / nrec;
    const int e = (int)(t - k * nrec);
    const long long id = perm ? perm[k0 + k] : k0 + k;
    const double v = T.rec[(size_t)id * T.rec_stride + e];
    if (e < nR) out_v[k * nR + e] = v;
    else if (e < nR + nc) out_c[k * nc + (e - nR)] = v;
    else out_u[k * nu + (e - nR - nc)] = v;
    if (e == 0) {
        out_t[k] = T.tstar[id];
        int32_t c = T.left[id];
        if (c >= 0 && inv) c = inv[c];
        out_l[k] = c;
        out_r[k] = (c < 0) ? -1 : c + 1;
        out_d[k] = T.didx[id];
        out_f[k] = T.flags[id];
    }
}

// ---- breadth-first numbering on the device (ehm_tree_export of the persistent engine) -----------
// The persistent kernel allocates node ids in the order its wavefronts split; the export
// renumbers them breadth first (roots in order, then per level the two children of every split
// node in parent order: the numbering of the level-synchronous engine).  Level by level:
// perm[lo..hi) holds level d in its final order; an exclusive scan of "was split" over it gives
// every parent the place of its children in level d + 1.  Three small launches per level with
// the bounds in device memory (no host round trip): count per block, scan of the block sums,
// write.  bounds[2 * parity] = lo, [2 * parity + 1] = hi.
#define BFS_BLOCKS 256
#define BFS_THREADS 256
__global__ void k_bfs_roots(int32_t* __restrict__ perm, int32_t* __restrict__ bounds, int n_roots) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_roots) perm[t] = t;
    if (t == 0) { bounds[0] = 0; bounds[1] = n_roots; bounds[2] = n_roots; bounds[3] = n_roots; }
}
__global__ void k_bfs_count(const int32_t* __restrict__ left, const int32_t* __restrict__ perm,
                            const int32_t* __restrict__ bounds, int parity,
                            int32_t* __restrict__ bsum) {
    const int lo = bounds[2 * parity], hi = bounds[2 * parity + 1];
    const int per = (hi - lo + (int)gridDim.x - 1) / (int)gridDim.x;
    const int b0 = lo + (int)blockIdx.x * per, b1 = min(hi, b0 + per);
    int c = 0;
    for (int i = b0 + (int)threadIdx.x; i < b1; i += (int)blockDim.x) c += left[perm[i]] >= 0;
    __shared__ int sh[BFS_THREADS / 64];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < BFS_THREADS / 64; ++w) tot += sh[w];
        bsum[blockIdx.x] = tot;
    }
}
__global__ void k_bfs_scan(int32_t* __restrict__ bsum, int32_t* __restrict__ bounds, int parity) {
    __shared__ int sh[BFS_BLOCKS];
    const int t = threadIdx.x;
    sh[t] = bsum[t];
    __syncthreads();
    for (int o = 1; o < BFS_BLOCKS; o <<= 1) {          // inclusive scan, BFS_BLOCKS threads
        const int v = (t >= o) ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    bsum[t] = t ? sh[t - 1] : 0;
    if (t == 0) {
        const int hi = bounds[2 * parity + 1];
        bounds[2 * (parity ^ 1)] = hi;
        bounds[2 * (parity ^ 1) + 1] = hi + 2 * sh[BFS_BLOCKS - 1];
    }
}
__global__ void k_bfs_write(const int32_t* __restrict__ left, int32_t* __restrict__ perm,
                            const int32_t* __restrict__ bounds, int parity,
                            const int32_t* __restrict__ boff) {
    const int lo = bounds[2 * parity], hi = bounds[2 * parity + 1];
    const int per = (hi - lo + (int)gridDim.x - 1) / (int)gridDim.x;
    const int b0 = lo + (int)blockIdx.x * per, b1 = min(hi, b0 + per);
    __shared__ int woff[BFS_THREADS / 64];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = boff[blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = b0; base < b1; base += (int)blockDim.x) {
        const int i = base + (int)threadIdx.x;
        const int c = (i < b1) ? left[perm[i]] : -1;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(c >= 0);
        const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) woff[wave] = __builtin_popcountll(m);
        __syncthreads();
        int off = s_run;
        for (int w = 0; w < wave; ++w) off += woff[w];
        if (c >= 0) {
            perm[hi + 2 * (off + before)] = c;
            perm[hi + 2 * (off + before) + 1] = c + 1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < BFS_THREADS / 64; ++w) tot += woff[w];
            s_run += tot;
        }
        __syncthreads();
    }
}
__global__ void k_bfs_invert(const int32_t* __restrict__ perm, int32_t* __restrict__ inv, long long n) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) inv[perm[k]] = (int32_t)k;
}

// gather / scatter of node records for the frontier hand-over
__global__ void k_take_nodes(DevTree T, const int32_t* __restrict__ ids, int n, int nrec,
                             double* __restrict__ rec_out, int32_t* __restrict__ meta_out) {
    const int k = blockIdx.x;
    if (k >= n) return;
    const int id = ids[k];
    const double* r = T.rec + (size_t)id * T.rec_stride;
    for (int q = threadIdx.x; q < nrec; q += blockDim.x) rec_out[(size_t)k * nrec + q] = r[q];
    if (threadIdx.x == 0) {
        meta_out[2 * k] = T.didx[id];
        meta_out[2 * k + 1] = T.depth[id];
        T.flags[id] |= 4;                      // subtree now owned by another rank
    }
}
__global__ void k_give_nodes(DevTree T, int first, int n, int nrec,
                             const double* __restrict__ rec_in,
                             const int32_t* __restrict__ meta_in, int32_t* __restrict__ frontier,
                             int frontier_at) {
    const int k = blockIdx.x;
    if (k >= n) return;
    const int id = first + k;
    double* r = T.rec + (size_t)id * T.rec_stride;
    for (int q = threadIdx.x; q < nrec; q += blockDim.x) r[q] = rec_in[(size_t)k * nrec + q];
    if (T.grad) {       // the vertex gradients do not travel: unknown (no cutting-plane closure)
        const int ng = (T.p + 1) * T.p;
        for (int q = threadIdx.x; q < ng; q += blockDim.x)
            T.grad[(size_t)id * ng + q] = __builtin_nan("");
    }
    if (T.wit)          // nor do witnesses
        for (int q = threadIdx.x; q < T.p + 2; q += blockDim.x)
            T.wit[(size_t)id * (T.p + 2) + q] = 0.0;
    if (threadIdx.x == 0) {
        T.left[id] = -1;
        T.didx[id] = meta_in[2 * k];
        T.depth[id] = meta_in[2 * k + 1];
        T.flags[id] = 2 | 32;                  // has data, received from another rank
        T.tstar[id] = 0.0;
        frontier[frontier_at + k] = id;
    }
}

static void run_release_events(ehm_tree* T) {
    for (hipEvent_t e : T->run.evs) (void)hipEventDestroy(e);
    T->run.evs.clear();
    T->run.ev_kind.clear();
    if (T->run.ev0) (void)hipEventDestroy(T->run.ev0);
    T->run.ev0 = nullptr;
}

int ehm_partition_begin(ehm_problem* P, int64_t n_roots, const double* root_vertices,
                        const ehm_node_init* init, const ehm_run_opts* opts, ehm_tree** out) {
    if (!P || !root_vertices || !out || n_roots < 1) return fail(EHM_E_INVALID, "bad argument");
    *out = nullptr;
    // frontier buffers, queue and counters of a run live in the problem handle: one run at a time
    if (P->active_run)
        return fail(EHM_E_INVALID, "another partition run is active on this problem handle "
                                   "(finish or destroy it first)");
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    const int p = P->dp.p, n_u = P->dp.n_u;
    long long cap = (opts && opts->max_nodes > 0) ? opts->max_nodes : (1LL << 21);
    cap = std::max<long long>(cap, 2 * n_roots);
    ehm_tree* T = new ehm_tree();
    auto& R = T->run;
    R.active = true;
    R.max_depth = (opts && opts->max_depth > 0) ? opts->max_depth : 0;
    R.action = opts ? opts->action : 0;
    R.engine = opts ? opts->engine : 0;
    if (const char* e = getenv("EHM_ENGINE")) R.engine = atoi(e);
    R.shard_world = (opts && opts->shard_world > 1) ? opts->shard_world : 1;
    R.shard_rank = opts ? opts->shard_rank : 0;
    R.shard_min = opts ? opts->shard_min_frontier : 0;
    R.deal_depth = (opts && opts->shard_world > 1 && opts->deal_depth > 0) ? opts->deal_depth : 0;
    if (R.shard_world > 1 && (R.shard_rank < 0 || R.shard_rank >= R.shard_world)) {
        delete T;
        return fail(EHM_E_INVALID, "shard_rank %d out of range for world %d", R.shard_rank,
                    R.shard_world);
    }
    R.sharded = (R.shard_world == 1);
    T->skip_volume = opts ? opts->skip_volume : 0;
    int rc = tree_alloc(T, P, cap);
    if (rc) { ehm_tree_destroy(T); return rc; }
    T->limit = cap;
#define RUN_TRY(expr)                          \
    do {                                       \
        int rc_ = (expr);                      \
        if (rc_) {                             \
            ehm_tree_destroy(T);               \
            return rc_;                        \
        }                                      \
    } while (0)
    const int stride = T->dt.rec_stride;
    const int nR = (p + 1) * p;
    {
        std::vector<double> recs((size_t)n_roots * stride, 0.0);
        std::vector<int32_t> left((size_t)n_roots, -1), didx((size_t)n_roots, 0),
            depth((size_t)n_roots, 0);
        std::vector<uint8_t> flags((size_t)n_roots, 2);
        std::vector<double> ts((size_t)n_roots, 0.0);
        for (int64_t k = 0; k < n_roots; ++k) {
            double* r = recs.data() + (size_t)k * stride;
            std::memcpy(r, root_vertices + (size_t)k * nR, nR * sizeof(double));
            if (R.action == 1 && init && init->vcost && init->vinput) {
                std::memcpy(r + rec_off_vcost(p), init->vcost + (size_t)k * (p + 1),
                            (p + 1) * sizeof(double));
                std::memcpy(r + rec_off_vinput(p), init->vinput + (size_t)k * (p + 1) * n_u,
                            (size_t)(p + 1) * n_u * sizeof(double));
            }
        }
        HIP_TRY(hipMemcpyAsync(T->dt.rec, recs.data(), recs.size() * sizeof(double),
                               hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(T->dt.left, left.data(), left.size() * 4, hipMemcpyHostToDevice,
                               P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(T->dt.didx, didx.data(), didx.size() * 4, hipMemcpyHostToDevice,
                               P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(T->dt.depth, depth.data(), depth.size() * 4,
                               hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(T->dt.flags, flags.data(), flags.size(), hipMemcpyHostToDevice,
                               P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpyAsync(T->dt.tstar, ts.data(), ts.size() * 8, hipMemcpyHostToDevice,
                               P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    }
    // frontier buffers (ping-pong) + open flags + open list + count live in the problem handle
    long long fr_cap = std::max<long long>(n_roots, 1024);
    RUN_TRY(P->fr_a.ensure((size_t)fr_cap * 4));
    RUN_TRY(P->fr_b.ensure((size_t)fr_cap * 4));
    RUN_TRY(P->open_flag.ensure((size_t)fr_cap * 4));
    RUN_TRY(P->open_list.ensure((size_t)fr_cap * 4));
    RUN_TRY(P->d_count.ensure(64));
    {
        std::vector<int32_t> ids((size_t)n_roots);
        for (int64_t k = 0; k < n_roots; ++k) ids[(size_t)k] = (int32_t)k;
        hipError_t e = hipMemcpyAsync(P->fr_a.ptr, ids.data(), ids.size() * 4,
                                      hipMemcpyHostToDevice, P->stream);
        if (e != hipSuccess) RUN_TRY(fail(EHM_E_HIP, "frontier upload failed"));
        (void)hipStreamSynchronize(P->stream);
    }
    RUN_TRY(read_counters(P, R.c0));
    if (T->dt.grad) {
        // root gradients start unknown (all-ones = NaN); the vertex solves of 'ecc' fill them
        hipError_t e = hipMemsetAsync(T->dt.grad, 0xFF, (size_t)n_roots * (p + 1) * p * 8,
                                      P->stream);
        if (e != hipSuccess) RUN_TRY(fail(EHM_E_HIP, "gradient buffer init failed"));
    }
    if (T->dt.wit) {        // roots carry no witness
        hipError_t e = hipMemsetAsync(T->dt.wit, 0, (size_t)n_roots * (p + 2) * 8, P->stream);
        if (e != hipSuccess) RUN_TRY(fail(EHM_E_HIP, "witness buffer init failed"));
    }
    if (T->dt.mt.state) {   // an empty table (the payload needs no clearing)
        hipError_t e = hipMemsetAsync(T->dt.mt.state, 0, ((size_t)T->dt.mt.mask + 1) * 8, P->stream);
        if (e != hipSuccess) RUN_TRY(fail(EHM_E_HIP, "midpoint table init failed"));
    }
    {
        unsigned long long inf_bits = 0x7FF0000000000000ULL;
        (void)hipMemcpyAsync(&P->d_cnt->min_margin_bits, &inf_bits, 8, hipMemcpyHostToDevice,
                             P->stream);
        unsigned long long zero = 0;
        (void)hipMemcpyAsync(&P->d_cnt->errors, &zero, 8, hipMemcpyHostToDevice, P->stream);
    }
    (void)hipEventCreate(&R.ev0);
    (void)hipEventRecord(R.ev0, P->stream);
    R.n_roots = n_roots;
    R.n_nodes = n_roots;
    R.nf = n_roots;
    R.cur_is_a = true;
    const bool start_empty = R.shard_world > 1 && R.shard_rank > 0 && opts &&
                             opts->shard_min_frontier < 0;
    if (start_empty && P->dp.n_delta == 1) {
        // dynamic balancing from a single source: rank 0 owns the roots, the others start with an
        // empty frontier and are fed by the rebalancing rounds (ehm_partition_give)
        std::vector<uint8_t> fl((size_t)n_roots, (uint8_t)(2 | 4));
        HIP_TRY(hipMemcpyAsync(T->dt.flags, fl.data(), fl.size(), hipMemcpyHostToDevice,
                               P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
        R.nf = 0;
        R.sharded = true;
        R.pre_nodes = n_roots;
        P->active_run = T;
        *out = T;
        return EHM_OK;
    }
    if (R.shard_world > 1 && opts && opts->shard_min_frontier < 0) R.sharded = true;
    if (P->dp.n_delta > 1) {
        // multi-commutation problems: the device engine of ehm_hybrid.h
        RUN_TRY(hy_begin(T, n_roots, init));
        P->active_run = T;
        *out = T;
        return EHM_OK;
    }
    // 'ecc' for a single-commutation problem: the only commutation is feasible at every
    // vertex of a feasible Theta, so V_R reduces to the vertex solves (lib/worker.py:279-291)
    if (R.action == 0) {
        if (P->solver_gen == 2) {
            K2Cfg cfg;
            RUN_TRY(k2_config(P, LP_POINT, LP_POINT, n_roots * (p + 1), cfg));
            cfg.api->vertex(cfg.L, P->dp, T->dt, P->fr_a.as<int32_t>(), (int)n_roots, P->d_cnt);
        } else {
            hipLaunchKernelGGL(k_vertex_solve, dim3(grid_for(P, n_roots * (p + 1))), dim3(64),
                               P->lds_point, P->stream, P->dp, T->dt, P->fr_a.as<int32_t>(),
                               (int)n_roots, P->d_cnt);
        }
        P->launches++;
        R.ref_solves += n_roots * (2 + (p + 1));   // P_theta check + V_R MICP + vertex solves
        // the reference stops a run whose Theta is not feasible everywhere (barycentre check,
        // lib/worker.py:264-266); with one commutation a root vertex without a feasible point
        // is that condition -- report it now instead of growing a tree on garbage costs
        DevCounters cv;
        RUN_TRY(read_counters(P, cv));
        if (cv.errors != 0)
            RUN_TRY(fail(EHM_E_INFEASIBLE,
                         "STOP, Theta contains infeasible regions (%llu of %lld root-vertex "
                         "solves found no feasible point)",
                         (unsigned long long)cv.errors, (long long)(n_roots * (p + 1))));
    }
    P->active_run = T;
    *out = T;
    return EHM_OK;
#undef RUN_TRY
}

// engine = 1: the whole run in ONE launch of the persistent frontier kernel (ehm_k2.hip,
// k2_persist).  Single rank, shared-block kernels only; anything else uses the sweeps.
static int persistent_run(ehm_tree* T, long long max_pops = 0) {
    ehm_problem* P = T->prob;
    auto& R = T->run;
    K2Cfg cfg;
    int rc = k2_config(P, LP_SLACK, LP_POINT, 1LL << 40, cfg, true);   // the full persistent grid
    if (rc) return rc;
    if (!cfg.api->persist) return fail(EHM_E_INVALID, "no persistent kernel for this LP size");
    // two solver widths where a pair is compiled: the midpoint LPs have p + 1 columns less
    const KpApi* kp = nullptr;
    if (!P->quadratic && !getenv("EHM_NO_KP")) {
        int n_d, ne_d, n_e, ne_e;
        kind_dims(P->dp, LP_SLACK, n_d, ne_d);
        kind_dims(P->dp, LP_POINT, n_e, ne_e);
        const int nE = P->dp.n - P->dp.nd0;
        const K2Api* ad = k2_pick(n_d - nE, lp_slots(P->dp.m, ne_d), false);
        const K2Api* ae = k2_pick(n_e - nE, lp_slots(P->dp.m, ne_e), false);
        const int slots = std::max(lp_slots(P->dp.m, ne_d), lp_slots(P->dp.m, ne_e));
        if (ad && ae && ae->np < ad->np) {
            for (kp_getter g : g_kp_getters) {
                const KpApi* a = g();
                if (a->np_decide == ad->np && a->np_expand == ae->np && a->slots == slots) kp = a;
            }
            if (P->mid_first && !P->decide_full)
                for (kp_getter g : g_kpm_getters) {
                    const KpApi* a = g();
                    if (a->np_decide == ad->np && a->np_expand == ae->np && a->slots == slots)
                        kp = a;
                }
        }
        if (kp) {
            const size_t wave = kp->wave_doubles(P->dp, n_d, ne_d, n_e);
            const size_t budget = EHM_LDS_BUDGET / sizeof(double);
            // w and c in LDS unless they cost a wavefront (k2_config; at config 2 they do: the
            // twelfth wavefront of the workgroup)
            DevProblem dp = P->dp;
            dp.wc_lds = 1;
            const size_t shared1 = kp->shared_doubles(dp);
            dp.wc_lds = 0;
            const size_t shared0 = kp->shared_doubles(dp);
            const long long nw1 = (shared1 + wave <= budget)
                ? std::min<long long>(kp->max_threads / 64, (long long)((budget - shared1) / wave)) : 0;
            const long long nw0 = (shared0 + wave <= budget)
                ? std::min<long long>(kp->max_threads / 64, (long long)((budget - shared0) / wave)) : 0;
            const int wc_lds = (nw1 >= nw0 && nw1 >= 1) ? 1 : 0;
            const size_t shared = wc_lds ? shared1 : shared0;
            long long nw = wc_lds ? nw1 : nw0;
            if (shared + wave > budget || nw < 1) {
                kp = nullptr;
            } else {
                cfg.L.wc_lds = wc_lds;
                if (!P->kp_ready.count(kp)) {
                    HIP_TRY(kp->set_lds(EHM_LDS_BUDGET), EHM_E_HIP);
                    P->kp_ready.insert(kp);
                }
                const size_t lds = (shared + (size_t)nw * wave) * sizeof(double);
                const long long reg_wg = (kp->max_threads / 64) / nw;
                const long long wg_per_cu =
                    std::max<long long>(1, std::min<long long>(EHM_LDS_BUDGET / lds, reg_wg));
                cfg.L.grid = (int)((long long)P->num_cu * wg_per_cu);
                cfg.L.threads = (int)(64 * nw);
                cfg.L.lds_bytes = lds;
                cfg.L.wave_doubles = (int)wave;
            }
        }
    }
    const long long waves = (long long)cfg.L.grid * (cfg.L.threads / 64);
    // (2 x limit: every node is queued once and may be put back once -- its midpoint was being
    // solved elsewhere; node ids stay below the put-back mark EHM_REQUEUED)
    const long long n_slots = 2 * T->limit + waves + 64;
    if (n_slots > 0x7fffffffLL || T->limit >= (long long)EHM_REQUEUED)
        return fail(EHM_E_INVALID, "node pool too large for the persistent engine");
    if ((rc = P->pq_slots.ensure((size_t)n_slots * 4))) return rc;
    if ((rc = P->pq_ctl.ensure(sizeof(PersistCtl)))) return rc;
    int32_t* slots = P->pq_slots.as<int32_t>();
    HIP_TRY(hipMemsetAsync(slots, 0xFF, (size_t)n_slots * 4, P->stream), EHM_E_HIP);
    const int32_t* cur = (R.cur_is_a ? P->fr_a : P->fr_b).as<int32_t>();
    HIP_TRY(hipMemcpyAsync(slots, cur, (size_t)R.nf * 4, hipMemcpyDeviceToDevice, P->stream),
            EHM_E_HIP);
    PersistDeal deal{0, 0, 0, 1, 0, 0};
    deal.check = P->check_witness ? 1 : 0;
    T->dt.code = nullptr;
    if (R.deal_depth > 0 && R.shard_world > 1 && R.sweeps == 0) {
        // one launch from the roots, dealt over the ranks at a tree depth (PersistDeal)
        deal = PersistDeal{0, R.deal_depth, R.shard_rank, R.shard_world,
                           getenv("EHM_DEAL_LOW_BITS") ? 0 : 1, 0, P->check_witness ? 1 : 0};
        if ((rc = T->code.ensure((size_t)T->cap * 4))) return rc;
        T->dt.code = T->code.as<uint32_t>();
        std::vector<uint32_t> codes((size_t)R.n_roots);
        for (long long k = 0; k < R.n_roots; ++k) codes[(size_t)k] = (uint32_t)k;
        HIP_TRY(hipMemcpyAsync(T->dt.code, codes.data(), codes.size() * 4, hipMemcpyHostToDevice,
                               P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    }
    // budgeted launch (ehm_partition_advance): queue positions >= pop_limit stay unprocessed
    if (max_pops > 0 && max_pops < n_slots) deal.pop_limit = (int)max_pops;
    deal.keep = (P->work_first && !getenv("EHM_NO_WORKFIRST")) ? 1 : 0;
    // budgeted launches queue both children unless option "budget_keep" is set: a wavefront that
    // keeps a child follows its chain depth first, what the launch leaves behind the pop limit is
    // then a frontier of deep small cells, and the rebalancing rounds move 4 x the nodes for 8 %
    // more LPs (two gloo ranks on one GPU: 38.2 against 33.4 ms per partition,
    // profiles/r5/bench_2_gloo_ranks_dynamic_budget_keep.json)
    if (deal.pop_limit > 0 && !P->budget_keep) deal.keep = 0;
    PersistCtl h{};
    h.head = 0;
    h.tail = (int)R.nf;
    h.pending = (int)R.nf;
    h.n_nodes = (int)R.n_nodes;
    HIP_TRY(hipMemcpyAsync(P->pq_ctl.ptr, &h, sizeof h, hipMemcpyHostToDevice, P->stream),
            EHM_E_HIP);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, P->stream);
    (kp ? kp->persist : cfg.api->persist)(cfg.L, P->dp, T->dt, slots, (int)n_slots,
                                          P->pq_ctl.as<PersistCtl>(), (int)T->limit, P->d_cnt,
                                          P->decide_full ? 0 : 1, R.max_depth, deal);
    (void)hipEventRecord(e1, P->stream);
    R.evs.push_back(e0);
    R.evs.push_back(e1);
    R.ev_kind.push_back(0);
    P->launches++;
    HIP_TRY(hipGetLastError(), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(&h, P->pq_ctl.ptr, sizeof h, hipMemcpyDeviceToHost, P->stream),
            EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    if (h.abort == 1)
        return fail(EHM_E_CAPACITY, "node pool exhausted at %d nodes (max_nodes=%lld)",
                    h.n_nodes, T->limit);
    // what a budgeted launch leaves: the contiguous queue slice [pop_limit, tail)
    const long long left_over =
        (deal.pop_limit > 0 && h.tail > deal.pop_limit) ? (long long)h.tail - deal.pop_limit : 0;
    if (h.abort != 0 || h.pending != left_over)
        return fail(EHM_E_HIP, "persistent frontier kernel stopped early (abort=%d, pending=%d, "
                               "expected %lld)", h.abort, h.pending, left_over);
    if (left_over > 0) {
        DevBuf& fb = P->fr_a;
        if ((rc = fb.ensure((size_t)left_over * 4 * 2))) return rc;
        HIP_TRY(hipMemcpyAsync(fb.ptr, slots + deal.pop_limit, (size_t)left_over * 4,
                               hipMemcpyDeviceToDevice, P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
        R.cur_is_a = true;
    }
    R.ref_solves += (long long)h.closed + 3LL * (long long)h.splits;
    if (deal.world > 1) {
        // replicated on every rank: everything above the deal depth (and the nodes AT it, which
        // replicated parents created -- own ones and the other ranks' placeholders alike)
        R.pre_closed = (long long)h.repl_closed;
        R.pre_nodes = R.n_roots + 2LL * (long long)h.repl_splits;
        R.pre_solves = (long long)h.repl_solves;
        R.sharded = true;
    }
    R.n_closed += (long long)h.closed;
    R.n_nodes = h.n_nodes;
    R.truncated = R.truncated || h.truncated;
    R.depth = std::max(R.depth, h.max_depth_seen);
    R.sweeps = R.depth + 1;
    R.nf = left_over;
    T->unordered = true;
    return EHM_OK;
}

// Up to max_pops node visits of the persistent frontier kernel (<= 0: to completion), then the
// unprocessed part of its queue becomes the live frontier again -- what ehm_partition_take /
// ehm_partition_give move between ranks.  The rebalancing rounds of the multi-GPU driver
// (explicit_hybrid_mpc_amd/distributed.py, engine='persistent') are this call + one all-gather.
int ehm_partition_advance(ehm_tree* T, int64_t max_pops, int64_t* frontier_size) {
    if (!T || !T->run.active) return fail(EHM_E_INVALID, "no partition run in progress");
    ehm_problem* P = T->prob;
    auto& R = T->run;
    if (T->hy || P->solver_gen != 2 || P->dp.Wr3)
        return fail(EHM_E_INVALID, "ehm_partition_advance: needs a single-commutation problem on "
                                   "the shared-block kernels (use ehm_partition_step)");
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    int rc = EHM_OK;
    T->keep_ids = true;
    if (R.nf > 0) rc = persistent_run(T, max_pops > 0 ? (long long)max_pops : 0);
    if (frontier_size) *frontier_size = R.nf;
    return rc;
}

// Runs up to max_sweeps frontier sweeps (<= 0: until the frontier is empty).
int ehm_partition_step(ehm_tree* T, int32_t max_sweeps, int64_t* frontier_size) {
    if (!T || !T->run.active) return fail(EHM_E_INVALID, "no partition run in progress");
    ehm_problem* P = T->prob;
    auto& R = T->run;
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    if (T->hy) {
        int done_h = 0;
        while (R.nf > 0 && (max_sweeps <= 0 || done_h < max_sweeps)) {
            int rc = hy_sweep(T);
            if (rc) return rc;
            ++done_h;
        }
        if (frontier_size) *frontier_size = R.nf;
        return EHM_OK;
    }
    // persistent frontier kernel: a run that goes to completion in this call.  A sharded run
    // first sweeps until the frontier has been dealt over the ranks, then grows its share in
    // one launch (static dealing: no rebalancing rounds, see distributed.py)
    // (wide LPs: the LDS-resident family has a persistent kernel of its own -- single rank only,
    // it does not deal nodes over ranks; EHM_K4_PERSIST=0 keeps the level-synchronous sweeps)
    bool wide_persist = false;
    if (P->dp.Wr3 && R.engine == 1 && P->solver_gen == 2 && max_sweeps <= 0 &&
        R.shard_world == 1 && R.deal_depth <= 0) {
        static const int off = [] {
            const char* e = getenv("EHM_K4_PERSIST");
            return (e && atoi(e) == 0) ? 1 : 0;
        }();
        K2Cfg pc;
        wide_persist = !off && k2_config(P, LP_SLACK, LP_POINT, 1LL << 40, pc, true) == EHM_OK &&
                       pc.api->persist != nullptr;
    }
    const bool want_persist = R.engine == 1 && P->solver_gen == 2 && max_sweeps <= 0 &&
                              (!P->dp.Wr3 || wide_persist);
    if (want_persist && R.nf > 0 && (R.shard_world == 1 || R.sharded || R.deal_depth > 0)) {
        int rc = persistent_run(T);
        if (frontier_size) *frontier_size = 0;
        return rc;
    }
    DevBuf &fr_a = P->fr_a, &fr_b = P->fr_b, &open_flag = P->open_flag,
           &open_list = P->open_list, &d_count = P->d_count;
    auto stamp = [&]() {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, P->stream);
        R.evs.push_back(e);
    };
    int done = 0;
    while (R.nf > 0 && !R.truncated && (max_sweeps <= 0 || done < max_sweeps)) {
        int32_t* cur = (R.cur_is_a ? fr_a : fr_b).as<int32_t>();
        if (!R.sharded && R.nf >= R.shard_min) {
            // deal the frontier round-robin over the ranks; the kept ids go to the other buffer
            DevBuf& ob = R.cur_is_a ? fr_b : fr_a;
            int rc = ob.ensure((size_t)(R.nf / R.shard_world + 1) * 4);
            if (rc) return rc;
            hipLaunchKernelGGL(k_shard_filter, dim3((unsigned)((R.nf + 255) / 256)), dim3(256), 0,
                               P->stream, T->dt, cur, (int)R.nf, R.shard_rank, R.shard_world,
                               ob.as<int32_t>());
            P->launches++;
            cur = ob.as<int32_t>();
            R.cur_is_a = !R.cur_is_a;
            // the frontier that is dealt is the last level grown: ids [n_nodes - nf, n_nodes)
            R.pre_first = (R.sweeps == 0) ? 0 : R.n_nodes - R.nf;
            R.nf = (R.nf - R.shard_rank + R.shard_world - 1) / R.shard_world;
            R.sharded = true;
            // work done so far is replicated on every rank
            DevCounters cs;
            int rc2 = read_counters(P, cs);
            if (rc2) return rc2;
            R.pre_closed = R.n_closed;
            R.pre_nodes = R.n_nodes;
            R.pre_solves = (long long)(cs.lp_solves - R.c0.lp_solves);
            if (R.nf == 0) break;
            if (want_persist) {
                int rc3 = persistent_run(T);
                if (frontier_size) *frontier_size = 0;
                return rc3;
            }
        }
        if ((long long)open_flag.cap < R.nf * 4) {
            int rc = open_flag.ensure((size_t)R.nf * 4 * 2);
            if (rc) return rc;
            rc = open_list.ensure((size_t)R.nf * 4 * 2);
            if (rc) return rc;
        }
        K2Cfg cfg_d;
        if (P->solver_gen == 2) {
            int rc = k2_config(P, LP_SLACK, LP_SLACK, R.nf, cfg_d);
            if (rc) return rc;
        }
        stamp();
        if (P->solver_gen == 2)
            cfg_d.api->decide(cfg_d.L, P->dp, T->dt, cur, (int)R.nf, open_flag.as<int32_t>(),
                              P->d_cnt, P->decide_full ? 0 : 1);
        else
            hipLaunchKernelGGL(k_lcss_decide, dim3(grid_for(P, R.nf)), dim3(64), P->lds_simplex,
                               P->stream, P->dp, T->dt, cur, (int)R.nf, open_flag.as<int32_t>(),
                               P->d_cnt);
        stamp();
        R.ev_kind.push_back(0);
        hipLaunchKernelGGL(k_scan_open, dim3(1), dim3(1024), 0, P->stream,
                           open_flag.as<int32_t>(), cur, (int)R.nf, open_list.as<int32_t>(),
                           d_count.as<int32_t>());
        P->launches += 2;
        int32_t n_open = 0;
        {
            hipError_t e = hipMemcpyAsync(&n_open, d_count.ptr, 4, hipMemcpyDeviceToHost,
                                          P->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(P->stream);
            if (e != hipSuccess)
                return fail(EHM_E_HIP, "sweep %d failed: %s", R.sweeps, hipGetErrorString(e));
        }
        ++R.sweeps;
        ++done;
        R.ref_solves += R.nf;                 // one bar_E MICP per visited node
        R.n_closed += R.nf - n_open;
        if (n_open == 0) {
            R.nf = 0;
            break;
        }
        if (R.max_depth > 0 && R.depth >= R.max_depth) {
            R.truncated = 1;
            break;
        }
        if (R.n_nodes + 2LL * n_open > T->limit)
            return fail(EHM_E_CAPACITY, "node pool exhausted at %lld nodes (max_nodes=%lld)",
                        R.n_nodes, T->limit);
        // next frontier buffer must hold 2*n_open ids
        DevBuf& nb = R.cur_is_a ? fr_b : fr_a;
        if ((long long)nb.cap < 2LL * n_open * 4) {
            int rc = nb.ensure((size_t)n_open * 2 * 4 * 2);
            if (rc) return rc;
        }
        int32_t* nxt = nb.as<int32_t>();
        K2Cfg cfg_e;
        if (P->solver_gen == 2) {
            int rc = k2_config(P, LP_POINT, LP_POINT, n_open, cfg_e);
            if (rc) return rc;
        }
        stamp();
        if (P->solver_gen == 2)
            cfg_e.api->expand(cfg_e.L, P->dp, T->dt, open_list.as<int32_t>(), (int)n_open,
                              (int)R.n_nodes, nxt, P->d_cnt);
        else
            hipLaunchKernelGGL(k_lcss_expand, dim3(grid_for(P, n_open)), dim3(64), P->lds_expand,
                               P->stream, P->dp, T->dt, open_list.as<int32_t>(), (int)n_open,
                               (int)R.n_nodes, nxt, P->d_cnt);
        stamp();
        R.ev_kind.push_back(1);
        P->launches++;
        R.ref_solves += 2LL * n_open;       // bar_D MICP + midpoint P_theta_delta per split
        R.n_nodes += 2LL * n_open;
        R.nf = 2LL * n_open;
        R.cur_is_a = !R.cur_is_a;
        ++R.depth;
    }
    if (frontier_size) *frontier_size = R.truncated ? 0 : R.nf;
    return EHM_OK;
}

// Removes the LAST `count` nodes from the live frontier: their records [count][rec_doubles]
// (vertices | vertex costs | vertex inputs), meta [count][2] = (commutation index, depth) and
// node ids go to the caller; the nodes are flagged bit2 (owned by another rank).
// (node_ids / records / meta may be HOST or DEVICE pointers -- hipMemcpyDefault: a multi-GPU
// driver hands over device-resident blocks, e.g. torch tensors it then sends over xGMI)
int ehm_partition_take(ehm_tree* T, int64_t count, int32_t* node_ids, double* records,
                       int32_t* meta) {
    if (!T || !T->run.active || !node_ids || !records || !meta || count < 0)
        return fail(EHM_E_INVALID, "bad argument");
    ehm_problem* P = T->prob;
    auto& R = T->run;
    if (T->hy) {
        // the hashed draws of option "any_admissible" read the path code of a node, and the
        // hand-over does not carry it: a moved subtree would draw as if it started at code 0 and
        // the run would no longer repeat the single-GPU one (nor the CPU oracle's rule 'hash')
        if (P->any_admissible && count > 0)
            return fail(EHM_E_INVALID, "nodes cannot be handed over under option any_admissible "
                                       "(its draws are reproducible only without rebalancing)");
        HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
        return hy_take(T, count, node_ids, records, meta);
    }
    if (count > R.nf) return fail(EHM_E_INVALID, "cannot take %lld of %lld frontier nodes",
                                  (long long)count, (long long)R.nf);
    if (count == 0) return EHM_OK;
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    const int nrec = rec_doubles(P->dp.p, P->dp.n_u);
    int rc;
    if ((rc = P->out0.ensure((size_t)count * nrec * sizeof(double)))) return rc;
    if ((rc = P->out2.ensure((size_t)count * 2 * sizeof(int32_t)))) return rc;
    // A representative sample of the frontier, not its newest (deepest, smallest) nodes: every
    // s-th queue entry, s = nf / count.  The frontier is permuted into the other buffer as
    // [what stays | what goes] by strided device copies.
    const long long stride = R.nf / count;
    if (stride >= 2 && !getenv("EHM_TAKE_NEWEST")) {
        DevBuf& src = R.cur_is_a ? P->fr_a : P->fr_b;
        DevBuf& dst = R.cur_is_a ? P->fr_b : P->fr_a;
        if ((rc = dst.ensure((size_t)R.nf * 4 * 2))) return rc;
        const int32_t* a = src.as<int32_t>();
        int32_t* b = dst.as<int32_t>();
        const long long rest = R.nf - count, tail = R.nf - count * stride;
        HIP_TRY(hipMemcpy2DAsync(b + rest, 4, a, (size_t)stride * 4, 4, (size_t)count,
                                 hipMemcpyDeviceToDevice, P->stream), EHM_E_HIP);
        HIP_TRY(hipMemcpy2DAsync(b, (size_t)(stride - 1) * 4, a + 1, (size_t)stride * 4,
                                 (size_t)(stride - 1) * 4, (size_t)count,
                                 hipMemcpyDeviceToDevice, P->stream), EHM_E_HIP);
        if (tail > 0)
            HIP_TRY(hipMemcpyAsync(b + count * (stride - 1), a + count * stride, (size_t)tail * 4,
                                   hipMemcpyDeviceToDevice, P->stream), EHM_E_HIP);
        R.cur_is_a = !R.cur_is_a;
    }
    const int32_t* cur = (R.cur_is_a ? P->fr_a : P->fr_b).as<int32_t>() + (R.nf - count);
    hipLaunchKernelGGL(k_take_nodes, dim3((unsigned)count), dim3(64), 0, P->stream, T->dt, cur,
                       (int)count, nrec, P->out0.as<double>(), P->out2.as<int32_t>());
    HIP_TRY(hipMemcpyAsync(node_ids, cur, (size_t)count * 4, hipMemcpyDefault, P->stream),
            EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(records, P->out0.ptr, (size_t)count * nrec * sizeof(double),
                           hipMemcpyDefault, P->stream), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(meta, P->out2.ptr, (size_t)count * 2 * sizeof(int32_t),
                           hipMemcpyDefault, P->stream), EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    R.nf -= count;
    R.given += count;
    return EHM_OK;
}

// Appends `count` nodes received from another rank (records / meta as produced by
// ehm_partition_take) to the pool and to the live frontier; first_id = id of the first one.
int ehm_partition_give(ehm_tree* T, int64_t count, const double* records, const int32_t* meta,
                       int32_t* first_id) {
    if (!T || !T->run.active || !records || !meta || count < 0)
        return fail(EHM_E_INVALID, "bad argument");
    ehm_problem* P = T->prob;
    auto& R = T->run;
    if (T->hy) {
        if (P->any_admissible && count > 0)
            return fail(EHM_E_INVALID, "nodes cannot be handed over under option any_admissible "
                                       "(its draws are reproducible only without rebalancing)");
        HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
        return hy_give(T, count, records, meta, first_id);
    }
    if (first_id) *first_id = (int32_t)R.n_nodes;
    if (count == 0) return EHM_OK;
    if (R.n_nodes + count > T->limit)
        return fail(EHM_E_CAPACITY, "node pool exhausted at %lld nodes (max_nodes=%lld)",
                    R.n_nodes, T->limit);
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    const int nrec = rec_doubles(P->dp.p, P->dp.n_u);
    int rc;
    if ((rc = P->in0.ensure((size_t)count * nrec * sizeof(double)))) return rc;
    if ((rc = P->in1.ensure((size_t)count * 2 * sizeof(int32_t)))) return rc;
    DevBuf& fb = R.cur_is_a ? P->fr_a : P->fr_b;
    if ((long long)fb.cap < (R.nf + count) * 4) {
        // grow the live frontier buffer, keeping its contents
        DevBuf bigger;
        if ((rc = bigger.ensure((size_t)(R.nf + count) * 4 * 2))) return rc;
        HIP_TRY(hipMemcpyAsync(bigger.ptr, fb.ptr, (size_t)R.nf * 4, hipMemcpyDeviceToDevice,
                               P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
        std::swap(bigger, fb);
        bigger.release();
    }
    HIP_TRY(hipMemcpyAsync(P->in0.ptr, records, (size_t)count * nrec * sizeof(double),
                           hipMemcpyDefault, P->stream), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(P->in1.ptr, meta, (size_t)count * 2 * sizeof(int32_t),
                           hipMemcpyDefault, P->stream), EHM_E_HIP);
    hipLaunchKernelGGL(k_give_nodes, dim3((unsigned)count), dim3(64), 0, P->stream, T->dt,
                       (int)R.n_nodes, (int)count, nrec, P->in0.as<double>(),
                       P->in1.as<int32_t>(), fb.as<int32_t>(), (int)R.nf);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    R.n_nodes += count;
    R.nf += count;
    R.received += count;
    return EHM_OK;
}

// Pool occupancy of a run in progress (no device work): nodes allocated, nodes the caller allowed.
int ehm_partition_counts(const ehm_tree* T, int64_t* n_nodes, int64_t* max_nodes) {
    if (!T || !T->run.active) return fail(EHM_E_INVALID, "no partition run in progress");
    if (n_nodes) *n_nodes = T->run.n_nodes;
    if (max_nodes) *max_nodes = T->limit;
    return EHM_OK;
}

// What ehm_partition_take can hand over right now and how wide one record is: single-commutation
// runs move any frontier node as rec_doubles doubles; multi-commutation runs move lcss nodes only
// (ecc nodes carry no data yet), each with its bit rows appended (ehm_hybrid.h).
int ehm_partition_movable(const ehm_tree* T, int64_t* movable, int32_t* record_doubles) {
    if (!T || !T->run.active) return fail(EHM_E_INVALID, "no partition run in progress");
    const DevProblem& dp = T->prob->dp;
    if (movable) *movable = T->hy ? T->hy->n_lcss : T->run.nf;
    if (record_doubles)
        *record_doubles = T->hy ? hy_record_doubles(dp.p, dp.n_u, T->hy->nw)
                                : rec_doubles(dp.p, dp.n_u);
    return EHM_OK;
}

int ehm_partition_progress(ehm_tree* T, ehm_progress* out) {
    if (!T || !out) return fail(EHM_E_INVALID, "null argument");
    if (!T->run.active) return fail(EHM_E_INVALID, "no partition run in progress");
    ehm_problem* P = T->prob;
    auto& R = T->run;
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    DevCounters c1;
    int rc = read_counters(P, c1);      // synchronises the stream
    if (rc) return rc;
    // multi-GPU: the top of the tree is grown identically on every rank until the frontier is
    // dealt; rank 0 reports it, the others only what they grew themselves
    // (its share of the dealt frontier -- still undecided then -- and everything appended since)
    const long long n = R.n_nodes;
    long long first = 0, own_nodes = n, own_closed = R.n_closed;
    if (R.shard_world > 1 && R.shard_rank > 0) {
        first = R.sharded ? R.pre_first : n;
        own_nodes = R.sharded ? n - R.pre_nodes : 0;
        own_closed = R.sharded ? R.n_closed - R.pre_closed : 0;
    }
    const int blocks = (int)((n - first + 255) / 256);
    double vol = 0.0;
    if (blocks > 0) {
        rc = P->out3.ensure((size_t)blocks * sizeof(double));
        if (rc) return rc;
        hipLaunchKernelGGL(k_closed_volume, dim3(blocks), dim3(256), 0, P->stream, T->dt, first,
                           n, P->out3.as<double>());
        std::vector<double> part((size_t)blocks);
        HIP_TRY(hipMemcpyAsync(part.data(), P->out3.ptr, part.size() * sizeof(double),
                               hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
        for (double v : part) vol += v;
    }
    out->n_nodes = own_nodes;
    // every node that is neither a root nor received from another rank is one of two children
    out->n_splits = (R.shard_world > 1 && R.shard_rank > 0)
                        ? (R.sharded ? (n - R.pre_nodes - R.received) / 2 : 0)
                        : (n - R.n_roots - R.received) / 2;
    out->n_closed = own_closed;
    out->frontier = R.nf;
    out->sweeps = R.sweeps;
    out->depth = R.depth;
    out->lp_solves = (int64_t)(c1.lp_solves - R.c0.lp_solves);
    out->ipm_iters = (int64_t)(c1.ipm_iters - R.c0.ipm_iters);
    out->volume_closed = vol;
    return EHM_OK;
}

int ehm_partition_finish(ehm_tree* T) {
    if (!T || !T->run.active) return fail(EHM_E_INVALID, "no partition run in progress");
    ehm_problem* P = T->prob;
    auto& R = T->run;
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    hipEvent_t ev1;
    (void)hipEventCreate(&ev1);
    (void)hipEventRecord(ev1, P->stream);
    (void)hipEventSynchronize(ev1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, R.ev0, ev1);
    (void)hipEventDestroy(ev1);
    double t_kind[2] = {0.0, 0.0};
    long long n_kind[2] = {0, 0};
    for (size_t k = 0; k < R.ev_kind.size(); ++k) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, R.evs[2 * k], R.evs[2 * k + 1]);
        t_kind[R.ev_kind[k]] += t * 1e-3;
        n_kind[R.ev_kind[k]]++;
    }
    run_release_events(T);
    DevCounters c1;
    int rc = read_counters(P, c1);
    if (rc) return rc;
    R.active = false;
    if (P->active_run == T) P->active_run = nullptr;
    if (c1.errors != 0 && !getenv("EHM_KEEP_GOING"))
        return fail(EHM_E_NUMERIC, "%llu oracle solves did not converge",
                    (unsigned long long)c1.errors);
    const long long received = R.received;
    T->info.n_nodes = R.n_nodes;
    T->info.n_roots = R.n_roots;
    // every non-root node that was not received is one of two children of a split
    T->info.n_leaves = R.n_nodes - (R.n_nodes - R.n_roots - received) / 2;
    T->info.n_closed = R.n_closed;
    T->info.lp_solves = (int64_t)(c1.lp_solves - R.c0.lp_solves);
    T->info.ipm_iters = (int64_t)(c1.ipm_iters - R.c0.ipm_iters);
    T->info.ref_solves = R.ref_solves;
    T->info.sweeps = R.sweeps;
    T->info.max_depth = R.depth;
    T->info.truncated = R.truncated;
    T->info.device_seconds = ms * 1e-3;
    T->info.decide_seconds = t_kind[0];
    T->info.expand_seconds = t_kind[1];
    T->info.decide_launches = n_kind[0];
    T->info.expand_launches = n_kind[1];
    T->info.decide_solves = (int64_t)(c1.slack_solves - R.c0.slack_solves);
    T->info.decide_iters = (int64_t)(c1.slack_iters - R.c0.slack_iters);
    T->info.cert_closed = (int64_t)(c1.cert_closed - R.c0.cert_closed);
    T->info.witness_open = (int64_t)(c1.wit_open - R.c0.wit_open);
    T->info.witness_inherited = (int64_t)(c1.wit_inherited - R.c0.wit_inherited);
    T->info.midpoints_shared = (int64_t)(c1.mid_shared - R.c0.mid_shared);
    T->info.witness_table = (int64_t)(c1.wit_table - R.c0.wit_table);
    for (int k = 0; k < 10; ++k) T->info.persist_ticks[k] = (int64_t)(c1.prof[k] - R.c0.prof[k]);
    T->info.near_threshold = (int64_t)(c1.routed - R.c0.routed);
    T->info.replicated_closed = R.pre_closed;
    T->info.replicated_nodes = R.pre_nodes;
    T->info.replicated_solves = R.pre_solves;
    {
        double mm;
        std::memcpy(&mm, &c1.min_margin_bits, 8);
        T->info.min_margin = mm;
    }
    for (int k = 0; k < 5; ++k) T->info.kind_solves[k] = T->info.kind_iters[k] = 0;
    T->info.kind_solves[LP_SLACK] = T->info.decide_solves;
    T->info.kind_iters[LP_SLACK] = T->info.decide_iters;
    T->info.kind_solves[LP_POINT] = T->info.lp_solves - T->info.decide_solves;
    T->info.kind_iters[LP_POINT] = T->info.ipm_iters - T->info.decide_iters;
    T->info.swaps = 0;
    T->info.blacklisted = 0;
    if (T->hy) {
        rc = hy_kind_totals(T, T->info.kind_solves, T->info.kind_iters);
        if (rc) return rc;
        T->info.decide_solves = T->info.kind_solves[LP_MIN_SIMPLEX] + T->info.kind_solves[LP_SLACK] +
                                T->info.kind_solves[LP_FEAS_SIMPLEX];
        T->info.decide_iters = T->info.kind_iters[LP_MIN_SIMPLEX] + T->info.kind_iters[LP_SLACK] +
                               T->info.kind_iters[LP_FEAS_SIMPLEX];
        const HyCtr& h = T->hy->h;
        double mm;
        std::memcpy(&mm, &h.min_margin_bits, 8);
        T->info.min_margin = mm;
        T->info.swaps = (int64_t)h.swaps;
        T->info.blacklisted = (int64_t)h.blacklisted;
        T->info.near_threshold = (int64_t)h.routed;
        P->slivers += (long long)h.slivers;
        P->fallbacks += (long long)h.fallbacks;
    }
    T->info.volume_closed = -1.0;   // filled lazily by ehm_tree_info_get
    return EHM_OK;
}

int ehm_partition_run(ehm_problem* P, int64_t n_roots, const double* root_vertices,
                      const ehm_node_init* init, const ehm_run_opts* opts, ehm_tree** out) {
    if (!out) return fail(EHM_E_INVALID, "bad argument");
    ehm_tree* T = nullptr;
    int rc = ehm_partition_begin(P, n_roots, root_vertices, init, opts, &T);
    if (rc) return rc;
    rc = ehm_partition_step(T, 0, nullptr);
    if (!rc) rc = ehm_partition_finish(T);
    if (rc) {
        std::string keep = g_err;      // the destroy below must not lose the message
        run_release_events(T);
        ehm_tree_destroy(T);
        g_err = keep;
        *out = nullptr;
        return rc;
    }
    *out = T;
    return EHM_OK;
}

int ehm_tree_info_get(const ehm_tree* Tc, ehm_tree_info* out) {
    if (!Tc || !out) return fail(EHM_E_INVALID, "null argument");
    if (!Tc->prob) return fail(EHM_E_INVALID, "the problem handle of this tree was destroyed");
    ehm_tree* T = const_cast<ehm_tree*>(Tc);
    if (T->info.volume_closed < 0.0 && !T->skip_volume) {
        // sum of closed-leaf volumes (lib/worker.py:374-375): a reduction kernel over the pool
        ehm_problem* P = T->prob;
        HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
        const long long n = T->info.n_nodes;
        const int blocks = (int)((n + 255) / 256);
        double vol = 0.0;
        if (blocks > 0) {
            int rc = P->out3.ensure((size_t)blocks * sizeof(double));
            if (rc) return rc;
            hipLaunchKernelGGL(k_closed_volume, dim3(blocks), dim3(256), 0, P->stream, T->dt, 0LL,
                               n, P->out3.as<double>());
            std::vector<double> part((size_t)blocks);
            HIP_TRY(hipMemcpyAsync(part.data(), P->out3.ptr, part.size() * sizeof(double),
                                   hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
            HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
            for (double v : part) vol += v;
        }
        T->info.volume_closed = vol;
    }
    *out = T->info;
    return EHM_OK;
}

int ehm_tree_export(const ehm_tree* Tc, double* vertices, int32_t* left, int32_t* right,
                    int32_t* delta_idx, double* vcost, double* vinput, uint8_t* flags,
                    double* tstar) {
    if (!Tc) return fail(EHM_E_INVALID, "null tree");
    if (!Tc->prob) return fail(EHM_E_INVALID, "the problem handle of this tree was destroyed");
    ehm_tree* T = const_cast<ehm_tree*>(Tc);
    ehm_problem* P = T->prob;
    HIP_TRY(hipSetDevice(P->device), EHM_E_HIP);
    const long long n = T->info.n_nodes;
    if (n <= 0) return EHM_OK;
    const int p = P->dp.p, n_u = P->dp.n_u;
    const int nR = (p + 1) * p;
    const bool relabel = T->unordered && !T->keep_ids;
    int rc = EHM_OK;
    if (relabel && !T->have_perm) {
        // breadth-first numbering, level by level ON THE DEVICE (k_bfs_*): no copy of the
        // structure to the host, no host loop over the nodes
        if ((rc = T->d_perm.ensure((size_t)n * 4)) || (rc = T->d_inv.ensure((size_t)n * 4))) return rc;
        if ((rc = P->ex_tmp.ensure((size_t)(BFS_BLOCKS + 8) * 4))) return rc;
        int32_t* bsum = P->ex_tmp.as<int32_t>();
        int32_t* bounds = bsum + BFS_BLOCKS;
        int32_t* perm = T->d_perm.as<int32_t>();
        const int n_roots = (int)T->info.n_roots;
        hipLaunchKernelGGL(k_bfs_roots, dim3((unsigned)((n_roots + 255) / 256)), dim3(256), 0, P->stream,
                           perm, bounds, n_roots);
        const int levels = T->info.max_depth + 2;       // (the last ones are empty)
        for (int d = 0; d < levels; ++d) {
            const int par = d & 1;
            hipLaunchKernelGGL(k_bfs_count, dim3(BFS_BLOCKS), dim3(BFS_THREADS), 0, P->stream,
                               T->dt.left, perm, bounds, par, bsum);
            hipLaunchKernelGGL(k_bfs_scan, dim3(1), dim3(BFS_BLOCKS), 0, P->stream, bsum, bounds, par);
            hipLaunchKernelGGL(k_bfs_write, dim3(BFS_BLOCKS), dim3(BFS_THREADS), 0, P->stream,
                               T->dt.left, perm, bounds, par, bsum);
        }
        int32_t hb[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(hb, bounds, sizeof hb, hipMemcpyDeviceToHost, P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
        const int reached = hb[2 * (levels & 1) + 1];
        if (reached != n || hb[2 * (levels & 1)] != reached)
            return fail(EHM_E_HIP, "tree structure inconsistent (%d of %lld nodes numbered in %d levels)",
                        reached, n, levels);
        hipLaunchKernelGGL(k_bfs_invert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, P->stream,
                           perm, T->d_inv.as<int32_t>(), n);
        T->have_perm = true;
    }
    // The records are gathered ON THE DEVICE into the caller's layout and numbering -- ONE pass
    // over all nodes into a staging buffer that lives with the problem handle (no allocation per
    // export) -- and copied out array by array with ONE wait at the end.  Callers that hand over
    // pinned memory (ehm_host_alloc) get the copies at the speed of the host link.
    const int nc = p + 1, nu = (p + 1) * n_u, nrec = nR + nc + nu;
    const size_t per_node = (size_t)nrec * 8 + 8 + 4 + 4 + 4 + 1;
    // (a pool too large for one staging buffer goes in parts, one wait each; the buffer stays with
    // the handle, so it is capped by BYTES -- 768 MB: the headline tree, 483 MB, still goes in one
    // piece -- not by nodes: at ~1 KB per node of a p = 8 law 2^22 nodes were 4 GB held for good)
    const long long cap_nodes = std::max<long long>(1, (768LL << 20) / (long long)per_node);
    const long long chunk = std::min<long long>(n, std::min<long long>(1LL << 22, cap_nodes));
    if ((rc = P->ex_stage.ensure((size_t)chunk * per_node + 64))) return rc;
    for (long long k0 = 0; k0 < n; k0 += chunk) {
        const long long nk = std::min(chunk, n - k0);
        double* s_v = P->ex_stage.as<double>();
        double* s_c = s_v + (size_t)nk * nR;
        double* s_u = s_c + (size_t)nk * nc;
        double* s_t = s_u + (size_t)nk * nu;
        int32_t* s_l = reinterpret_cast<int32_t*>(s_t + nk);
        int32_t* s_r = s_l + nk;
        int32_t* s_d = s_r + nk;
        uint8_t* s_f = reinterpret_cast<uint8_t*>(s_d + nk);
        hipLaunchKernelGGL(k_export_gather, dim3((unsigned)((nk * nrec + 255) / 256)), dim3(256), 0,
                           P->stream, T->dt, relabel ? T->d_perm.as<int32_t>() : (const int32_t*)nullptr,
                           relabel ? T->d_inv.as<int32_t>() : (const int32_t*)nullptr, k0, nk, n_u, s_v,
                           s_c, s_u, s_t, s_l, s_r, s_d, s_f);
        hipError_t e = hipGetLastError();
#define EXP_COPY(dst, srcp, bytes)                                                         \
        if (e == hipSuccess && (dst))                                                      \
            e = hipMemcpyAsync((dst), (srcp), (bytes), hipMemcpyDeviceToHost, P->stream)
        EXP_COPY(vertices ? vertices + (size_t)k0 * nR : nullptr, s_v, (size_t)nk * nR * 8);
        EXP_COPY(vcost ? vcost + (size_t)k0 * nc : nullptr, s_c, (size_t)nk * nc * 8);
        EXP_COPY(vinput ? vinput + (size_t)k0 * nu : nullptr, s_u, (size_t)nk * nu * 8);
        EXP_COPY(tstar ? tstar + k0 : nullptr, s_t, (size_t)nk * 8);
        EXP_COPY(left ? left + k0 : nullptr, s_l, (size_t)nk * 4);
        EXP_COPY(right ? right + k0 : nullptr, s_r, (size_t)nk * 4);
        EXP_COPY(delta_idx ? delta_idx + k0 : nullptr, s_d, (size_t)nk * 4);
        EXP_COPY(flags ? flags + k0 : nullptr, s_f, (size_t)nk);
#undef EXP_COPY
        if (e == hipSuccess) e = hipStreamSynchronize(P->stream);
        if (e != hipSuccess) return fail(EHM_E_HIP, "tree export failed: %s", hipGetErrorString(e));
    }
    return EHM_OK;
}

// Page-locked host memory for the arrays ehm_tree_export fills: copies into it run at the speed
// of the host link (into pageable memory they are staged by the runtime, several times slower).
int ehm_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(EHM_E_INVALID, "ehm_host_alloc: out is NULL");
    *out = nullptr;
    void* q = nullptr;
    if (hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
        return fail(EHM_E_CAPACITY, "ehm_host_alloc: %zu bytes of page-locked memory refused", bytes);
    *out = q;
    return EHM_OK;
}
int ehm_host_free(void* q) {
    if (q && hipHostFree(q) != hipSuccess) return fail(EHM_E_HIP, "ehm_host_free failed");
    return EHM_OK;
}

}  // extern "C"
