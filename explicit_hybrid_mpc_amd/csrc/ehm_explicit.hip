// Batched evaluation of the explicit control law: the consumer of the
// partition, ExplicitMPC.__call__ of the reference (lib/mpc_library.py:685-792), for whole
// batches of states on the device.
//
// Reference semantics kept: the partition is one binary tree; at an internal node the query
// goes LEFT iff it lies in the left child's simplex (all barycentric weights in
// [-eps, 1+eps], eps = machine epsilon, lib/mpc_library.py:714-735), otherwise right without a
// test; at the leaf the input is the barycentric interpolation of the vertex inputs
// (lib/mpc_library.py:786-789).  The root simplices hang off a right spine (tools.delaunay,
// lib/tools.py:152-189): root i is the left child of spine node i, the last two roots share
// the last spine node -- so the walk over the spine is "first root 0..Nsx-2 that contains x,
// else the last root".
//
// Layout: one 64-byte-aligned record per node, [v0 (p) | Minv (p x p, row-major)] with
// Minv = inv([v1-v0 ... vp-v0]) (computed on the device once, Gauss-Jordan with partial
// pivoting), child pair (left, right) as int2, vertex inputs (p+1) x n_u.  One thread per
// query: a walk reads one record per level (top levels L2-resident), i.e. the kernel is
// latency/HBM-bound, not arithmetic-bound.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ehmpc.h"

#define EHM_XP 8   // max parameter dimension (EHM_MAX_P)
#define EHM_X_LOCATE_MIN 128     // spines at least this long get the root locator

namespace {

struct DevExplicit {
    const double* rec;       // [n_nodes][rec_stride]: v0 | Minv
    const int2* child;       // (left, right), -1 for a leaf
    const double* vinput;    // [n_nodes][(p+1) n_u]
    int rec_stride, p, n_u, n_roots;
    long long n_nodes;
};

// Minv per node from its vertices [(p+1) p]
__global__ void k_explicit_setup(long long n_nodes, int p, int rec_stride,
                                 const double* __restrict__ vertices, double* __restrict__ rec,
                                 int32_t* __restrict__ singular) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) return;
    const double* V = vertices + (size_t)k * (p + 1) * p;
    double A[EHM_XP * 2 * EHM_XP];       // [E | I], E[r][q] = V[q+1][r] - V[0][r]
    const int w = 2 * p;
    for (int r = 0; r < p; ++r)
        for (int c = 0; c < w; ++c)
            A[r * w + c] = (c < p) ? (V[(c + 1) * p + r] - V[r]) : ((c - p == r) ? 1.0 : 0.0);
    bool bad = false;
    for (int c = 0; c < p; ++c) {
        int piv = c;
        double best = fabs(A[c * w + c]);
        for (int r = c + 1; r < p; ++r)
            if (fabs(A[r * w + c]) > best) {
                best = fabs(A[r * w + c]);
                piv = r;
            }
        if (best == 0.0) {
            bad = true;
            break;
        }
        if (piv != c)
            for (int q = 0; q < w; ++q) {
                const double t = A[c * w + q];
                A[c * w + q] = A[piv * w + q];
                A[piv * w + q] = t;
            }
        const double rp = 1.0 / A[c * w + c];
        for (int q = 0; q < w; ++q) A[c * w + q] *= rp;
        for (int r = 0; r < p; ++r) {
            if (r == c) continue;
            const double f = A[r * w + c];
            for (int q = 0; q < w; ++q) A[r * w + q] = fma(-f, A[c * w + q], A[r * w + q]);
        }
    }
    double* out = rec + (size_t)k * rec_stride;
    for (int c = 0; c < p; ++c) out[c] = V[c];
    for (int r = 0; r < p; ++r)
        for (int c = 0; c < p; ++c) out[p + r * p + c] = bad ? 0.0 : A[r * w + p + c];
    if (bad) atomicAdd(singular, 1);
}

// barycentric weights of x in node k; returns containment (lib/mpc_library.py:714-735)
__device__ __forceinline__ bool weights(const DevExplicit& E, long long k, const double* x,
                                        double* alpha, double& alpha0) {
    const double* r = E.rec + (size_t)k * E.rec_stride;
    const int p = E.p;
    const double eps = 2.220446049250313e-16;
    double d[EHM_XP];
    for (int c = 0; c < p; ++c) d[c] = x[c] - r[c];
    double s = 0.0;
    bool in = true;
    for (int q = 0; q < p; ++q) {
        double a = 0.0;
        for (int c = 0; c < p; ++c) a += r[p + q * p + c] * d[c];     // Minv.dot(x - c)
        alpha[q] = a;
        s += a;
        in = in && (a >= -eps) && (a <= 1.0 + eps);
    }
    alpha0 = 1.0 - s;
    return in && (alpha0 >= -eps) && (alpha0 <= 1.0 + eps);
}

// Containment only, the coordinates tested one after the other and the test left at the first
// weight outside [-eps, 1+eps]: the SAME sums in the same order as `weights`, so the verdict is
// bit-identical -- but a root that does not hold x is usually dismissed after one or two rows of
// Minv instead of p (the spine walk reads 3p instead of p + p^2 doubles per dismissed root).
__device__ __forceinline__ bool contains(const DevExplicit& E, long long k, const double* x) {
    const double* r = E.rec + (size_t)k * E.rec_stride;
    const int p = E.p;
    const double eps = 2.220446049250313e-16;
    double d[EHM_XP];
    for (int c = 0; c < p; ++c) d[c] = x[c] - r[c];
    double s = 0.0;
    for (int q = 0; q < p; ++q) {
        double a = 0.0;
        for (int c = 0; c < p; ++c) a += r[p + q * p + c] * d[c];
        if (!((a >= -eps) && (a <= 1.0 + eps))) return false;
        s += a;
    }
    const double a0 = 1.0 - s;
    return (a0 >= -eps) && (a0 <= 1.0 + eps);
}

// Root locator for long spines (tools.delaunay of a p = 8 box: 34 871 roots).  The reference walks
// the right spine and takes the FIRST root, in spine order, that contains x
// (lib/mpc_library.py:737-767): 17 000 containment tests per state on that spine, and one
// wavefront per query testing 64 roots at a time costs the same lane-steps (measured: 0.85 M
// against 0.95 M states/s).  The roots are a triangulation, so a state in the INTERIOR of a root
// (every barycentric weight > EHM_X_STRICT) is in no other root and the first one in spine order
// is that one -- found by a visibility walk over the face adjacency of the roots (built at
// set-up): from the current root cross the face opposite its most negative weight.  A state
// within EHM_X_STRICT of a face, a walk that leaves the hull or does not end in EHM_X_STEPS steps
// gets root = -1 and the reference's serial walk (k_explicit_eval): same leaf in every case.
#define EHM_X_STRICT 1e-9
#define EHM_X_STEPS 96
__global__ void k_explicit_locate(DevExplicit E, long long n, const double* __restrict__ X,
                                  const int32_t* __restrict__ nbr, int32_t* __restrict__ root) {
#pragma clang fp contract(off)
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int p = E.p;
    double x[EHM_XP], alpha[EHM_XP], a0;
    for (int c = 0; c < p; ++c) x[c] = X[q * p + c];
    int k = (int)(q % E.n_roots);           // any start: the walk is short from everywhere
    int found = -1, steps = 0;
    for (int step = 0; step < EHM_X_STEPS; ++step) {
        ++steps;
        (void)weights(E, k, x, alpha, a0);
        double lo = a0;
        int at = 0;
        for (int i = 0; i < p; ++i)
            if (alpha[i] < lo) {
                lo = alpha[i];
                at = i + 1;
            }
        if (lo > EHM_X_STRICT) {            // strictly inside: the only root that holds x
            found = k;
            break;
        }
        if (lo >= -EHM_X_STRICT) break;     // on a face: the serial walk decides
        const int k2 = nbr[(size_t)k * (p + 1) + at];
        if (k2 < 0) break;                  // left the hull (x outside the set, or rounding)
        k = k2;
    }
    // (the containment tests made ride in the upper bits: `visited` reports the device's work)
    root[q] = (found < 0) ? -1 : (found | (steps << 20));
}

__global__ void k_explicit_eval(DevExplicit E, long long n, const double* __restrict__ X,
                                double* __restrict__ U, int32_t* __restrict__ leaf,
                                int32_t* __restrict__ depth_out,
                                const int32_t* __restrict__ root) {
#pragma clang fp contract(off)
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int p = E.p, n_u = E.n_u;
    double x[EHM_XP], alpha[EHM_XP], a0;
    for (int c = 0; c < p; ++c) x[c] = X[q * p + c];
    // spine: first root that contains x, the last one without a test (found by k_explicit_locate
    // where the spine is long; `visited` counts the tests the reference's walk makes)
    long long k = E.n_roots - 1;
    int visited = 0;
    if (root && root[q] >= 0) {
        k = root[q] & 0xfffff;
        visited = root[q] >> 20;
    } else {
        for (int r = 0; r + 1 < E.n_roots; ++r) {
            ++visited;
            if (contains(E, r, x)) {
                k = r;
                break;
            }
        }
    }
    // partition subtree: left iff inside the left child
    for (;;) {
        const int2 ch = E.child[k];
        if (ch.x < 0) break;
        ++visited;
        k = contains(E, ch.x, x) ? ch.x : ch.y;
    }
    (void)weights(E, k, x, alpha, a0);
    const double* vi = E.vinput + (size_t)k * (p + 1) * n_u;
    for (int c = 0; c < n_u; ++c) {
        double u = a0 * vi[c];
        for (int i = 0; i < p; ++i) u += vi[(i + 1) * n_u + c] * alpha[i];
        U[q * n_u + c] = u;
    }
    if (leaf) leaf[q] = (int32_t)k;
    if (depth_out) depth_out[q] = visited;
}

thread_local std::string x_err;
int xfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    x_err = buf;
    return code;
}

}  // namespace

struct ehm_explicit {
    int device = 0;
    DevExplicit d{};
    void *rec = nullptr, *child = nullptr, *vinput = nullptr;
    void *x = nullptr, *u = nullptr, *leaf = nullptr, *depth = nullptr, *root = nullptr;
    void* nbr = nullptr;        // [n_roots][p+1] root across the face opposite vertex i (-1: hull)
    size_t cap = 0;
    hipStream_t stream = nullptr;
};

extern "C" {

const char* ehm_explicit_last_error(void) { return x_err.c_str(); }

int ehm_explicit_destroy(ehm_explicit* E) {
    if (!E) return EHM_OK;
    (void)hipSetDevice(E->device);
    for (void* p : {E->rec, E->child, E->vinput, E->x, E->u, E->leaf, E->depth, E->root, E->nbr})
        if (p) (void)hipFree(p);
    if (E->stream) (void)hipStreamDestroy(E->stream);
    delete E;
    return EHM_OK;
}

#define X_TRY(expr)                                                                       \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ehm_explicit_destroy(E);                                                      \
            return xfail(EHM_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));       \
        }                                                                                 \
    } while (0)

int ehm_explicit_create(int device, int64_t n_nodes, int32_t n_roots, int32_t p, int32_t n_u,
                        const int32_t* left, const int32_t* right, const double* vertices,
                        const double* vinput, ehm_explicit** out) {
    if (!out || !left || !right || !vertices || !vinput || n_nodes < 1 || n_roots < 1 ||
        n_roots > n_nodes || p < 1 || p > EHM_XP || n_u < 1)
        return xfail(EHM_E_INVALID, "bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return xfail(EHM_E_NO_DEVICE, "no HIP device %d (libehmpc has no CPU fallback)", device);
    ehm_explicit* E = new ehm_explicit();
    E->device = device;
    X_TRY(hipSetDevice(device));
    X_TRY(hipStreamCreate(&E->stream));
    const int stride = ((p + p * p + 7) / 8) * 8;
    const size_t nV = (size_t)n_nodes * (p + 1) * p, nU = (size_t)n_nodes * (p + 1) * n_u;
    void* d_vert = nullptr;
    int32_t* d_sing = nullptr;
    X_TRY(hipMalloc(&E->rec, (size_t)n_nodes * stride * sizeof(double)));
    X_TRY(hipMalloc(&E->child, (size_t)n_nodes * sizeof(int2)));
    X_TRY(hipMalloc(&E->vinput, nU * sizeof(double)));
    X_TRY(hipMalloc(&d_vert, nV * sizeof(double)));
    X_TRY(hipMalloc((void**)&d_sing, sizeof(int32_t)));
    std::vector<int2> ch((size_t)n_nodes);
    for (int64_t k = 0; k < n_nodes; ++k) ch[(size_t)k] = make_int2(left[k], right[k]);
    X_TRY(hipMemcpy(E->child, ch.data(), ch.size() * sizeof(int2), hipMemcpyHostToDevice));
    X_TRY(hipMemcpy(E->vinput, vinput, nU * sizeof(double), hipMemcpyHostToDevice));
    X_TRY(hipMemcpy(d_vert, vertices, nV * sizeof(double), hipMemcpyHostToDevice));
    X_TRY(hipMemset(d_sing, 0, sizeof(int32_t)));
    hipLaunchKernelGGL(k_explicit_setup, dim3((unsigned)((n_nodes + 127) / 128)), dim3(128), 0,
                       E->stream, (long long)n_nodes, (int)p, stride, (const double*)d_vert,
                       (double*)E->rec, d_sing);
    int32_t sing = 0;
    X_TRY(hipMemcpyAsync(&sing, d_sing, sizeof sing, hipMemcpyDeviceToHost, E->stream));
    X_TRY(hipStreamSynchronize(E->stream));
    (void)hipFree(d_vert);
    (void)hipFree(d_sing);
    if (sing) {
        ehm_explicit_destroy(E);
        return xfail(EHM_E_NUMERIC, "%d degenerate simplices in the partition", (int)sing);
    }
    if (n_roots >= EHM_X_LOCATE_MIN && n_roots < (1 << 20)) {
        // face adjacency of the roots: vertices by value, faces by their sorted vertex ids
        std::unordered_map<std::string, int32_t> vid;
        std::vector<int32_t> ids((size_t)n_roots * (p + 1));
        for (int64_t r = 0; r < n_roots; ++r)
            for (int i = 0; i <= p; ++i) {
                std::string key((const char*)(vertices + ((size_t)r * (p + 1) + i) * p),
                                sizeof(double) * p);
                auto it = vid.find(key);
                if (it == vid.end()) it = vid.emplace(std::move(key), (int32_t)vid.size()).first;
                ids[(size_t)r * (p + 1) + i] = it->second;
            }
        std::vector<int32_t> nbr((size_t)n_roots * (p + 1), -1);
        std::unordered_map<std::string, int64_t> face;      // key -> root * (p+1) + i of the first owner
        std::vector<int32_t> f((size_t)p);
        for (int64_t r = 0; r < n_roots; ++r)
            for (int i = 0; i <= p; ++i) {
                int w = 0;
                for (int j = 0; j <= p; ++j)
                    if (j != i) f[(size_t)w++] = ids[(size_t)r * (p + 1) + j];
                std::sort(f.begin(), f.end());
                std::string key((const char*)f.data(), sizeof(int32_t) * p);
                auto it = face.find(key);
                if (it == face.end()) {
                    face.emplace(std::move(key), r * (p + 1) + i);
                } else {
                    const int64_t o = it->second;
                    nbr[(size_t)r * (p + 1) + i] = (int32_t)(o / (p + 1));
                    nbr[(size_t)o] = (int32_t)r;
                }
            }
        X_TRY(hipMalloc(&E->nbr, nbr.size() * sizeof(int32_t)));
        X_TRY(hipMemcpy(E->nbr, nbr.data(), nbr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    E->d.rec = (const double*)E->rec;
    E->d.child = (const int2*)E->child;
    E->d.vinput = (const double*)E->vinput;
    E->d.rec_stride = stride;
    E->d.p = p;
    E->d.n_u = n_u;
    E->d.n_roots = n_roots;
    E->d.n_nodes = n_nodes;
    *out = E;
    return EHM_OK;
}

int ehm_explicit_eval_batch(ehm_explicit* E, int64_t n, const double* x, double* u,
                            int32_t* leaf, int32_t* visited, double* kernel_seconds) {
    if (!E || !x || !u || n < 0) return xfail(EHM_E_INVALID, "bad argument");
    if (n == 0) return EHM_OK;
    hipError_t e = hipSetDevice(E->device);
    if (e != hipSuccess) return xfail(EHM_E_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    const int p = E->d.p, n_u = E->d.n_u;
    if ((size_t)n > E->cap) {
        for (void** q : {&E->x, &E->u, &E->leaf, &E->depth, &E->root}) {
            if (*q) (void)hipFree(*q);
            *q = nullptr;
        }
        E->cap = 0;
        if (hipMalloc(&E->x, (size_t)n * p * sizeof(double)) != hipSuccess ||
            hipMalloc(&E->u, (size_t)n * n_u * sizeof(double)) != hipSuccess ||
            hipMalloc(&E->leaf, (size_t)n * sizeof(int32_t)) != hipSuccess ||
            hipMalloc(&E->depth, (size_t)n * sizeof(int32_t)) != hipSuccess ||
            hipMalloc(&E->root, (size_t)n * sizeof(int32_t)) != hipSuccess)
            return xfail(EHM_E_HIP, "out of device memory for %lld queries", (long long)n);
        E->cap = (size_t)n;
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
#define Y_TRY(expr)                                                                        \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return xfail(EHM_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));        \
    } while (0)
    Y_TRY(hipMemcpyAsync(E->x, x, (size_t)n * p * sizeof(double), hipMemcpyHostToDevice,
                         E->stream));
    (void)hipEventRecord(e0, E->stream);
    // long spines: the visibility walk over the roots finds the root first (EHM_EXPLICIT_LOCATE=0:
    // the reference's serial walk for every state)
    static const int locate_off = [] {
        const char* e = getenv("EHM_EXPLICIT_LOCATE");
        return (e && atoi(e) == 0) ? 1 : 0;
    }();
    const bool locate = E->nbr && !locate_off;
    if (locate)
        hipLaunchKernelGGL(k_explicit_locate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           E->stream, E->d, (long long)n, (const double*)E->x,
                           (const int32_t*)E->nbr, (int32_t*)E->root);
    hipLaunchKernelGGL(k_explicit_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       E->stream, E->d, (long long)n, (const double*)E->x, (double*)E->u,
                       (int32_t*)E->leaf, (int32_t*)E->depth,
                       locate ? (const int32_t*)E->root : (const int32_t*)nullptr);
    (void)hipEventRecord(e1, E->stream);
    Y_TRY(hipGetLastError());
    Y_TRY(hipMemcpyAsync(u, E->u, (size_t)n * n_u * sizeof(double), hipMemcpyDeviceToHost,
                         E->stream));
    if (leaf)
        Y_TRY(hipMemcpyAsync(leaf, E->leaf, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost,
                             E->stream));
    if (visited)
        Y_TRY(hipMemcpyAsync(visited, E->depth, (size_t)n * sizeof(int32_t),
                             hipMemcpyDeviceToHost, E->stream));
    Y_TRY(hipStreamSynchronize(E->stream));
    if (kernel_seconds) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *kernel_seconds = ms * 1e-3;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return EHM_OK;
}

}  // extern "C"
