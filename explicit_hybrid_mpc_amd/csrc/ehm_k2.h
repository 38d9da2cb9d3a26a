// Interface between ehm_capi.hip and the per-(NP, SLOTS) instances of ehm_k2.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ehm_dev.h"

// wavefronts per workgroup are chosen at launch (<= EHM_K2_THREADS / 64); the launch bound
// also caps the registers: 768 threads = 3 wavefronts per SIMD = 168 VGPRs
#ifndef EHM_K2_THREADS
#define EHM_K2_THREADS 768
#endif

#define K2_AUG_MIN 32
// persistent frontier kernel: offset (doubles) into the wavefront's LP workspace where a node's
// vertex gradients are staged between its load (or its creation as the kept child) and the
// tangent-plane bound -- behind the bound's own scratch, 2 (p+1)^2 <= 162 doubles, and inside
// every instance's workspace (>= 512 doubles)
#define K2_HOT_GRAD 176
// Near-threshold routing (hard part 1): a node's close / split decision may be
// taken by a shortcut -- sign-only stop of the suboptimality-test LP, tangent-plane bound,
// midpoint witness, inherited negative verdict -- only when it establishes |t*| >= this
// (relative to 1 + |V_0|); every closer call is decided by the LP solved to FULL accuracy
// (tolerances 1e-10), like the CPU oracle's.  ehm_tree_info.near_threshold counts those.
#define EHM_ROUTE_TOL 1e-6
// persistent frontier kernel: queue-slot mark of a node that has been put back once (its midpoint
// optimum was being solved by another wavefront)
#define EHM_REQUEUED 0x40000000

namespace ehm {

// per-wave scratch in front of the LP workspace (doubles): node record, Gauss-Jordan
// tableau (p x 2p), inverse (p x p), parameter (<= 8)
__host__ __device__ inline size_t k2_node_doubles(int p, int n_u) {
    const size_t nrec = ((size_t)rec_doubles(p, n_u) + 7) & ~(size_t)7;
    // the middle block (2 p^2, at least K2_AUG_MIN doubles) also holds the per-wavefront
    // statistics of the persistent frontier kernel
    const size_t aug = (2 * (size_t)p * p > K2_AUG_MIN) ? 2 * (size_t)p * p : K2_AUG_MIN;
    return (nrec + aug + (size_t)p * p + 8 + 1) & ~(size_t)1;
}

// Midpoint-first flow of the persistent frontier kernel: what a node keeps across its
// suboptimality test, parked behind the wavefront's LP workspace -- [0, n_u) the first input of
// the midpoint solve, [n_u, n_u + p) its cost gradient, [n_u + p, n_u + 2p + 2) the witness that
// is handed on to the children (DevTree::wit).
__host__ __device__ inline int k2_stash_doubles(int p, int n_u) {
    return (n_u + 2 * p + 2 + 1) & ~1;
}

// Row layout of an LP with m MPC rows and ne extra rows: MPC row i sits at (lane i%64,
// slot i/64); the extras follow at xbase (directly behind the MPC rows when they fit into
// the same slot, otherwise in a slot of their own).  Only the LAST slot can hold extras.
__host__ __device__ inline int lp_xbase(int m, int ne) {
    if (ne == 0) return m;
    const int r = m & 63;
    return (r != 0 && r + ne <= 64) ? m : ((m + 63) & ~63);
}
__host__ __device__ inline int lp_slots(int m, int ne) {
    return (lp_xbase(m, ne) + ne + 63) >> 6;
}

// Control block of the persistent frontier kernel (device memory, agent-scope atomics).  The
// four words every node touches sit on cache lines of their own: device-scope atomics on one
// line are serviced one after the other, and at >10 M nodes/s that line is the bottleneck.
struct PersistCtl {
    int head;           // next queue slot to pop
    int pad0[31];
    int tail;           // next queue slot to push
    int pad1[31];
    int pending;        // nodes pushed and not yet completed; 0 = the partition is finished
    int pad2[31];
    int n_nodes;        // node pool allocation counter
    int pad3[31];
    int abort;          // 0 ok, 1 node pool exhausted, 3 watchdog
    int max_depth_seen;
    int truncated;      // some open node was left unsplit at max_depth
    int pad4;
    unsigned long long closed;      // closed leaves      } accumulated per wavefront,
    unsigned long long splits;      // expanded nodes     } added once when it leaves
    // sharded launches: the part above the deal depth, grown identically on every rank
    unsigned long long repl_closed, repl_splits, repl_solves;
};

// Sharding of ONE persistent launch over the ranks of a multi-GPU run: every rank starts from
// the roots and grows the tree above `depth` identically (replicated, a few thousand nodes);
// a node created AT that depth is pursued by rank (path code % world) only -- the others flag
// it "owned by another rank" (bit2) and do not queue it.  world <= 1: no dealing.
struct PersistDeal {
    // pop_limit > 0: budgeted launch -- a wavefront that draws queue position >= pop_limit
    // leaves; the queue is consumed in order, so what stays unprocessed is the contiguous
    // slice [pop_limit, tail) (ehm_partition_advance: rebalancing rounds between launches)
    int pop_limit;
    int depth, rank, world;
    int mix;        // 1 (default) = multiplicative hash of the path code; 0 = its low bits, i.e.
                    // the last turns of the path (measured: 12 % imbalance at 8 ranks against 5 %)
    int keep;       // 1 = work first: a wavefront that splits a node goes on with one of the
                    // children itself and queues only the other (option "work_first")
    int check;      // 1 = cross-check of the two LP-free verdicts (option "check_witness"): the
                    // tangent-plane bound is evaluated ALSO for nodes the inherited witness proves
                    // open; a node it would close at the same time (t* > 0 and t* < 0 cannot both
                    // hold) is counted in ehm_tree_info.errors.  Test builds of a run, not the default:
                    // it costs the bound for half a million nodes of the headline tree.
};

// Optional indirection of the batched oracle kernels: the hybrid partition engine
// (ehm_hybrid.h) builds its work lists on the device and reads / writes through them, so a
// sweep needs no host round trip between its stages.  All-null = dense batches.
struct K2Gather {
    const long long* src;   // per instance: offset (doubles) of its input from the base pointer
                            // (theta / R); null = dense (inst * p, inst * (p+1)p)
    const int32_t* dst;     // per instance: index its results are written at; null = inst
    const int32_t* n_dev;   // device-side instance count overriding n_inst (null = n_inst)
    int v_off;              // simplex kinds with src: Vbar = R + v_off (tree records)
    double* grad;           // point solves: dJ/dtheta out, [dst][p]; null = not wanted
    int sign_mode;          // 0 = full accuracy; 1 / 2 = sign-only stop (ehm_ipm2.h), the value
                            // written is then the bound min(|primal|, |dual|) with the sign
    MidTable pt;            // point batches of the multi-commutation engine: results shared by
                            // (parameter, commutation, kind) -- the simplices around an edge ask
                            // for the same problems at its midpoint (state == nullptr: off)
};

struct K2Launch {
    int grid;
    int threads;        // 64 * wavefronts per workgroup
    size_t lds_bytes;
    int wave_doubles;   // LDS doubles per wavefront
    hipStream_t stream;
    int wc_lds;         // DevProblem::wc_lds of this launch (the wrappers copy it into P)
};

// One compiled kernel family: the wave-local instances of ehm_k2.hip (threads_per_lp = 64,
// several LPs per workgroup around one LDS copy of the constant block), the wide instances of
// ehm_k3.hip (threads_per_lp = 256, one LP per workgroup, constant block streamed from L2) or the
// LDS-resident wide family of ehm_k4.hip (threads_per_lp = 512, one LP per workgroup and CU, the
// reduced block of the commutation in LDS).
struct K2Api {
    int np, slots, max_threads, threads_per_lp;
    hipError_t (*set_lds)(int bytes);
    // LDS doubles per LP; persist: with what the midpoint-first persistent kernel parks per node
    size_t (*wave_doubles)(const DevProblem& P, int n_lp, int ne, int persist);
    size_t (*shared_doubles)(const DevProblem& P);
    void (*point)(const K2Launch&, DevProblem, long long n_inst, const double* theta,
                  const int32_t* seg, int feas, double* J, double* u0, int32_t* status,
                  int32_t* iters, DevCounters*, K2Gather);
    void (*simplex)(const K2Launch&, DevProblem, long long n_inst, const double* R,
                    const double* Vbar, const int32_t* seg, int mode, double* obj,
                    double* alpha, int32_t* status, int32_t* iters, DevCounters*, K2Gather);
    void (*decide)(const K2Launch&, DevProblem, DevTree, const int32_t* frontier, int nf,
                   int32_t* open_flag, DevCounters*, int sign_only);
    void (*expand)(const K2Launch&, DevProblem, DevTree, const int32_t* open_list, int n_open,
                   int child_base, int32_t* next_frontier, DevCounters*);
    void (*vertex)(const K2Launch&, DevProblem, DevTree, const int32_t* nodes, int n_nodes,
                   DevCounters*);
    void (*selftest)(hipStream_t, double* out);
    // persistent frontier kernel (ehm_k2.hip: k2_persist); null where not compiled (wide)
    void (*persist)(const K2Launch&, DevProblem, DevTree, int32_t* slots, int n_slots,
                    PersistCtl* ctl, int node_cap, DevCounters*, int sign_only, int max_depth,
                    PersistDeal);
    // families that cannot take every problem of their (np, slots) class say which they take
    // (ehm_k4.hip: the reduced block must fit in LDS); null = all of them
    int (*fits)(const DevProblem& P, int lds_budget_bytes);
};

// The persistent frontier kernel compiled at two solver widths (ehm_kp.hip).
struct KpApi {
    int np_decide, np_expand, slots, max_threads;
    hipError_t (*set_lds)(int bytes);
    size_t (*wave_doubles)(const DevProblem& P, int n_lp_decide, int ne_decide, int n_lp_expand);
    size_t (*shared_doubles)(const DevProblem& P);
    void (*persist)(const K2Launch&, DevProblem, DevTree, int32_t* slots, int n_slots,
                    PersistCtl* ctl, int node_cap, DevCounters*, int sign_only, int max_depth,
                    PersistDeal);
};

}  // namespace ehm
