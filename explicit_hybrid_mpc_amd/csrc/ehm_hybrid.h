// Partition engine for multi-commutation (hybrid) problems: Worker.ecc / Worker.lcss of the
// reference (lib/worker.py:241-417) with its mixed-integer oracles V_R / bar_E_delta_R /
// bar_D_delta_R / in_variability_ball (lib/oracle.py:175-414) evaluated for a whole frontier at
// a time, entirely on the device.  Included by ehm_capi.hip (one translation unit).
//
// A sweep is a fixed sequence of launches with no host round trip in between: small list
// kernels turn per-node bit masks ("which commutations still need which LP") into work lists
// sorted by commutation, the batched oracle kernels of ehm_k2.hip / ehm_k3.hip read their
// instances straight from the node records through those lists (K2Gather) and write their
// results at (node, commutation) addresses, decision kernels consume them.  The host reads one
// control block per sweep (sizes of the next frontiers, node count, error word).
//
// What a node carries besides its record (DevTree):
//   vf   [node][p+1][nw]  bit d of row v: commutation d is feasible at vertex v (phase-one LP,
//                         lib/oracle.py:141-173 with check_feasibility) -- children inherit p of
//                         their p+1 rows, the midpoint's row is inferred where convexity decides
//                         it (feasible at both ends of the split edge => feasible; infeasible on
//                         the whole parent simplex => infeasible) and solved for the rest;
//   cand [node][nw]       commutations that were feasible somewhere on the PARENT simplex;
//   neg  [node][nw]       commutations whose suboptimality-test optimum t* is known to be negative
//                         on this simplex WITHOUT solving anything: t*_d can only go down from a
//                         node to a child that was split with the node's own commutation (the
//                         child's interpolated cost lies below the parent's on the smaller simplex
//                         -- the optimal cost is convex -- and the maximisation runs over a
//                         subset), so a commutation found negative stays negative until the
//                         node's commutation changes.  tneg[node] = the largest (closest to 0) of
//                         those inherited values, so that the recorded margin stays a lower bound;
//   black[node][nw]       commutations blacklisted for this node's bar_D after a failed vertex
//                         solve (the reference's __delta_neq_constraint retry, lib/oracle.py:406-414).
// Canonical commutation rule and tolerances: DESIGN.md section 3 ("canonical commutation rule");
// per-node semantics are exactly those of ehm_lcss_batch + partition.grow_hybrid of round 1.

typedef unsigned long long hy_u64;

struct HyCtr {
    int n_nodes;            // node pool allocation counter
    int next_ecc;           // next-frontier cursors
    int next_lcss;
    int n_split;            // split list of the current chunk
    int n_items;            // length of the work list being built
    int error;              // 0 ok, 1 node pool exhausted, 2 numeric, 3 infeasible Theta
    int err_node;
    int truncated;
    int max_depth_seen;
    int err_kind;           // numeric errors: 1 slack problem, 2 min over the simplex, 3 midpoint
    int dealt;              // sharded runs: the deal depth has been reached (children dealt)
    int pad2;
    unsigned long long closed, splits, swaps, slivers, blacklisted, ref_solves, fallbacks, routed;
    unsigned long long min_margin_bits;
};

#define HY_BLOCK 256

__device__ __forceinline__ bool hy_bit(const hy_u64* row, int d) {
    return (row[d >> 6] >> (d & 63)) & 1ULL;
}
__device__ __forceinline__ hy_u64 hy_valid_word(int nd, int w) {
    const int lo = w * 64;
    if (nd >= lo + 64) return ~0ULL;
    if (nd <= lo) return 0ULL;
    return (1ULL << (nd - lo)) - 1ULL;
}

// ---- work lists sorted by commutation ----------------------------------------------------------
// bits[k][nw]: row k wants an instance for every set commutation bit.  Threads are laid out
// commutation-major (t = d * ns_pad + k), so a wavefront works on ONE commutation and adds to its
// counter once.
__global__ void hy_list_count(const hy_u64* __restrict__ bits, int ns, int nd, int nw,
                              int* __restrict__ count) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ns_pad = (ns + 63) & ~63;
    const int d = (int)(t / ns_pad), k = (int)(t % ns_pad);
    if (d >= nd) return;
    const bool has = (k < ns) && hy_bit(bits + (size_t)k * nw, d);
    const hy_u64 b = __ballot(has);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&count[d], __popcll(b));
}
// seg[0..nd] = exclusive prefix of count, cursor = seg, n_items = total; count is cleared
__global__ void hy_list_scan(int* count, int nd, int* seg, int* cursor, int* n_items) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int acc = 0;
    for (int d = 0; d < nd; ++d) {
        seg[d] = acc;
        cursor[d] = acc;
        acc += count[d];
        count[d] = 0;
    }
    seg[nd] = acc;
    *n_items = acc;
}
// instance of (row k, commutation d): input at koff[k], results at k * nd + d
__global__ void hy_list_scatter(const hy_u64* __restrict__ bits, int ns, int nd, int nw,
                                const long long* __restrict__ koff, int* __restrict__ cursor,
                                long long* __restrict__ src, int32_t* __restrict__ dst,
                                int32_t* __restrict__ dcomm) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ns_pad = (ns + 63) & ~63;
    const int d = (int)(t / ns_pad), k = (int)(t % ns_pad);
    if (d >= nd) return;
    const bool has = (k < ns) && hy_bit(bits + (size_t)k * nw, d);
    const hy_u64 b = __ballot(has);
    if (!b) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(&cursor[d], __popcll(b));
    base = __shfl(base, 0, 64);
    if (has) {
        const int idx = base + __popcll(b & ((1ULL << lane) - 1ULL));
        src[idx] = koff[k];
        dst[idx] = k * nd + d;
        dcomm[idx] = d;
    }
}
// entries with one commutation each: dsel[e] (-1 = no instance), input at eoff[e], results at e
__global__ void hy_sel_count(const int32_t* __restrict__ dsel, int ne, int* __restrict__ count) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int d = dsel[e];
    if (d >= 0) atomicAdd(&count[d], 1);
}
__global__ void hy_sel_scatter(const int32_t* __restrict__ dsel, int ne,
                               const long long* __restrict__ eoff, int* __restrict__ cursor,
                               long long* __restrict__ src, int32_t* __restrict__ dst,
                               int32_t* __restrict__ dcomm) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int d = dsel[e];
    if (d < 0) return;
    const int idx = atomicAdd(&cursor[d], 1);
    src[idx] = eoff[e];
    dst[idx] = e;
    dcomm[idx] = d;
}
// entries whose solve stalled: again, for the one-wavefront kernels
__global__ void hy_sel_stalled(const int32_t* __restrict__ dsel, int ne,
                               const int32_t* __restrict__ st, int32_t* __restrict__ dsel2) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    dsel2[e] = (dsel[e] >= 0 && st[e] != 0) ? dsel[e] : -1;
}
__global__ void hy_add_items(HyCtr* ctr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctr->fallbacks += (unsigned long long)ctr->n_items;
}

// out[k] = forced[k] | (ask[k] & {tau[k][d] <= tol})
__global__ void hy_bits_finish(int ns, int nd, int nw, const hy_u64* __restrict__ forced,
                               const hy_u64* __restrict__ ask, const double* __restrict__ tau,
                               double tol, hy_u64* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ns * nw) return;
    const int k = t / nw, w = t - k * nw;
    hy_u64 f = forced ? forced[t] : 0ULL;
    hy_u64 a = ask[t];
    while (a) {
        const int b = __ffsll((long long)a) - 1;
        a &= a - 1;
        if (tau[(size_t)k * nd + w * 64 + b] <= tol) f |= 1ULL << b;
    }
    out[t] = f;
}

// ---- roots: every vertex row asks for every commutation ---------------------------------------
__global__ void hy_root_rows(int row0, int nrows, int nd, int nw, int nv, int p, int stride,
                             hy_u64* __restrict__ ask, long long* __restrict__ koff) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows * nw) return;
    const int k = t / nw, w = t - k * nw;
    ask[t] = hy_valid_word(nd, w);
    if (w == 0) {
        const int row = row0 + k;
        koff[k] = (long long)(row / nv) * stride + (long long)(row % nv) * p;
    }
}
__global__ void hy_node_init(int first, int n, int nd, int nw, hy_u64* __restrict__ cand,
                             hy_u64* __restrict__ black, hy_u64* __restrict__ neg,
                             double* __restrict__ tneg) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nw) return;
    const int k = t / nw, w = t - k * nw;
    cand[(size_t)(first + k) * nw + w] = hy_valid_word(nd, w);
    black[(size_t)(first + k) * nw + w] = 0ULL;
    neg[(size_t)(first + k) * nw + w] = 0ULL;
    if (w == 0) tneg[first + k] = -INFINITY;
}

// ---- lcss: which (node, commutation) pairs need which problem ---------------------------------
// known feasible somewhere (feasible at a vertex) -> slack problem directly; infeasible on the
// parent simplex -> nothing; the rest -> phase one over the simplex first.
__global__ void hy_lcss_classify(DevTree T, const int32_t* __restrict__ frontier, int ns, int nd,
                                 int nw, const hy_u64* __restrict__ vf,
                                 const hy_u64* __restrict__ cand, const hy_u64* __restrict__ neg,
                                 hy_u64* __restrict__ known1, hy_u64* __restrict__ ask,
                                 hy_u64* __restrict__ vall, long long* __restrict__ koff) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ns * nw) return;
    const int k = t / nw, w = t - k * nw;
    const int id = frontier[k];
    const int nv = T.p + 1;
    hy_u64 any = 0ULL, all = ~0ULL;
    for (int v = 0; v < nv; ++v) {
        const hy_u64 x = vf[((size_t)id * nv + v) * nw + w];
        any |= x;
        all &= x;
    }
    const hy_u64 valid = hy_valid_word(nd, w) & ~neg[(size_t)id * nw + w];
    known1[t] = any & valid;
    ask[t] = ~any & cand[(size_t)id * nw + w] & valid;
    vall[t] = all & valid;
    if (w == 0) koff[k] = (long long)id * T.rec_stride;
}
// after phase one: pairs that go to the slack problem; tau of the known-feasible ones := -1
__global__ void hy_after_phase1(int ns, int nd, int nw, const hy_u64* __restrict__ known1,
                                const hy_u64* __restrict__ ask, double* __restrict__ tau,
                                double tol, hy_u64* __restrict__ slk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ns * nw) return;
    const int k = t / nw, w = t - k * nw;
    hy_u64 f = known1[t], a = ask[t];
    hy_u64 q = f;
    while (q) {
        const int b = __ffsll((long long)q) - 1;
        q &= q - 1;
        tau[(size_t)k * nd + w * 64 + b] = -1.0;
    }
    while (a) {
        const int b = __ffsll((long long)a) - 1;
        a &= a - 1;
        if (tau[(size_t)k * nd + w * 64 + b] <= tol) f |= 1ULL << b;
    }
    slk[t] = f;
}
// pairs taken as feasible on a vertex's word whose slack problem stalled: phase one after all
__global__ void hy_redo_bits(int ns, int nd, int nw, const hy_u64* __restrict__ known1,
                             const int32_t* __restrict__ st, hy_u64* __restrict__ redo) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ns * nw) return;
    const int k = t / nw, w = t - k * nw;
    hy_u64 q = known1[t], r = 0ULL;
    while (q) {
        const int b = __ffsll((long long)q) - 1;
        q &= q - 1;
        if (st[(size_t)k * nd + w * 64 + b] != 0) r |= 1ULL << b;
    }
    redo[t] = r;
}

// stalled slack problems of commutations that ARE feasible with an interior (not slivers): once
// more on the one-wavefront kernels (private copy of the LP, barycentric coordinates)
__global__ void hy_retry_bits(int ns, int nd, int nw, const hy_u64* __restrict__ slk,
                              const int32_t* __restrict__ st, const double* __restrict__ tau,
                              double sliver_tol, hy_u64* __restrict__ retry) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ns * nw) return;
    const int k = t / nw, w = t - k * nw;
    hy_u64 q = slk[t], r = 0ULL;
    while (q) {
        const int b = __ffsll((long long)q) - 1;
        q &= q - 1;
        const size_t i = (size_t)k * nd + w * 64 + b;
        if (st[i] != 0 && !(tau[i] > -sliver_tol)) r |= 1ULL << b;
    }
    retry[t] = r;
}

// "Any admissible commutation" (option any_admissible = seed + 1): the reference's V_R and bar_D
// are FEASIBILITY problems -- Minimize(0), lib/oracle.py:201, 347 -- and which admissible
// commutation comes back is its solver's choice.  The canonical rule below (first feasible / the
// largest slack) is one such choice; this one draws uniformly by a hash of (seed, oracle, path
// code of the node), so that a run can be repeated bit for bit and the CPU oracle
// (oracle/oracle_cpu.py, rule 'hash') takes the same draws.  salt: 1 = V_R, 2 = bar_D.
// draw >= HY_RULE_BASE: deterministic extremes instead of a draw, for the envelope of what the
// choice can do to a tree -- bit 0: V_R returns the LAST commutation feasible at every vertex
// (canonical: the first); bit 1: bar_D returns the admissible commutation with the SMALLEST slack
// t* >= 0 (canonical: the largest).
#define HY_RULE_BASE 0x40000000u
__host__ __device__ inline uint32_t hy_draw(uint32_t draw, uint32_t salt, uint32_t code) {
    unsigned long long z = ((unsigned long long)((draw - 1u) * 4u + salt) << 32) | code;
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}

// One Worker.lcss decision per node (lib/worker.py:368-401 with the oracles of
// lib/oracle.py:285-394): bar_E from the slacks of every commutation, then bar_D's choice.
//   act: 0 closed, 1 split with the node's own commutation, 2 better commutation found (its
//        vertex solves / in_variability_ball follow), 3 open but left at max_depth
__global__ void hy_lcss_decide(DevTree T, const int32_t* __restrict__ frontier, int ns, int nd,
                               int nw, const hy_u64* __restrict__ slk,
                               const hy_u64* __restrict__ vall, const hy_u64* __restrict__ black,
                               const double* __restrict__ tau, const double* __restrict__ tval,
                               const int32_t* __restrict__ st, const double* __restrict__ alpha,
                               double sliver_tol, double tie_tol, int max_depth, int prune,
                               hy_u64* __restrict__ neg, double* __restrict__ tneg,
                               hy_u64* __restrict__ cand, int32_t* __restrict__ act,
                               int32_t* __restrict__ best_out, double* __restrict__ ths,
                               HyCtr* ctr, uint32_t draw) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const int id = frontier[k];
    const int p = T.p, nv = p + 1;
    // commutations inherited as negative keep their last value (a bound of the child's)
    double tb = tneg[id], tbm = -INFINITY, tn = tneg[id];
    hy_u64 feas[4] = {0ULL, 0ULL, 0ULL, 0ULL};
    hy_u64 ng[4] = {0ULL, 0ULL, 0ULL, 0ULL};
    const double neg_tol = EHM_ROUTE_TOL * (1.0 + fabs(T.rec[(size_t)id * T.rec_stride +
                                                           rec_off_vcost(p)]));
    for (int w = 0; w < nw; ++w) {
        ng[w] = neg[(size_t)id * nw + w];
        feas[w] = ng[w] & cand[(size_t)id * nw + w];     // still "feasible somewhere" as far as known
    }
    for (int d = 0; d < nd; ++d) {
        if (!hy_bit(slk + (size_t)k * nw, d)) continue;
        const size_t q = (size_t)k * nd + d;
        if (st[q] != 0) {
            if (tau[q] > -sliver_tol) {      // feasible only within the phase-one accuracy:
                atomicAdd(&ctr->slivers, 1ULL);                   // no interior, dropped
            } else {
                atomicMax(&ctr->error, 2);
                ctr->err_node = id;
                ctr->err_kind = 1 + 16 * d;
                T.flags[id] |= 8;
            }
            continue;
        }
        feas[d >> 6] |= 1ULL << (d & 63);
        const double t = tval[q];
        tb = fmax(tb, t);
        if (prune && t < -neg_tol) {
            ng[d >> 6] |= 1ULL << (d & 63);
            tn = fmax(tn, t);
        }
        if (t >= 0.0 && hy_bit(vall + (size_t)k * nw, d) && !hy_bit(black + (size_t)id * nw, d))
            tbm = fmax(tbm, t);
    }
    for (int w = 0; w < nw; ++w) {
        cand[(size_t)id * nw + w] = feas[w];
        neg[(size_t)id * nw + w] = ng[w];
    }
    tneg[id] = tn;
    if (fabs(tb) < neg_tol) atomicAdd(&ctr->routed, 1ULL);
    T.tstar[id] = tb;
    atomicMin(&ctr->min_margin_bits, (unsigned long long)__double_as_longlong(fabs(tb)));
    atomicAdd(&ctr->ref_solves, 1ULL);
    const int dep = T.depth[id];
    atomicMax(&ctr->max_depth_seen, dep);
    if (!(tb >= 0.0)) {
        T.flags[id] |= 1;
        atomicAdd(&ctr->closed, 1ULL);
        act[k] = 0;
        return;
    }
    if (max_depth > 0 && dep >= max_depth) {
        ctr->truncated = 1;
        act[k] = 3;
        return;
    }
    atomicAdd(&ctr->ref_solves, 1ULL);
    int best = -1;
    if (tbm >= 0.0 && draw) {
        // any admissible commutation: the r-th of those with t* >= 0 (enumeration order)
        int cnt = 0;
        for (int d = 0; d < nd; ++d) {
            if (!((feas[d >> 6] >> (d & 63)) & 1ULL)) continue;
            if (!hy_bit(vall + (size_t)k * nw, d) || hy_bit(black + (size_t)id * nw, d)) continue;
            if (tval[(size_t)k * nd + d] >= 0.0) ++cnt;
        }
        if (draw >= HY_RULE_BASE && !(draw & 2u)) {
            cnt = 0;            // bar_D canonical (bit 1 clear): fall through to the rule below
        } else if (draw >= HY_RULE_BASE) {
            double tmin = INFINITY;         // smallest admissible slack, first in enumeration order
            for (int d = 0; d < nd; ++d) {
                if (!((feas[d >> 6] >> (d & 63)) & 1ULL)) continue;
                if (!hy_bit(vall + (size_t)k * nw, d) || hy_bit(black + (size_t)id * nw, d)) continue;
                const double t = tval[(size_t)k * nd + d];
                if (t >= 0.0 && t < tmin) {
                    tmin = t;
                    best = d;
                }
            }
        } else {
            int r = (int)(hy_draw(draw, 2u, T.code[id]) % (uint32_t)cnt);
            for (int d = 0; d < nd && best < 0; ++d) {
                if (!((feas[d >> 6] >> (d & 63)) & 1ULL)) continue;
                if (!hy_bit(vall + (size_t)k * nw, d) || hy_bit(black + (size_t)id * nw, d)) continue;
                if (tval[(size_t)k * nd + d] >= 0.0 && r-- == 0) best = d;
            }
        }
        if (cnt == 0) {
            const double thr = tbm - tie_tol * (1.0 + fabs(tbm));
            for (int d = 0; d < nd && best < 0; ++d) {
                if (!((feas[d >> 6] >> (d & 63)) & 1ULL)) continue;
                if (!hy_bit(vall + (size_t)k * nw, d) || hy_bit(black + (size_t)id * nw, d)) continue;
                const double t = tval[(size_t)k * nd + d];
                if (t >= 0.0 && t >= thr) best = d;
            }
        }
    } else if (tbm >= 0.0) {
        const double thr = tbm - tie_tol * (1.0 + fabs(tbm));
        for (int d = 0; d < nd && best < 0; ++d) {
            if (!((feas[d >> 6] >> (d & 63)) & 1ULL)) continue;
            if (!hy_bit(vall + (size_t)k * nw, d) || hy_bit(black + (size_t)id * nw, d)) continue;
            const double t = tval[(size_t)k * nd + d];
            if (t >= 0.0 && t >= thr) best = d;
        }
    }
    if (best >= 0 && best == T.didx[id]) best = -1;          // lib/oracle.py:384-394
    best_out[k] = best;
    if (best < 0) {
        act[k] = 1;
        return;
    }
    act[k] = 2;
    atomicAdd(&ctr->ref_solves, (unsigned long long)(p + 3));
    const double* al = alpha + ((size_t)k * nd + best) * nv;
    const double* R = T.rec + (size_t)id * T.rec_stride;
    for (int c = 0; c < p; ++c) {
        double acc = 0.0;
        for (int v = 0; v < nv; ++v) acc += al[v] * R[v * p + c];
        ths[(size_t)k * p + c] = acc;
    }
}

// entries of the follow-up solves of the nodes with act == 2 (or, ecc: a commutation to adopt)
//   V: P_theta_delta at every vertex with the new commutation   (lib/oracle.py:399-401)
//   T: P_theta_delta at theta* with the new commutation           (lib/oracle.py:281)
//   M: min over the simplex with the node's own commutation       (lib/oracle.py:276)
__global__ void hy_delta_entries(DevTree T, const int32_t* __restrict__ frontier, int ns,
                                 const int32_t* __restrict__ act, const int32_t* __restrict__ best,
                                 int32_t* __restrict__ dselV, long long* __restrict__ eoffV,
                                 int32_t* __restrict__ dselT, long long* __restrict__ eoffT,
                                 int32_t* __restrict__ dselM, long long* __restrict__ eoffM) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const int id = frontier[k];
    const int p = T.p, nv = p + 1;
    const bool on = act[k] == 2;
    for (int v = 0; v < nv; ++v) {
        dselV[k * nv + v] = on ? best[k] : -1;
        eoffV[k * nv + v] = (long long)id * T.rec_stride + (long long)v * p;
    }
    if (dselT) {
        dselT[k] = on ? best[k] : -1;
        eoffT[k] = (long long)k * p;
        dselM[k] = on ? T.didx[id] : -1;
        eoffM[k] = (long long)id * T.rec_stride;
    }
}

// in_variability_ball (lib/oracle.py:220-283) and what follows from it (lib/worker.py:396-401):
//   act 4 = the node takes the better commutation in place and is visited again,
//   act 5 = split with the better commutation's data;
// a failed solve blacklists the commutation for this node and the node is visited again
// (lib/oracle.py:406-414).
__global__ void hy_varsmall(DevTree T, const int32_t* __restrict__ frontier, int ns, int nw,
                            int32_t* __restrict__ act, const int32_t* __restrict__ best,
                            const double* __restrict__ vJ, const double* __restrict__ vu,
                            const int32_t* __restrict__ vst, const double* __restrict__ Jth,
                            const int32_t* __restrict__ Jth_st, const double* __restrict__ Jmin,
                            const int32_t* __restrict__ Jmin_st, double eps_a, double eps_r,
                            int fail_delta, hy_u64* __restrict__ black, hy_u64* __restrict__ neg,
                            double* __restrict__ tneg, HyCtr* ctr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns || act[k] != 2) return;
    const int id = frontier[k];
    const int p = T.p, nv = p + 1, n_u = T.n_u;
    bool ok = (Jth_st[k] == 0) && (best[k] != fail_delta);
    for (int v = 0; v < nv; ++v) ok = ok && (vst[k * nv + v] == 0);
    if (!ok) {
        black[(size_t)id * nw + (best[k] >> 6)] |= 1ULL << (best[k] & 63);
        atomicAdd(&ctr->blacklisted, 1ULL);
        act[k] = 6;                                   // visit again, nothing changed
        return;
    }
    if (Jmin_st[k] != 0) {                            // the node's OWN commutation failed
        atomicMax(&ctr->error, 2);
        ctr->err_node = id;
        ctr->err_kind = 2 + 16 * T.didx[id];
        return;
    }
    double* rec = T.rec + (size_t)id * T.rec_stride;
    const double* V = rec + rec_off_vcost(p);
    double vmax = -INFINITY;
    for (int v = 0; v < nv; ++v) vmax = fmax(vmax, V[v]);
    const double rhs = fmax(eps_a, eps_r * Jth[k]);
    if (vmax - Jmin[k] < rhs) {
        for (int v = 0; v < nv; ++v) rec[rec_off_vcost(p) + v] = vJ[k * nv + v];
        for (int q = 0; q < nv * n_u; ++q) rec[rec_off_vinput(p) + q] = vu[(size_t)k * nv * n_u + q];
        T.didx[id] = best[k];
        // a new interpolated cost: nothing is known about the other commutations any more
        for (int w = 0; w < nw; ++w) neg[(size_t)id * nw + w] = 0ULL;
        tneg[id] = -INFINITY;
        atomicAdd(&ctr->swaps, 1ULL);
        act[k] = 4;
    } else {
        act[k] = 5;
    }
}

// nodes that stay in the frontier (act 4 / 6)
__global__ void hy_revisit(const int32_t* __restrict__ frontier, int ns,
                           const int32_t* __restrict__ act, int32_t* __restrict__ next_lcss,
                           HyCtr* ctr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    if (act[k] == 4 || act[k] == 6) next_lcss[atomicAdd(&ctr->next_lcss, 1)] = frontier[k];
}

// ---- splits (lib/worker.py:403-414, 269-278; tools.split_along_longest_edge) -------------------
// has_data = 1: lcss nodes with act 1 / 5; 0: ecc nodes with act 1.
__global__ void hy_split_collect(DevTree T, const int32_t* __restrict__ frontier, int ns, int nd,
                                 int nw, int has_data, int node_cap,
                                 const int32_t* __restrict__ act, const int32_t* __restrict__ best,
                                 const hy_u64* __restrict__ vf, const hy_u64* __restrict__ cand,
                                 int32_t* __restrict__ sp_k, int32_t* __restrict__ sp_c0,
                                 int32_t* __restrict__ sp_ij, int32_t* __restrict__ sp_d,
                                 double* __restrict__ mids, hy_u64* __restrict__ midforced,
                                 hy_u64* __restrict__ midask, long long* __restrict__ moff,
                                 HyCtr* ctr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const int a = act[k];
    if (!(a == 1 || (has_data && a == 5))) return;
    const int id = frontier[k];
    const int p = T.p, nv = p + 1;
    const int c0 = atomicAdd(&ctr->n_nodes, 2);
    if (c0 + 2 > node_cap) {
        atomicMax(&ctr->error, 1);
        return;
    }
    const int s = atomicAdd(&ctr->n_split, 1);
    const double* R = T.rec + (size_t)id * T.rec_stride;
    int bi, bj;
    longest_edge(R, p, bi, bj);
    {
#pragma clang fp contract(off)
        for (int c = 0; c < p; ++c) mids[(size_t)s * p + c] = (R[bi * p + c] + R[bj * p + c]) / 2.0;
    }
    sp_k[s] = k;
    sp_c0[s] = c0;
    sp_ij[2 * s] = bi;
    sp_ij[2 * s + 1] = bj;
    sp_d[s] = has_data ? ((a == 5) ? best[k] : T.didx[id]) : -1;
    moff[s] = (long long)s * p;
    for (int w = 0; w < nw; ++w) {
        const hy_u64 f = vf[((size_t)id * nv + bi) * nw + w] & vf[((size_t)id * nv + bj) * nw + w];
        // ecc nodes have not been through the suboptimality test: nothing known about the simplex
        const hy_u64 c = has_data ? cand[(size_t)id * nw + w] : ~0ULL;
        midforced[(size_t)s * nw + w] = f;
        midask[(size_t)s * nw + w] = ~f & c & hy_valid_word(nd, w);
    }
    atomicAdd(&ctr->splits, 1ULL);
    if (has_data) atomicAdd(&ctr->ref_solves, 1ULL);
}

// one wavefront per split: the two child records, their feasibility rows, structure words
__global__ void hy_children(DevTree T, const int32_t* __restrict__ frontier, int nd, int nw,
                            int has_data, const int32_t* __restrict__ act,
                            const int32_t* __restrict__ sp_k, const int32_t* __restrict__ sp_c0,
                            const int32_t* __restrict__ sp_ij, const int32_t* __restrict__ sp_d,
                            const double* __restrict__ mids, const double* __restrict__ Jm,
                            const double* __restrict__ um, const int32_t* __restrict__ mst,
                            const double* __restrict__ vJ, const double* __restrict__ vu,
                            const hy_u64* __restrict__ midbits, hy_u64* __restrict__ vf,
                            hy_u64* __restrict__ cand, hy_u64* __restrict__ black,
                            hy_u64* __restrict__ neg, double* __restrict__ tneg,
                            int32_t* __restrict__ next, HyCtr* ctr, PersistDeal deal) {
    const int s = blockIdx.x;
    if (s >= ctr->n_split) return;
    const int lane = threadIdx.x;
    const int k = sp_k[s];
    const int id = frontier[k];
    const int p = T.p, nv = p + 1, n_u = T.n_u;
    const int c0 = sp_c0[s], bi = sp_ij[2 * s], bj = sp_ij[2 * s + 1];
    if (has_data && mst[s] != 0) {
        if (lane == 0) {
            atomicMax(&ctr->error, 2);
            ctr->err_node = id;
            ctr->err_kind = 3 + 16 * sp_d[s];
            T.flags[id] |= 16;
        }
        return;
    }
    const double* rec = T.rec + (size_t)id * T.rec_stride;
    double* rec0 = T.rec + (size_t)c0 * T.rec_stride;
    double* rec1 = rec0 + T.rec_stride;
    const int ov = rec_off_vcost(p), ou = rec_off_vinput(p), nrec = rec_doubles(p, n_u);
    // the children carry the data of the commutation the split was made with: the node's own, or
    // the better one's vertex solves (the parent keeps its own record, lib/worker.py:356-365)
    const bool fresh = has_data && act[k] == 5;
    const double* mid = mids + (size_t)s * p;
    for (int q = lane; q < nrec; q += 64) {
        double v0, v1;
        if (q < ov) {
            v0 = v1 = rec[q];
            if (q >= bi * p && q < bi * p + p) v0 = mid[q - bi * p];
            if (q >= bj * p && q < bj * p + p) v1 = mid[q - bj * p];
        } else if (!has_data) {
            v0 = v1 = 0.0;
        } else if (q < ou) {
            v0 = v1 = fresh ? vJ[(size_t)k * nv + (q - ov)] : rec[q];
            if (q - ov == bi) v0 = Jm[s];
            if (q - ov == bj) v1 = Jm[s];
        } else {
            const int r = q - ou;
            v0 = v1 = fresh ? vu[(size_t)k * nv * n_u + r] : rec[q];
            if (r >= bi * n_u && r < bi * n_u + n_u) v0 = um[(size_t)s * n_u + r - bi * n_u];
            if (r >= bj * n_u && r < bj * n_u + n_u) v1 = um[(size_t)s * n_u + r - bj * n_u];
        }
        rec0[q] = v0;
        rec1[q] = v1;
    }
    for (int q = lane; q < nv * nw; q += 64) {
        const int v = q / nw, w = q - v * nw;
        const hy_u64 x = vf[((size_t)id * nv + v) * nw + w];
        const hy_u64 m = midbits[(size_t)s * nw + w];
        vf[((size_t)c0 * nv + v) * nw + w] = (v == bi) ? m : x;
        vf[((size_t)(c0 + 1) * nv + v) * nw + w] = (v == bj) ? m : x;
    }
    for (int w = lane; w < nw; w += 64) {
        const hy_u64 c = has_data ? cand[(size_t)id * nw + w] : hy_valid_word(nd, w);
        cand[(size_t)c0 * nw + w] = c;
        cand[(size_t)(c0 + 1) * nw + w] = c;
        black[(size_t)c0 * nw + w] = 0ULL;
        black[(size_t)(c0 + 1) * nw + w] = 0ULL;
        // split with the node's OWN commutation: what was negative stays negative
        const hy_u64 g = (has_data && !fresh) ? neg[(size_t)id * nw + w] : 0ULL;
        neg[(size_t)c0 * nw + w] = g;
        neg[(size_t)(c0 + 1) * nw + w] = g;
    }
    if (lane == 0) {
        const double tn = (has_data && !fresh) ? tneg[id] : -INFINITY;
        tneg[c0] = tn;
        tneg[c0 + 1] = tn;
    }
    if (lane == 0) {
        const int dep = T.depth[id] + 1;
        T.left[id] = c0;
        // sharded runs (PersistDeal, ehm_k2.h): the children created at the deal depth are kept by
        // the rank a hash of their path code names, flagged "another rank's" (bit2) elsewhere
        int own[2] = {1, 1};
        if (T.code) {
            const uint32_t pc = T.code[id];
            for (int b = 0; b < 2; ++b) {
                const uint32_t code = 2u * pc + (uint32_t)b;
                T.code[c0 + b] = code;
                if (deal.world > 1 && dep == deal.depth) {
                    const uint32_t h = deal.mix ? ((code * 2654435761u) >> 12) : code;
                    own[b] = (int)(h % (uint32_t)deal.world) == deal.rank;
                    ctr->dealt = 1;
                }
            }
        }
        for (int b = 0; b < 2; ++b) {
            const int c = c0 + b;
            T.left[c] = -1;
            T.didx[c] = sp_d[s];
            T.depth[c] = dep;
            T.flags[c] = (uint8_t)((has_data ? 2 : 0) | (own[b] ? 0 : 4));
            T.tstar[c] = 0.0;
        }
        const int nown = own[0] + own[1];
        if (nown) {
            int at = atomicAdd(has_data ? &ctr->next_lcss : &ctr->next_ecc, nown);
            if (own[0]) next[at++] = c0;
            if (own[1]) next[at] = c0 + 1;
        }
    }
}

// ---- ecc (lib/worker.py:262-291) ----------------------------------------------------------------
// V_R's choice: first commutation feasible at every vertex (act 2: adopt it after its vertex
// solves), otherwise split (act 1) -- those nodes get the barycentre check of lib/worker.py:264-266
// (a commutation feasible at every vertex is feasible at the barycentre, nothing to check there).
__global__ void hy_ecc_classify(DevTree T, const int32_t* __restrict__ frontier, int ns, int nd,
                                int nw, const hy_u64* __restrict__ vf,
                                const hy_u64* __restrict__ black, int32_t* __restrict__ act,
                                int32_t* __restrict__ best, double* __restrict__ ctrs,
                                hy_u64* __restrict__ cask, long long* __restrict__ coff,
                                HyCtr* ctr, uint32_t draw) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const int id = frontier[k];
    const int p = T.p, nv = p + 1;
    int found = -1;
    if (draw) {
        // any admissible commutation: the r-th of those feasible at every vertex
        hy_u64 allw[4] = {0ULL, 0ULL, 0ULL, 0ULL};
        int cnt = 0;
        for (int w = 0; w < nw; ++w) {
            hy_u64 all = hy_valid_word(nd, w) & ~black[(size_t)id * nw + w];
            for (int v = 0; v < nv; ++v) all &= vf[((size_t)id * nv + v) * nw + w];
            allw[w] = all;
            cnt += __popcll(all);
        }
        if (cnt > 0) {
            int r = (draw >= HY_RULE_BASE) ? ((draw & 1u) ? cnt - 1 : 0)
                                           : (int)(hy_draw(draw, 1u, T.code[id]) % (uint32_t)cnt);
            for (int w = 0; w < nw && found < 0; ++w) {
                hy_u64 all = allw[w];
                const int c = __popcll(all);
                if (r >= c) {
                    r -= c;
                    continue;
                }
                while (r-- > 0) all &= all - 1ULL;          // drop the r lowest set bits
                found = w * 64 + __ffsll((long long)all) - 1;
            }
        }
    } else
    for (int w = 0; w < nw && found < 0; ++w) {
        hy_u64 all = hy_valid_word(nd, w) & ~black[(size_t)id * nw + w];
        for (int v = 0; v < nv; ++v) all &= vf[((size_t)id * nv + v) * nw + w];
        if (all) found = w * 64 + __ffsll((long long)all) - 1;
    }
    best[k] = found;
    act[k] = (found >= 0) ? 2 : 1;
    atomicAdd(&ctr->ref_solves, (unsigned long long)(2 + (found >= 0 ? nv : 0)));
    atomicMax(&ctr->max_depth_seen, T.depth[id]);
    const double* R = T.rec + (size_t)id * T.rec_stride;
    for (int c = 0; c < p; ++c) {
        double acc = 0.0;
        for (int v = 0; v < nv; ++v) acc += R[v * p + c];
        ctrs[(size_t)k * p + c] = acc / nv;
    }
    coff[k] = (long long)k * p;
    for (int w = 0; w < nw; ++w) cask[(size_t)k * nw + w] = (found >= 0) ? 0ULL : hy_valid_word(nd, w);
}
__global__ void hy_ecc_centre_check(const int32_t* __restrict__ frontier, int ns, int nw,
                                    const int32_t* __restrict__ act,
                                    const hy_u64* __restrict__ cbits, HyCtr* ctr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns || act[k] != 1) return;
    hy_u64 any = 0ULL;
    for (int w = 0; w < nw; ++w) any |= cbits[(size_t)k * nw + w];
    if (!any) {
        atomicMax(&ctr->error, 3);
        ctr->err_node = frontier[k];
    }
}
__global__ void hy_ecc_adopt(DevTree T, const int32_t* __restrict__ frontier, int ns,
                             const int32_t* __restrict__ act, const int32_t* __restrict__ best,
                             const double* __restrict__ vJ, const double* __restrict__ vu,
                             const int32_t* __restrict__ vst, int nw, int fail_delta,
                             hy_u64* __restrict__ black, int32_t* __restrict__ next_ecc,
                             int32_t* __restrict__ next_lcss, HyCtr* ctr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns || act[k] != 2) return;
    const int id = frontier[k];
    const int p = T.p, nv = p + 1, n_u = T.n_u;
    bool ok = best[k] != fail_delta;
    for (int v = 0; v < nv; ++v) ok = ok && (vst[k * nv + v] == 0);
    if (!ok) {
        // V_R's retry (lib/oracle.py:198-218): blacklist the commutation, visit the node again
        black[(size_t)id * nw + (best[k] >> 6)] |= 1ULL << (best[k] & 63);
        atomicAdd(&ctr->blacklisted, 1ULL);
        next_ecc[atomicAdd(&ctr->next_ecc, 1)] = id;
        return;
    }
    for (int w = 0; w < nw; ++w) black[(size_t)id * nw + w] = 0ULL;   // local to the V_R call
    double* rec = T.rec + (size_t)id * T.rec_stride;
    for (int v = 0; v < nv; ++v) rec[rec_off_vcost(p) + v] = vJ[k * nv + v];
    for (int q = 0; q < nv * n_u; ++q) rec[rec_off_vinput(p) + q] = vu[(size_t)k * nv * n_u + q];
    T.didx[id] = best[k];
    T.flags[id] |= 2;
    next_lcss[atomicAdd(&ctr->next_lcss, 1)] = id;
}

// =============================================================================================
// host side
// =============================================================================================
struct HyState {
    bool on = false;
    int nw = 0;
    long long ch = 0;                    // frontier nodes per chunk
    DevBuf vf, cand, black, neg, tneg;   // per node
    int prune = 1;                       // inherit negative suboptimality-test verdicts
    DevBuf fr_ecc[2], fr_lcss[2];        // frontiers, ping-pong
    long long n_ecc = 0, n_lcss = 0;
    int cur = 0;
    DevBuf known1, ask, slk, vall, redo, koff, src, dst, dcomm, dsel2, cnt, tau, tval, st, alpha;
    DevBuf act, best, ths, vJ, vu, vst, Jth, Jth_st, Jmin, Jmin_st;
    DevBuf dselV, eoffV, dselT, eoffT, dselM, eoffM;
    DevBuf sp_k, sp_c0, sp_ij, sp_d, mids, Jm, um, mst, midforced, midask, midbits, moff;
    DevBuf ctr;
    // results of the point problems shared by (parameter, commutation, kind): the simplices
    // around an edge ask for the same midpoint solve and the same rows of phase-one problems
    // (K2Gather::pt, ehm_midtable.h); option "share_midpoints"
    DevBuf pt_state, pt_data;
    MidTable pt{nullptr, nullptr, 0u};
    DevBuf snaps;                        // DevCounters after every batched launch
    std::vector<int> snap_kind;          // LP kind of that launch
    HyCtr h{};
    int fail_delta = -1;                 // test hook: vertex solves of this commutation "fail"
    PersistDeal deal{0, 0, 0, 1, 1};     // sharded runs: deal depth / rank / world
    void release() {
        DevBuf* all[] = {&vf, &cand, &black, &neg, &tneg, &fr_ecc[0], &fr_ecc[1], &fr_lcss[0], &fr_lcss[1],
                         &known1, &ask, &slk, &vall, &redo, &koff, &src, &dst, &dcomm, &dsel2, &cnt,
                         &tau, &tval,
                         &st, &alpha, &act, &best, &ths, &vJ, &vu, &vst, &Jth, &Jth_st, &Jmin,
                         &Jmin_st, &dselV, &eoffV, &dselT, &eoffT, &dselM, &eoffM, &sp_k, &sp_c0,
                         &sp_ij, &sp_d, &mids, &Jm, &um, &mst, &midforced, &midask, &midbits,
                         &moff, &ctr, &snaps, &pt_state, &pt_data};
        for (DevBuf* b : all) b->release();
    }
};

#define HY_MAX_SNAPS 8192
#define HY_GRID(n) dim3((unsigned)(((long long)(n) + HY_BLOCK - 1) / HY_BLOCK)), dim3(HY_BLOCK)

static int hy_alloc(ehm_tree* T) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    const int nd = P->dp.n_delta, p = P->dp.p, nv = p + 1, n_u = P->dp.n_u;
    H.nw = (nd + 63) / 64;
    if (H.nw > 4)
        return fail(EHM_E_INVALID, "hybrid engine: at most 256 commutations (got %d)", nd);
    const int nw = H.nw;
    H.ch = std::max<long long>(1024, std::min<long long>(1LL << 17, (1LL << 22) / nd));
    if (const char* e = getenv("EHM_HY_CHUNK")) H.ch = std::max(64, atoi(e));
    const size_t ch = (size_t)H.ch, cap = (size_t)T->cap;
    int rc;
#define HY_ENSURE(buf, bytes) if ((rc = H.buf.ensure(bytes))) return rc
    HY_ENSURE(vf, cap * nv * nw * 8);
    HY_ENSURE(cand, cap * nw * 8);
    HY_ENSURE(black, cap * nw * 8);
    HY_ENSURE(neg, cap * nw * 8);
    HY_ENSURE(tneg, cap * 8);
    HY_ENSURE(known1, ch * nw * 8); HY_ENSURE(ask, ch * nw * 8); HY_ENSURE(slk, ch * nw * 8);
    HY_ENSURE(vall, ch * nw * 8);   HY_ENSURE(redo, ch * nw * 8);
    HY_ENSURE(koff, ch * 8);
    const size_t nlist = ch * (size_t)std::max(nd, nv);
    HY_ENSURE(src, nlist * 8);      HY_ENSURE(dst, nlist * 4);     HY_ENSURE(dcomm, nlist * 4);
    HY_ENSURE(dsel2, ch * nv * 4);
    HY_ENSURE(cnt, (size_t)(3 * nd + 8) * 4);
    HY_ENSURE(tau, ch * nd * 8);    HY_ENSURE(tval, ch * nd * 8);  HY_ENSURE(st, ch * nd * 4);
    HY_ENSURE(alpha, ch * nd * nv * 8);
    HY_ENSURE(act, ch * 4);         HY_ENSURE(best, ch * 4);       HY_ENSURE(ths, ch * p * 8);
    HY_ENSURE(vJ, ch * nv * 8);     HY_ENSURE(vu, ch * nv * n_u * 8); HY_ENSURE(vst, ch * nv * 4);
    HY_ENSURE(Jth, ch * 8);         HY_ENSURE(Jth_st, ch * 4);
    HY_ENSURE(Jmin, ch * 8);        HY_ENSURE(Jmin_st, ch * 4);
    HY_ENSURE(dselV, ch * nv * 4);  HY_ENSURE(eoffV, ch * nv * 8);
    HY_ENSURE(dselT, ch * 4);       HY_ENSURE(eoffT, ch * 8);
    HY_ENSURE(dselM, ch * 4);       HY_ENSURE(eoffM, ch * 8);
    HY_ENSURE(sp_k, ch * 4);        HY_ENSURE(sp_c0, ch * 4);      HY_ENSURE(sp_ij, ch * 8);
    HY_ENSURE(sp_d, ch * 4);        HY_ENSURE(mids, ch * p * 8);   HY_ENSURE(Jm, ch * 8);
    HY_ENSURE(um, ch * n_u * 8);    HY_ENSURE(mst, ch * 4);
    HY_ENSURE(midforced, ch * nw * 8); HY_ENSURE(midask, ch * nw * 8);
    HY_ENSURE(midbits, ch * nw * 8);   HY_ENSURE(moff, ch * 8);
    HY_ENSURE(ctr, sizeof(HyCtr));
    HY_ENSURE(snaps, (size_t)HY_MAX_SNAPS * sizeof(DevCounters));
    H.pt = MidTable{nullptr, nullptr, 0u};
    if (P->share_mid && P->solver_gen == 2 && !getenv("EHM_NO_MIDTABLE")) {
        // distinct (point, commutation) pairs: a few per node on average -- 8 slots per node
        // record, at most 4 M (a full neighbourhood degrades to "solve it yourself")
        size_t slots = 4096;
        while (slots < 8 * cap && slots < ((size_t)1 << 22)) slots <<= 1;
        HY_ENSURE(pt_state, slots * sizeof(unsigned long long));
        HY_ENSURE(pt_data, slots * MT_DOUBLES * sizeof(double));
        H.pt = MidTable{H.pt_state.as<unsigned long long>(), H.pt_data.as<double>(),
                        (unsigned int)(slots - 1)};
        HIP_TRY(hipMemsetAsync(H.pt.state, 0, slots * sizeof(unsigned long long), P->stream),
                EHM_E_HIP);
    }
#undef HY_ENSURE
    HIP_TRY(hipMemsetAsync(H.cnt.ptr, 0, (size_t)(3 * nd + 8) * 4, P->stream), EHM_E_HIP);
    return EHM_OK;
}

struct HyList {      // the work list sitting in H.src / H.dst / seg
    const long long* src;
    const int32_t* dst;
    const int32_t* seg;
    const int32_t* n_dev;
};

// event pair + counter snapshot around a batched launch: kernel seconds and iterations by LP kind
static void hy_stamp(ehm_tree* T) {
    if (!T->prob->hy_timing) return;
    hipEvent_t e;
    (void)hipEventCreate(&e);
    (void)hipEventRecord(e, T->prob->stream);
    T->run.evs.push_back(e);
}
static void hy_after_batch(ehm_tree* T, int lp_kind, int family) {
    HyState& H = *T->hy;
    if (!T->prob->hy_timing) return;
    hy_stamp(T);
    T->run.ev_kind.push_back(family);
    if (H.snap_kind.size() < HY_MAX_SNAPS) {
        (void)hipMemcpyAsync(H.snaps.as<DevCounters>() + H.snap_kind.size(), T->prob->d_cnt,
                             sizeof(DevCounters), hipMemcpyDeviceToDevice, T->prob->stream);
        H.snap_kind.push_back(lp_kind);
    }
}

static HyList hy_list_of(ehm_tree* T) {
    HyState& H = *T->hy;
    const int nd = T->prob->dp.n_delta;
    return HyList{H.src.as<long long>(), H.dst.as<int32_t>(), H.cnt.as<int32_t>() + nd,
                  &H.ctr.as<HyCtr>()->n_items};
}

static void hy_build_bits(ehm_tree* T, const hy_u64* bits, int ns, const long long* koff) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    const int nd = P->dp.n_delta, nw = H.nw;
    int32_t* count = H.cnt.as<int32_t>();
    int32_t* seg = count + nd;
    int32_t* cursor = seg + nd + 1;
    const long long threads = (long long)nd * ((ns + 63) & ~63);
    hipLaunchKernelGGL(hy_list_count, HY_GRID(threads), 0, P->stream, bits, ns, nd, nw, count);
    hipLaunchKernelGGL(hy_list_scan, dim3(1), dim3(64), 0, P->stream, count, nd, seg, cursor,
                       &H.ctr.as<HyCtr>()->n_items);
    hipLaunchKernelGGL(hy_list_scatter, HY_GRID(threads), 0, P->stream, bits, ns, nd, nw, koff,
                       cursor, H.src.as<long long>(), H.dst.as<int32_t>(), H.dcomm.as<int32_t>());
    P->launches += 3;
}

static void hy_build_sel(ehm_tree* T, const int32_t* dsel, int ne, const long long* eoff) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    const int nd = P->dp.n_delta;
    int32_t* count = H.cnt.as<int32_t>();
    int32_t* seg = count + nd;
    int32_t* cursor = seg + nd + 1;
    hipLaunchKernelGGL(hy_sel_count, HY_GRID(ne), 0, P->stream, dsel, ne, count);
    hipLaunchKernelGGL(hy_list_scan, dim3(1), dim3(64), 0, P->stream, count, nd, seg, cursor,
                       &H.ctr.as<HyCtr>()->n_items);
    hipLaunchKernelGGL(hy_sel_scatter, HY_GRID(ne), 0, P->stream, dsel, ne, eoff, cursor,
                       H.src.as<long long>(), H.dst.as<int32_t>(), H.dcomm.as<int32_t>());
    P->launches += 3;
}

static void hy_run_simplex_v1(ehm_tree* T, int mode, double* obj, double* alpha, int32_t* status);
static void hy_run_point_v1(ehm_tree* T, const double* base, int feas, double* J, double* u0,
                            int32_t* status);

// batched problems over the simplices the current list points at (records of the node pool)
static int hy_run_simplex(ehm_tree* T, int mode, double* obj, double* alpha, int32_t* status) {
    ehm_problem* P = T->prob;
    if (P->solver_gen == 1) {      // cross-check mode: everything on the one-wavefront kernels
        hy_stamp(T);
        hy_run_simplex_v1(T, mode, obj, alpha, status);
        hy_after_batch(T, (mode == SX_SLACK) ? LP_SLACK : (mode == SX_FEAS) ? LP_FEAS_SIMPLEX
                                                                             : LP_MIN_SIMPLEX, 0);
        return EHM_OK;
    }
    const int kind = (mode == SX_SLACK) ? LP_SLACK : (mode == SX_FEAS) ? LP_FEAS_SIMPLEX
                                                                        : LP_MIN_SIMPLEX;
    K2Cfg cfg;
    int rc = k2_config(P, kind, kind, 1LL << 40, cfg);
    if (rc) return rc;
    const HyList L = hy_list_of(T);
    // sign-only stops: a phase-one optimum is only compared with ~0; a slack below zero only has
    // to be known as negative (bar_E), one above is ranked by its value (bar_D)
    const int sign_mode = P->decide_full ? 0 : (mode == SX_FEAS) ? 1 : (mode == SX_SLACK) ? 2 : 0;
    K2Gather G{L.src, L.dst, L.n_dev, rec_off_vcost(P->dp.p), nullptr, sign_mode};
    hy_stamp(T);
    cfg.api->simplex(cfg.L, P->dp, 0, T->dt.rec, nullptr, L.seg, mode, obj, alpha, status, nullptr,
                     P->d_cnt, G);
    hy_after_batch(T, kind, 0);
    P->launches++;
    return EHM_OK;
}

// batched P_theta_delta (or its phase-one form) at the parameters the current list points at
static int hy_run_point(ehm_tree* T, const double* base, int feas, double* J, double* u0,
                        int32_t* status) {
    ehm_problem* P = T->prob;
    if (P->solver_gen == 1) {
        hy_stamp(T);
        hy_run_point_v1(T, base, feas, J, u0, status);
        hy_after_batch(T, feas ? LP_FEAS : LP_POINT, 1);
        return EHM_OK;
    }
    K2Cfg cfg;
    int rc = k2_config(P, feas ? LP_FEAS : LP_POINT, feas ? LP_FEAS : LP_POINT, 1LL << 40, cfg);
    if (rc) return rc;
    const HyList L = hy_list_of(T);
    K2Gather G{L.src, L.dst, L.n_dev, 0, nullptr, (feas && !P->decide_full) ? 1 : 0};
    G.pt = T->hy->pt;
    hy_stamp(T);
    cfg.api->point(cfg.L, P->dp, 0, base, L.seg, feas, J, u0, status, nullptr, P->d_cnt, G);
    hy_after_batch(T, feas ? LP_FEAS : LP_POINT, 1);
    P->launches++;
    return EHM_OK;
}

#define HY_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// The same lists on the one-wavefront kernels (ehm_kernels.h): the numerical safety net of the
// batched oracles (simplex_batch / point_batch above), here for the few problems per ten million
// the shared-block solver leaves stalled.  Results overwrite the stalled ones.
static void hy_run_simplex_v1(ehm_tree* T, int mode, double* obj, double* alpha, int32_t* status) {
    ehm_problem* P = T->prob;
    const HyList L = hy_list_of(T);
    K2Gather G{L.src, L.dst, L.n_dev, rec_off_vcost(P->dp.p), nullptr, 0};
    hipLaunchKernelGGL(k_simplex_batch, dim3(P->num_cu * 8), dim3(64), P->lds_simplex, P->stream,
                       P->dp, 0LL, T->dt.rec, (const double*)nullptr, T->hy->dcomm.as<int32_t>(),
                       mode, obj, alpha, status, (int32_t*)nullptr, P->d_cnt, G);
    if (P->solver_gen == 2)
        hipLaunchKernelGGL(hy_add_items, dim3(1), dim3(64), 0, P->stream, T->hy->ctr.as<HyCtr>());
    P->launches += 2;
}
static void hy_run_point_v1(ehm_tree* T, const double* base, int feas, double* J, double* u0,
                            int32_t* status) {
    ehm_problem* P = T->prob;
    const HyList L = hy_list_of(T);
    K2Gather G{L.src, L.dst, L.n_dev, 0, nullptr, 0};
    hipLaunchKernelGGL(k_point_batch, dim3(P->num_cu * 8), dim3(64), P->lds_point, P->stream,
                       P->dp, 0LL, base, T->hy->dcomm.as<int32_t>(), feas, J, u0, status,
                       (int32_t*)nullptr, P->d_cnt, G);
    if (P->solver_gen == 2)
        hipLaunchKernelGGL(hy_add_items, dim3(1), dim3(64), 0, P->stream, T->hy->ctr.as<HyCtr>());
    P->launches += 2;
}
// stalled entries of a one-commutation-per-entry batch, again
static void hy_retry_sel_point(ehm_tree* T, const int32_t* dsel, int ne, const long long* eoff,
                               const double* base, double* J, double* u0, int32_t* status) {
    ehm_problem* P = T->prob;
    if (!P->v1_ok || P->solver_gen == 1) return;
    HyState& H = *T->hy;
    hipLaunchKernelGGL(hy_sel_stalled, HY_GRID(ne), 0, P->stream, dsel, ne, status,
                       H.dsel2.as<int32_t>());
    hy_build_sel(T, H.dsel2.as<int32_t>(), ne, eoff);
    hy_run_point_v1(T, base, 0, J, u0, status);
}

// the split of the collected nodes: midpoint solve (lcss), the midpoint's feasibility row,
// children
static int hy_split_stage(ehm_tree* T, const int32_t* fr, int ns, int has_data, int32_t* next) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    const int nd = P->dp.n_delta, nw = H.nw;
    HyCtr* ctr = H.ctr.as<HyCtr>();
    HIP_TRY(hipMemsetAsync(&ctr->n_split, 0, 4, P->stream), EHM_E_HIP);
    HIP_TRY(hipMemsetAsync(H.sp_d.ptr, 0xFF, (size_t)ns * 4, P->stream), EHM_E_HIP);
    HIP_TRY(hipMemsetAsync(H.midask.ptr, 0, (size_t)ns * nw * 8, P->stream), EHM_E_HIP);
    hipLaunchKernelGGL(hy_split_collect, HY_GRID(ns), 0, P->stream, T->dt, fr, ns, nd, nw, has_data,
                       (int)T->limit, H.act.as<int32_t>(), H.best.as<int32_t>(),
                       H.vf.as<hy_u64>(), H.cand.as<hy_u64>(), H.sp_k.as<int32_t>(),
                       H.sp_c0.as<int32_t>(), H.sp_ij.as<int32_t>(), H.sp_d.as<int32_t>(),
                       H.mids.as<double>(), H.midforced.as<hy_u64>(), H.midask.as<hy_u64>(),
                       H.moff.as<long long>(), ctr);
    if (has_data) {
        hy_build_sel(T, H.sp_d.as<int32_t>(), ns, H.moff.as<long long>());
        HY_TRY(hy_run_point(T, H.mids.as<double>(), 0, H.Jm.as<double>(), H.um.as<double>(),
                            H.mst.as<int32_t>()));
        hy_retry_sel_point(T, H.sp_d.as<int32_t>(), ns, H.moff.as<long long>(),
                           H.mids.as<double>(), H.Jm.as<double>(), H.um.as<double>(),
                           H.mst.as<int32_t>());
    }
    hy_build_bits(T, H.midask.as<hy_u64>(), ns, H.moff.as<long long>());
    HY_TRY(hy_run_point(T, H.mids.as<double>(), 1, H.tau.as<double>(), nullptr, nullptr));
    hipLaunchKernelGGL(hy_bits_finish, HY_GRID((long long)ns * nw), 0, P->stream, ns, nd, nw,
                       H.midforced.as<hy_u64>(), H.midask.as<hy_u64>(), H.tau.as<double>(),
                       EHM_FEAS_TOL, H.midbits.as<hy_u64>());
    hipLaunchKernelGGL(hy_children, dim3((unsigned)ns), dim3(64), 0, P->stream, T->dt, fr, nd, nw,
                       has_data, H.act.as<int32_t>(), H.sp_k.as<int32_t>(), H.sp_c0.as<int32_t>(),
                       H.sp_ij.as<int32_t>(), H.sp_d.as<int32_t>(), H.mids.as<double>(),
                       H.Jm.as<double>(), H.um.as<double>(), H.mst.as<int32_t>(),
                       H.vJ.as<double>(), H.vu.as<double>(), H.midbits.as<hy_u64>(),
                       H.vf.as<hy_u64>(), H.cand.as<hy_u64>(), H.black.as<hy_u64>(),
                       H.neg.as<hy_u64>(), H.tneg.as<double>(), next, ctr, H.deal);
    P->launches += 3;
    return EHM_OK;
}

static int hy_lcss_chunk(ehm_tree* T, const int32_t* fr, int ns, int32_t* next_lcss) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    const int nd = P->dp.n_delta, nw = H.nw, nv = P->dp.p + 1;
    HyCtr* ctr = H.ctr.as<HyCtr>();
    const long long nbw = (long long)ns * nw;
    hipLaunchKernelGGL(hy_lcss_classify, HY_GRID(nbw), 0, P->stream, T->dt, fr, ns, nd, nw,
                       H.vf.as<hy_u64>(), H.cand.as<hy_u64>(), H.neg.as<hy_u64>(),
                       H.known1.as<hy_u64>(), H.ask.as<hy_u64>(), H.vall.as<hy_u64>(),
                       H.koff.as<long long>());
    // phase one over the simplex for the commutations nothing is known about
    hy_build_bits(T, H.ask.as<hy_u64>(), ns, H.koff.as<long long>());
    HY_TRY(hy_run_simplex(T, SX_FEAS, H.tau.as<double>(), nullptr, nullptr));
    hipLaunchKernelGGL(hy_after_phase1, HY_GRID(nbw), 0, P->stream, ns, nd, nw,
                       H.known1.as<hy_u64>(), H.ask.as<hy_u64>(), H.tau.as<double>(),
                       EHM_FEAS_TOL, H.slk.as<hy_u64>());
    // suboptimality-test problem of every commutation feasible somewhere on the simplex
    hy_build_bits(T, H.slk.as<hy_u64>(), ns, H.koff.as<long long>());
    HY_TRY(hy_run_simplex(T, SX_SLACK, H.tval.as<double>(), H.alpha.as<double>(),
                          H.st.as<int32_t>()));
    hipLaunchKernelGGL(hy_redo_bits, HY_GRID(nbw), 0, P->stream, ns, nd, nw,
                       H.known1.as<hy_u64>(), H.st.as<int32_t>(), H.redo.as<hy_u64>());
    hy_build_bits(T, H.redo.as<hy_u64>(), ns, H.koff.as<long long>());
    HY_TRY(hy_run_simplex(T, SX_FEAS, H.tau.as<double>(), nullptr, nullptr));
    if (P->v1_ok && P->solver_gen == 2) {
        hipLaunchKernelGGL(hy_retry_bits, HY_GRID(nbw), 0, P->stream, ns, nd, nw,
                           H.slk.as<hy_u64>(), H.st.as<int32_t>(), H.tau.as<double>(),
                           EHM_SLIVER_TOL, H.redo.as<hy_u64>());
        hy_build_bits(T, H.redo.as<hy_u64>(), ns, H.koff.as<long long>());
        hy_run_simplex_v1(T, SX_SLACK, H.tval.as<double>(), H.alpha.as<double>(),
                          H.st.as<int32_t>());
    }
    hipLaunchKernelGGL(hy_lcss_decide, HY_GRID(ns), 0, P->stream, T->dt, fr, ns, nd, nw,
                       H.slk.as<hy_u64>(), H.vall.as<hy_u64>(), H.black.as<hy_u64>(),
                       H.tau.as<double>(), H.tval.as<double>(), H.st.as<int32_t>(),
                       H.alpha.as<double>(), EHM_SLIVER_TOL, EHM_TIE_TOL, T->run.max_depth,
                       H.prune, H.neg.as<hy_u64>(), H.tneg.as<double>(), H.cand.as<hy_u64>(), H.act.as<int32_t>(), H.best.as<int32_t>(),
                       H.ths.as<double>(), ctr, (uint32_t)P->any_admissible);
    // the nodes with a better commutation: its vertex solves and in_variability_ball
    hipLaunchKernelGGL(hy_delta_entries, HY_GRID(ns), 0, P->stream, T->dt, fr, ns,
                       H.act.as<int32_t>(), H.best.as<int32_t>(), H.dselV.as<int32_t>(),
                       H.eoffV.as<long long>(), H.dselT.as<int32_t>(), H.eoffT.as<long long>(),
                       H.dselM.as<int32_t>(), H.eoffM.as<long long>());
    P->launches += 5;
    hy_build_sel(T, H.dselV.as<int32_t>(), ns * nv, H.eoffV.as<long long>());
    HY_TRY(hy_run_point(T, T->dt.rec, 0, H.vJ.as<double>(), H.vu.as<double>(), H.vst.as<int32_t>()));
    hy_retry_sel_point(T, H.dselV.as<int32_t>(), ns * nv, H.eoffV.as<long long>(), T->dt.rec,
                       H.vJ.as<double>(), H.vu.as<double>(), H.vst.as<int32_t>());
    hy_build_sel(T, H.dselT.as<int32_t>(), ns, H.eoffT.as<long long>());
    HY_TRY(hy_run_point(T, H.ths.as<double>(), 0, H.Jth.as<double>(), nullptr,
                        H.Jth_st.as<int32_t>()));
    hy_retry_sel_point(T, H.dselT.as<int32_t>(), ns, H.eoffT.as<long long>(), H.ths.as<double>(),
                       H.Jth.as<double>(), nullptr, H.Jth_st.as<int32_t>());
    hy_build_sel(T, H.dselM.as<int32_t>(), ns, H.eoffM.as<long long>());
    HY_TRY(hy_run_simplex(T, SX_MIN, H.Jmin.as<double>(), nullptr, H.Jmin_st.as<int32_t>()));
    if (P->v1_ok && P->solver_gen == 2) {
        hipLaunchKernelGGL(hy_sel_stalled, HY_GRID(ns), 0, P->stream, H.dselM.as<int32_t>(), ns,
                           H.Jmin_st.as<int32_t>(), H.dsel2.as<int32_t>());
        hy_build_sel(T, H.dsel2.as<int32_t>(), ns, H.eoffM.as<long long>());
        hy_run_simplex_v1(T, SX_MIN, H.Jmin.as<double>(), nullptr, H.Jmin_st.as<int32_t>());
    }
    hipLaunchKernelGGL(hy_varsmall, HY_GRID(ns), 0, P->stream, T->dt, fr, ns, nw,
                       H.act.as<int32_t>(), H.best.as<int32_t>(), H.vJ.as<double>(),
                       H.vu.as<double>(), H.vst.as<int32_t>(), H.Jth.as<double>(),
                       H.Jth_st.as<int32_t>(), H.Jmin.as<double>(), H.Jmin_st.as<int32_t>(),
                       P->dp.eps_a, P->dp.eps_r, H.fail_delta, H.black.as<hy_u64>(),
                       H.neg.as<hy_u64>(), H.tneg.as<double>(), ctr);
    hipLaunchKernelGGL(hy_revisit, HY_GRID(ns), 0, P->stream, fr, ns, H.act.as<int32_t>(),
                       next_lcss, ctr);
    P->launches += 2;
    return hy_split_stage(T, fr, ns, 1, next_lcss);
}

static int hy_ecc_chunk(ehm_tree* T, const int32_t* fr, int ns, int32_t* next_ecc,
                        int32_t* next_lcss) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    const int nd = P->dp.n_delta, nw = H.nw, nv = P->dp.p + 1;
    HyCtr* ctr = H.ctr.as<HyCtr>();
    hipLaunchKernelGGL(hy_ecc_classify, HY_GRID(ns), 0, P->stream, T->dt, fr, ns, nd, nw,
                       H.vf.as<hy_u64>(), H.black.as<hy_u64>(), H.act.as<int32_t>(),
                       H.best.as<int32_t>(),
                       H.ths.as<double>(), H.ask.as<hy_u64>(), H.koff.as<long long>(), ctr,
                       (uint32_t)P->any_admissible);
    // barycentre check of the nodes without a commutation feasible at every vertex
    hy_build_bits(T, H.ask.as<hy_u64>(), ns, H.koff.as<long long>());
    HY_TRY(hy_run_point(T, H.ths.as<double>(), 1, H.tau.as<double>(), nullptr, nullptr));
    hipLaunchKernelGGL(hy_bits_finish, HY_GRID((long long)ns * nw), 0, P->stream, ns, nd, nw,
                       (const hy_u64*)nullptr, H.ask.as<hy_u64>(), H.tau.as<double>(),
                       EHM_FEAS_TOL, H.slk.as<hy_u64>());
    hipLaunchKernelGGL(hy_ecc_centre_check, HY_GRID(ns), 0, P->stream, fr, ns, nw,
                       H.act.as<int32_t>(), H.slk.as<hy_u64>(), ctr);
    // vertex solves of the commutations V_R picked
    hipLaunchKernelGGL(hy_delta_entries, HY_GRID(ns), 0, P->stream, T->dt, fr, ns,
                       H.act.as<int32_t>(), H.best.as<int32_t>(), H.dselV.as<int32_t>(),
                       H.eoffV.as<long long>(), (int32_t*)nullptr, (long long*)nullptr,
                       (int32_t*)nullptr, (long long*)nullptr);
    hy_build_sel(T, H.dselV.as<int32_t>(), ns * nv, H.eoffV.as<long long>());
    HY_TRY(hy_run_point(T, T->dt.rec, 0, H.vJ.as<double>(), H.vu.as<double>(), H.vst.as<int32_t>()));
    hy_retry_sel_point(T, H.dselV.as<int32_t>(), ns * nv, H.eoffV.as<long long>(), T->dt.rec,
                       H.vJ.as<double>(), H.vu.as<double>(), H.vst.as<int32_t>());
    hipLaunchKernelGGL(hy_ecc_adopt, HY_GRID(ns), 0, P->stream, T->dt, fr, ns, H.act.as<int32_t>(),
                       H.best.as<int32_t>(), H.vJ.as<double>(), H.vu.as<double>(),
                       H.vst.as<int32_t>(), nw, H.fail_delta, H.black.as<hy_u64>(), next_ecc,
                       next_lcss, ctr);
    P->launches += 5;
    return hy_split_stage(T, fr, ns, 0, next_ecc);
}

static int hy_read_ctr(ehm_tree* T) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    HIP_TRY(hipMemcpyAsync(&H.h, H.ctr.ptr, sizeof(HyCtr), hipMemcpyDeviceToHost, P->stream),
            EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    if (H.h.error == 1)
        return fail(EHM_E_CAPACITY, "node pool exhausted at %d nodes (max_nodes=%lld)",
                    H.h.n_nodes, T->limit);
    if (H.h.error == 3)
        return fail(EHM_E_INFEASIBLE, "STOP, Theta contains infeasible regions (node %d)",
                    H.h.err_node);
    if (H.h.error != 0) {
        static const char* what[] = {"?", "suboptimality-test problem", "min over the simplex",
                                     "midpoint solve"};
        if (const char* path = getenv("EHM_DUMP_FAIL")) {      // debugging aid: the instance
            const int p = P->dp.p, nv = p + 1, stride = T->dt.rec_stride;
            std::vector<double> rec((size_t)stride);
            (void)hipMemcpy(rec.data(), T->dt.rec + (size_t)H.h.err_node * stride, stride * 8,
                            hipMemcpyDeviceToHost);
            if (FILE* fp = fopen(path, "w")) {
                fprintf(fp, "%d %d %d\n", H.h.err_kind >> 4, p, H.h.err_kind & 3);
                for (int q = 0; q < nv * p; ++q) fprintf(fp, "%.17g ", rec[(size_t)q]);
                fprintf(fp, "\n");
                for (int q = 0; q < nv; ++q) fprintf(fp, "%.17g ", rec[(size_t)(nv * p + q)]);
                fprintf(fp, "\n%.17g %.17g\n", P->dp.eps_a, P->dp.eps_r);
                fclose(fp);
            }
        }
        return fail(EHM_E_NUMERIC, "node %d: %s with commutation %d did not converge",
                    H.h.err_node, what[H.h.err_kind & 3], H.h.err_kind >> 4);
    }
    return EHM_OK;
}

// one frontier sweep: every ecc node and every lcss node of the live frontiers is visited once
static int hy_sweep(ehm_tree* T) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    auto& R = T->run;
    const int cur = H.cur, nxt = 1 - cur;
    int rc;
    if ((rc = H.fr_ecc[nxt].ensure((size_t)(3 * H.n_ecc + 64) * 4))) return rc;
    if ((rc = H.fr_lcss[nxt].ensure((size_t)(2 * H.n_lcss + H.n_ecc + 64) * 4))) return rc;
    int32_t* next_ecc = H.fr_ecc[nxt].as<int32_t>();
    int32_t* next_lcss = H.fr_lcss[nxt].as<int32_t>();
    for (long long f0 = 0; f0 < H.n_ecc; f0 += H.ch) {
        const int ns = (int)std::min<long long>(H.ch, H.n_ecc - f0);
        HY_TRY(hy_ecc_chunk(T, H.fr_ecc[cur].as<int32_t>() + f0, ns, next_ecc, next_lcss));
    }
    for (long long f0 = 0; f0 < H.n_lcss; f0 += H.ch) {
        const int ns = (int)std::min<long long>(H.ch, H.n_lcss - f0);
        HY_TRY(hy_lcss_chunk(T, H.fr_lcss[cur].as<int32_t>() + f0, ns, next_lcss));
    }
    HIP_TRY(hipGetLastError(), EHM_E_HIP);
    HY_TRY(hy_read_ctr(T));
    H.n_ecc = H.h.next_ecc;
    H.n_lcss = H.h.next_lcss;
    H.cur = nxt;
    HyCtr* ctr = H.ctr.as<HyCtr>();
    HIP_TRY(hipMemsetAsync(&ctr->next_ecc, 0, 8, P->stream), EHM_E_HIP);   // next_ecc, next_lcss
    R.n_nodes = H.h.n_nodes;
    R.n_closed = (long long)H.h.closed;
    R.ref_solves = (long long)H.h.ref_solves;
    R.truncated = H.h.truncated;
    R.depth = H.h.max_depth_seen;
    R.nf = H.n_ecc + H.n_lcss;
    ++R.sweeps;
    if (H.h.dealt && !R.sharded) {
        // everything up to here was grown identically on every rank (level-synchronous sweeps)
        DevCounters cs;
        HY_TRY(read_counters(P, cs));
        R.pre_closed = R.n_closed;
        R.pre_nodes = R.n_nodes;
        R.pre_solves = (long long)(cs.lp_solves - R.c0.lp_solves);
        R.sharded = true;
    }
    if (getenv("EHM_HY_TRACE"))
        fprintf(stderr, "[hybrid] sweep %d: nodes %d closed %llu splits %llu swaps %llu -> next ecc "
                        "%lld lcss %lld\n", R.sweeps, H.h.n_nodes, H.h.closed, H.h.splits,
                H.h.swaps, H.n_ecc, H.n_lcss);
    return EHM_OK;
}

// roots of a hybrid run: structure words, feasibility rows of every root vertex, frontiers
static int hy_begin(ehm_tree* T, int64_t n_roots, const ehm_node_init* init) {
    ehm_problem* P = T->prob;
    auto& R = T->run;
    if (P->solver_gen != 2 && !P->v1_ok)
        return fail(EHM_E_INVALID, "this problem does not fit the generation-1 kernels");
    T->hy = new HyState();
    HyState& H = *T->hy;
    H.on = true;
    int rc = hy_alloc(T);
    if (rc) return rc;
    const int nd = P->dp.n_delta, nw = H.nw, p = P->dp.p, nv = p + 1;
    if (const char* e = getenv("EHM_HY_FAIL_DELTA")) H.fail_delta = atoi(e);
    if (getenv("EHM_HY_NO_PRUNE")) H.prune = 0;
    std::vector<int32_t> didx((size_t)n_roots, -1);
    std::vector<uint8_t> flags((size_t)n_roots, 0);
    if (R.action == 1) {
        if (!init || !init->delta || !init->vcost || !init->vinput)
            return fail(EHM_E_INVALID, "action 'lcss' needs delta / vertex costs / vertex inputs");
        if ((rc = map_deltas(P, n_roots, init->delta, didx))) return rc;
        std::fill(flags.begin(), flags.end(), (uint8_t)2);
    }
    HIP_TRY(hipMemcpyAsync(T->dt.didx, didx.data(), didx.size() * 4, hipMemcpyHostToDevice,
                           P->stream), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(T->dt.flags, flags.data(), flags.size(), hipMemcpyHostToDevice,
                           P->stream), EHM_E_HIP);
    HyCtr h{};
    h.n_nodes = (int)n_roots;
    h.min_margin_bits = 0x7FF0000000000000ULL;
    HIP_TRY(hipMemcpyAsync(H.ctr.ptr, &h, sizeof h, hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);      // didx / flags / h leave scope
    T->dt.code = nullptr;
    const bool sharded = R.shard_world > 1 && R.shard_min >= 0;
    if (sharded || P->any_admissible) {
        // path codes: the deal of a sharded run and the draws of option any_admissible hash them
        if (sharded && R.deal_depth <= 0)
            return fail(EHM_E_INVALID, "sharded multi-commutation runs need ehm_run_opts.deal_depth");
        if (sharded) H.deal = PersistDeal{0, R.deal_depth, R.shard_rank, R.shard_world, 1};
        if ((rc = T->code.ensure((size_t)T->cap * 4))) return rc;
        T->dt.code = T->code.as<uint32_t>();
        std::vector<uint32_t> codes((size_t)n_roots);
        for (int64_t k = 0; k < n_roots; ++k) codes[(size_t)k] = (uint32_t)k;
        HIP_TRY(hipMemcpyAsync(T->dt.code, codes.data(), codes.size() * 4, hipMemcpyHostToDevice,
                               P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    }
    hipLaunchKernelGGL(hy_node_init, HY_GRID(n_roots * nw), 0, P->stream, 0, (int)n_roots, nd, nw,
                       H.cand.as<hy_u64>(), H.black.as<hy_u64>(), H.neg.as<hy_u64>(),
                       H.tneg.as<double>());
    // feasibility of every commutation at every root vertex
    const long long rows = n_roots * nv;
    for (long long r0 = 0; r0 < rows; r0 += H.ch) {
        const int nr = (int)std::min<long long>(H.ch, rows - r0);
        hipLaunchKernelGGL(hy_root_rows, HY_GRID((long long)nr * nw), 0, P->stream, (int)r0, nr, nd,
                           nw, nv, p, T->dt.rec_stride, H.ask.as<hy_u64>(),
                           H.koff.as<long long>());
        hy_build_bits(T, H.ask.as<hy_u64>(), nr, H.koff.as<long long>());
        HY_TRY(hy_run_point(T, T->dt.rec, 1, H.tau.as<double>(), nullptr, nullptr));
        hipLaunchKernelGGL(hy_bits_finish, HY_GRID((long long)nr * nw), 0, P->stream, nr, nd, nw,
                           (const hy_u64*)nullptr, H.ask.as<hy_u64>(), H.tau.as<double>(),
                           EHM_FEAS_TOL, H.vf.as<hy_u64>() + (size_t)r0 * nw);
        P->launches += 2;
    }
    if (R.shard_world > 1 && R.shard_rank > 0 && R.shard_min < 0) {
        // dynamic balancing from a single source: this rank starts with nothing
        std::vector<uint8_t> fl((size_t)n_roots, (uint8_t)(flags[0] | 4));
        HIP_TRY(hipMemcpyAsync(T->dt.flags, fl.data(), fl.size(), hipMemcpyHostToDevice,
                               P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
        H.cur = 0;
        H.n_ecc = H.n_lcss = 0;
        R.nf = 0;
        R.sharded = true;
        R.pre_nodes = n_roots;
        T->unordered = true;
        T->keep_ids = true;
        int rc2;
        if ((rc2 = H.fr_lcss[0].ensure(4096)) || (rc2 = H.fr_ecc[0].ensure(4096))) return rc2;
        return EHM_OK;
    }
    DevBuf& f0 = (R.action == 1) ? H.fr_lcss[0] : H.fr_ecc[0];
    if ((rc = f0.ensure((size_t)n_roots * 4))) return rc;
    std::vector<int32_t> ids((size_t)n_roots);
    for (int64_t k = 0; k < n_roots; ++k) ids[(size_t)k] = (int32_t)k;
    HIP_TRY(hipMemcpyAsync(f0.ptr, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, P->stream),
            EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    HIP_TRY(hipGetLastError(), EHM_E_HIP);
    H.cur = 0;
    H.n_ecc = (R.action == 1) ? 0 : n_roots;
    H.n_lcss = (R.action == 1) ? n_roots : 0;
    T->unordered = true;      // node ids follow the allocation order; the export relabels
    return EHM_OK;
}

// iterations / solves by LP kind from the counter snapshots (ehm_partition_finish)
static int hy_kind_totals(ehm_tree* T, int64_t (&solves)[5], int64_t (&iters)[5]) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    for (int k = 0; k < 5; ++k) solves[k] = iters[k] = 0;
    const size_t n = H.snap_kind.size();
    if (n == 0) return EHM_OK;
    std::vector<DevCounters> c(n);
    HIP_TRY(hipMemcpyAsync(c.data(), H.snaps.ptr, n * sizeof(DevCounters), hipMemcpyDeviceToHost,
                           P->stream), EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    DevCounters prev = T->run.c0;
    for (size_t i = 0; i < n; ++i) {
        solves[H.snap_kind[i]] += (int64_t)(c[i].lp_solves - prev.lp_solves);
        iters[H.snap_kind[i]] += (int64_t)(c[i].ipm_iters - prev.ipm_iters);
        prev = c[i];
    }
    return EHM_OK;
}

// ---- frontier hand-over between ranks (ehm_partition_take / _give for multi-commutation runs) ---
// A node travels with what the engine knows about it: record | vf rows | cand | neg | black (bit
// rows as raw 64-bit words in double slots) | tneg.  Only lcss nodes (they carry data) move.
__host__ __device__ inline int hy_record_doubles(int p, int n_u, int nw) {
    return rec_doubles(p, n_u) + (p + 1 + 3) * nw + 1;
}
__global__ void hy_take_nodes(DevTree T, const int32_t* __restrict__ ids, int n, int nw,
                              const hy_u64* __restrict__ vf, const hy_u64* __restrict__ cand,
                              const hy_u64* __restrict__ neg, const hy_u64* __restrict__ black,
                              const double* __restrict__ tneg, double* __restrict__ out,
                              int32_t* __restrict__ meta) {
    const int k = blockIdx.x;
    if (k >= n) return;
    const int id = ids[k];
    const int p = T.p, nv = p + 1, nrec = rec_doubles(p, T.n_u);
    const int W = hy_record_doubles(p, T.n_u, nw);
    double* row = out + (size_t)k * W;
    const double* r = T.rec + (size_t)id * T.rec_stride;
    for (int q = threadIdx.x; q < nrec; q += blockDim.x) row[q] = r[q];
    hy_u64* bits = reinterpret_cast<hy_u64*>(row + nrec);
    for (int q = threadIdx.x; q < nv * nw; q += blockDim.x) bits[q] = vf[(size_t)id * nv * nw + q];
    for (int q = threadIdx.x; q < nw; q += blockDim.x) {
        bits[nv * nw + q] = cand[(size_t)id * nw + q];
        bits[(nv + 1) * nw + q] = neg[(size_t)id * nw + q];
        bits[(nv + 2) * nw + q] = black[(size_t)id * nw + q];
    }
    if (threadIdx.x == 0) {
        row[W - 1] = tneg[id];
        meta[2 * k] = T.didx[id];
        meta[2 * k + 1] = T.depth[id];
        T.flags[id] |= 4;                      // subtree now owned by another rank
    }
}
__global__ void hy_give_nodes(DevTree T, int first, int n, int nw, const double* __restrict__ in,
                              const int32_t* __restrict__ meta, hy_u64* __restrict__ vf,
                              hy_u64* __restrict__ cand, hy_u64* __restrict__ neg,
                              hy_u64* __restrict__ black, double* __restrict__ tneg,
                              int32_t* __restrict__ frontier, int frontier_at) {
    const int k = blockIdx.x;
    if (k >= n) return;
    const int id = first + k;
    const int p = T.p, nv = p + 1, nrec = rec_doubles(p, T.n_u);
    const int W = hy_record_doubles(p, T.n_u, nw);
    const double* row = in + (size_t)k * W;
    double* r = T.rec + (size_t)id * T.rec_stride;
    for (int q = threadIdx.x; q < nrec; q += blockDim.x) r[q] = row[q];
    const hy_u64* bits = reinterpret_cast<const hy_u64*>(row + nrec);
    for (int q = threadIdx.x; q < nv * nw; q += blockDim.x) vf[(size_t)id * nv * nw + q] = bits[q];
    for (int q = threadIdx.x; q < nw; q += blockDim.x) {
        cand[(size_t)id * nw + q] = bits[nv * nw + q];
        neg[(size_t)id * nw + q] = bits[(nv + 1) * nw + q];
        black[(size_t)id * nw + q] = bits[(nv + 2) * nw + q];
    }
    if (threadIdx.x == 0) {
        tneg[id] = row[W - 1];
        T.left[id] = -1;
        T.didx[id] = meta[2 * k];
        T.depth[id] = meta[2 * k + 1];
        T.flags[id] = 2 | 32;                  // has data, received from another rank
        T.tstar[id] = 0.0;
        if (T.code) T.code[id] = 0u;
        frontier[frontier_at + k] = id;
    }
}

static int hy_take(ehm_tree* T, int64_t count, int32_t* node_ids, double* records, int32_t* meta) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    auto& R = T->run;
    if (count > H.n_lcss)
        return fail(EHM_E_INVALID, "cannot take %lld of %lld movable frontier nodes",
                    (long long)count, H.n_lcss);
    if (count == 0) return EHM_OK;
    const int W = hy_record_doubles(P->dp.p, P->dp.n_u, H.nw);
    int rc;
    if ((rc = P->out0.ensure((size_t)count * W * sizeof(double)))) return rc;
    if ((rc = P->out2.ensure((size_t)count * 2 * sizeof(int32_t)))) return rc;
    const int32_t* cur = H.fr_lcss[H.cur].as<int32_t>() + (H.n_lcss - count);
    hipLaunchKernelGGL(hy_take_nodes, dim3((unsigned)count), dim3(64), 0, P->stream, T->dt, cur,
                       (int)count, H.nw, H.vf.as<hy_u64>(), H.cand.as<hy_u64>(),
                       H.neg.as<hy_u64>(), H.black.as<hy_u64>(), H.tneg.as<double>(),
                       P->out0.as<double>(), P->out2.as<int32_t>());
    HIP_TRY(hipMemcpyAsync(node_ids, cur, (size_t)count * 4, hipMemcpyDefault, P->stream),
            EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(records, P->out0.ptr, (size_t)count * W * sizeof(double),
                           hipMemcpyDefault, P->stream), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(meta, P->out2.ptr, (size_t)count * 2 * sizeof(int32_t),
                           hipMemcpyDefault, P->stream), EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    H.n_lcss -= count;
    R.nf -= count;
    R.given += count;
    T->keep_ids = true;
    return EHM_OK;
}

static int hy_give(ehm_tree* T, int64_t count, const double* records, const int32_t* meta,
                   int32_t* first_id) {
    ehm_problem* P = T->prob;
    HyState& H = *T->hy;
    auto& R = T->run;
    if (first_id) *first_id = (int32_t)R.n_nodes;
    if (count == 0) return EHM_OK;
    if (R.n_nodes + count > T->limit)
        return fail(EHM_E_CAPACITY, "node pool exhausted at %lld nodes (max_nodes=%lld)",
                    R.n_nodes, T->limit);
    const int W = hy_record_doubles(P->dp.p, P->dp.n_u, H.nw);
    int rc;
    if ((rc = P->in0.ensure((size_t)count * W * sizeof(double)))) return rc;
    if ((rc = P->in1.ensure((size_t)count * 2 * sizeof(int32_t)))) return rc;
    DevBuf& fb = H.fr_lcss[H.cur];
    if ((long long)fb.cap < (H.n_lcss + count) * 4) {
        DevBuf bigger;
        if ((rc = bigger.ensure((size_t)(H.n_lcss + count) * 4 * 2))) return rc;
        if (H.n_lcss > 0)
            HIP_TRY(hipMemcpyAsync(bigger.ptr, fb.ptr, (size_t)H.n_lcss * 4,
                                   hipMemcpyDeviceToDevice, P->stream), EHM_E_HIP);
        HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
        std::swap(bigger, fb);
        bigger.release();
    }
    HIP_TRY(hipMemcpyAsync(P->in0.ptr, records, (size_t)count * W * sizeof(double),
                           hipMemcpyDefault, P->stream), EHM_E_HIP);
    HIP_TRY(hipMemcpyAsync(P->in1.ptr, meta, (size_t)count * 2 * sizeof(int32_t),
                           hipMemcpyDefault, P->stream), EHM_E_HIP);
    hipLaunchKernelGGL(hy_give_nodes, dim3((unsigned)count), dim3(64), 0, P->stream, T->dt,
                       (int)R.n_nodes, (int)count, H.nw, P->in0.as<double>(),
                       P->in1.as<int32_t>(), H.vf.as<hy_u64>(), H.cand.as<hy_u64>(),
                       H.neg.as<hy_u64>(), H.black.as<hy_u64>(), H.tneg.as<double>(),
                       fb.as<int32_t>(), (int)H.n_lcss);
    // the device-side allocation counter follows
    HyCtr* ctr = H.ctr.as<HyCtr>();
    const int n_new = (int)(R.n_nodes + count);
    HIP_TRY(hipMemcpyAsync(&ctr->n_nodes, &n_new, 4, hipMemcpyHostToDevice, P->stream), EHM_E_HIP);
    HIP_TRY(hipStreamSynchronize(P->stream), EHM_E_HIP);
    R.n_nodes += count;
    H.n_lcss += count;
    R.nf += count;
    R.received += count;
    T->keep_ids = true;
    return EHM_OK;
}
