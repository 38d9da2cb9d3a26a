// Device-side data structures shared by every kernel generation of libehmpc
// (ehm_kernels.h: one wavefront per workgroup; ehm_k2.hip: shared constant matrix,
// several wavefronts per workgroup).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ehm {

// Problem constants, read-only, L2 resident.  Stored column-major per commutation so that
// the LDS fill of one LP column is one coalesced stream over the rows.
struct DevProblem {
    int n, m, p, n_u, n_delta;
    const double* Gt;   // [n_delta][n][m]
    const double* St;   // [n_delta][p][m]
    const double* w;    // [n_delta][m]
    const double* c;    // [n]
    double eps_a, eps_r;
    // second-generation kernels (ehm_k2.hip): the LDS image of the constant part of every
    // LP of commutation d, column-major with column stride lda2 (odd), columns
    //   [ G (n) | -S (p) | -1 | 0 ]            (ncw2 = n + p + 2)
    // so that  G z - S psi <= w + S R0  with  psi = theta - R0  (ehm_ipm2.h).
    const double* Wc2;  // [n_delta][ncw2][lda2]
    int lda2, ncw2;
    // shared-block kernels (ehm_ipm2.h): the LDS image WITHOUT the eliminated z-columns
    // [nd0, n) -- the trailing range of which every MPC row holds at most one entry (the epigraph
    // variables of an infinity-norm cost); nd0 = n where there is no such range (and always for
    // quadratic costs).  Per commutation, column stride lda4:
    //   [ G_D (nd0) | -S (p) | -1 | 0 | aE ]   (ncw4 = nd0 + p + 3; aE[i] = the entry of row i in
    //                                           its eliminated column, 0 where it has none)
    // followed by  eval[nE][LE4] (doubles), erow[nE][LE4], eidx[m] (int32): the rows of every
    // eliminated column (0-padded) and the eliminated column of every row.  tot4 = doubles per
    // commutation (even).
    const double* Wc4;  // [n_delta][tot4]
    int lda4, ncw4, tot4;
    int nd0, LE4;
    int wc_lds;         // shared-block kernels: 1 = w and c of the commutation sit behind Wc in LDS,
                        // 0 = they are read from device memory (set per launch: K2Launch::wc_lds)
    // wide kernels (ehm_k3.hip, LPs with more than 32 columns): the same block row-major,
    // padded to 64 columns and to mpad3 rows (multiple of 64) with zeros -- operand layout of
    // v_mfma_f64_16x16x4_f64 and of the lane-per-column products (ehm_ipm3.h).  Null when
    // every LP of the problem fits the wave-local kernels.
    const double* Wr3;  // [n_delta][mpad3][64]
    int mpad3;
    // quadratic part of the cost (null Hq: linear cost only), per commutation
    //   V(z,theta) = c'z + 1/2 z'H z + (f0 + F theta)'z + 1/2 theta'C theta + c1'theta + c0
    // (ehm_problem_set_quadratic; handled by the generation-1 kernels, ehm_ipm.h)
    const double* Hq;   // [n_delta][n][n]
    const double* Fq;   // [n_delta][p][n]   column q of F as an n-vector
    const double* f0q;  // [n_delta][n]
    const double* Cq;   // [n_delta][p][p]
    const double* c1q;  // [n_delta][p]
    const double* c0q;  // [n_delta]
};

}  // namespace ehm
#include "ehm_midtable.h"
namespace ehm {

// Node pool of the partition tree (structure of arrays of fixed-size records).
//   rec[k] = [ vertices (p+1)*p | vertex_costs (p+1) | vertex_inputs (p+1)*n_u ]  (doubles)
// which is the payload of the reference's NodeData (lib/tree.py:31-39).
struct DevTree {
    double*  rec;
    int32_t* left;      // left child index (right = left+1), -1 for a leaf
    int32_t* didx;      // commutation index, -1 = none yet
    int32_t* depth;
    uint8_t* flags;     // bit0 closed (is_epsilon_suboptimal), bit1 has commutation data
    double*  tstar;     // slack of the last close/split decision
    int rec_stride;     // doubles per record (multiple of 8)
    int p, n_u;
    // grad[k][(p+1)*p]: gradient of the optimal cost V*_delta at every vertex of node k (from the
    // multipliers of the vertex / midpoint solves: dJ/dtheta = -S^T lambda), NaN = unknown.
    // Null unless the run maintains them (shared-block kernels, linear cost).  They give the
    // cutting-plane bound that closes most leaves without a suboptimality-test LP (cut_bound).
    double*  grad;
    // path code of every node (root r: r; children: 2 code, 2 code + 1; modulo 2^32) -- kept by
    // sharded runs of the persistent frontier kernel, which deal the nodes of one tree depth over
    // the ranks by it (code % world: the last turns of the path, so the descendants of one
    // ancestor spread over all ranks).  Null otherwise.
    uint32_t* code;
    // wit[k][p+2]: a point of node k at which it is known NOT to be epsilon-suboptimal yet --
    // [0] an upper bound c_w of the optimal cost at theta_w, [1..p+1] the barycentric weights
    // of theta_w in the node's vertices.  It comes from the suboptimality-test LP of an
    // ancestor (the iterate that proved that ancestor open) and travels down to the child that
    // contains it -- its sibling gets the point's projection onto the shared face
    // (witness_for_children) --: where  min(Vbar(theta_w) - c_w - eps_a, Vbar(theta_w) - (1+eps_r) c_w)  is
    // still positive with the CHILD's vertex costs, the child is open without an LP of its own
    // (persistent frontier kernel).  c_w includes the safety amount
    // EHM_WIT_REL x (slack proved at the ancestor).  All zero = no witness (the test then gives
    // min(-eps_a, 0)).  Null unless the run keeps them.
    double*  wit;
    // table of midpoint optima shared by the wavefronts of the persistent frontier kernel
    // (ehm_midtable.h); state == nullptr unless the run keeps one
    MidTable mt;
};

// The iterate a witness comes from is not exactly feasible: the solver accepted it when its
// residuals, scaled by 1 + |objective|, were below 1e-3 of the slack it proves
// (EHM2_SIGN_RES_REL).  The stored cost bound c_w is therefore RAISED by this fraction of that
// slack -- 20 times the solver's own criterion -- and the witness stops counting once the
// child's interpolated cost has come that close to it.
#define EHM_WIT_REL 0.02

// Children's witnesses (DevTree::wit) from the parent's: wit = [c_w, alpha_0..alpha_p] in the
// parent's vertices, Vc its vertex costs, (bi, bj) the split edge; lane < p + 2 computes entry
// `lane` of child 0 (vertex bi -> midpoint) in v0 and of child 1 (vertex bj -> midpoint) in v1.
// The child that contains theta_w keeps it (weights 2 a_far at the midpoint, a_near - a_far at
// the vertex it keeps); the other one gets theta_w slid towards ITS vertex of the edge until the
// shared face (weights equal), mu = (a_near - a_far) / (1 + a_near - a_far), with the cost bound
// (1 - mu) c_w + mu V_far: there the midpoint carries 2 (1 - mu) a_near and the kept vertex 0.
__device__ __forceinline__ void witness_for_children(const double* wit, const double* Vc, int bi,
                                                     int bj, int lane, double& v0, double& v1) {
    const double ai = wit[1 + bi], aj = wit[1 + bj];
    const bool to0 = aj >= ai;              // child 0 holds theta_w
    const double an = to0 ? aj : ai, af = to0 ? ai : aj;
    const int far = to0 ? bi : bj;          // the other child's own vertex of the edge
    const int near = to0 ? bj : bi;
    const double mu = (an - af) / (1.0 + an - af);
    double vh = wit[lane], vo = (1.0 - mu) * wit[lane];
    if (lane == 0) vo = fma(mu, Vc[far], vo);
    if (lane == 1 + far) {                  // holder: the midpoint; other: its kept vertex
        vh = 2.0 * af;
        vo = 0.0;
    }
    if (lane == 1 + near) {                 // holder: its kept vertex; other: the midpoint
        vh = an - af;
        vo = 2.0 * (1.0 - mu) * an;
    }
    v0 = to0 ? vh : vo;
    v1 = to0 ? vo : vh;
}

// Upper bound of the suboptimality-test optimum t* from the tangent planes of the convex optimal
// cost at the vertices: V*(theta) >= L_i(theta) = V_i + g_i.(theta - v_i), so with the 2(p+1)
// functions, LINEAR in the barycentric weights alpha,
//     f_{2i}   = Vbar - L_i - eps_a ,      f_{2i+1} = Vbar - (1 + eps_r) L_i ,
// t* <= max_alpha min_r f_r(alpha).  Two relaxations of that max-min, both in closed form:
//   * one function at a time: max_alpha f_r sits at a vertex;
//   * two at a time: max_alpha min(f_a, f_b) sits at a vertex or where an EDGE of the simplex
//     crosses the plane f_a = f_b (the pieces of a concave piecewise-linear function are
//     polytopes whose corners are those points),
// and t* is at most the smallest of all of them.  Measured on closed leaves (CPU oracle, HiGHS
// duals): the first closes 59 %, the second 93 %, the exact LP over all cuts 95.5 % -- and none
// ever closes an open node, a negative bound means t* < 0.
// node = [vertices | vertex costs ...] in LDS, g = the node's (p+1)*p gradients in global memory
// (NaN = unknown: every function it enters becomes NaN and is skipped), scr = 2(p+1)^2 doubles of
// LDS scratch.  Every lane returns the bound (+1e300 when nothing is known).
// thr: a caller that only asks "is the bound below thr?" (the leaf is then closed) gets the
// one-function bound back as soon as that one answers yes -- 59 % of the closed leaves -- and the
// 45-pair pass runs for the rest.
__device__ inline double cut_bound(const double* node, const double* g, int p, double eps_a,
                                   double eps_r, int lane, double* scr, double thr = -1e301) {
    const int na = p + 1, nr = 2 * na;
    const double* V = node + na * p;
    for (int k = lane; k < na * na; k += 64) {      // f_r at the vertices
        const int i = k / na, j = k - i * na;
        double d = 0.0;
        for (int q = 0; q < p; ++q) d = fma(g[i * p + q], node[j * p + q] - node[i * p + q], d);
        const double Li = V[i] + d;
        scr[(2 * i) * na + j] = V[j] - Li - eps_a;
        scr[(2 * i + 1) * na + j] = V[j] - (1.0 + eps_r) * Li;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double b = 1e300;
    if (lane < nr) {                                // one function at a time
        double m = -1e300;
        bool ok = true;
        for (int j = 0; j < na; ++j) {
            const double f = scr[lane * na + j];
            ok = ok && (f == f);
            m = fmax(m, f);
        }
        if (ok) b = m;
    }
    if (thr > -1e300) {
        double b1 = b;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) b1 = fmin(b1, __shfl_xor(b1, o, 64));    // nr <= 18 lanes
        const double b1a = __shfl(b1, 0, 64), b1b = __shfl(b1, 16, 64);
        b1 = fmin(b1a, b1b);
        if (b1 < thr) return b1;
    }
    const int npairs = nr * (nr - 1) / 2;
    for (int t = lane; t < npairs; t += 64) {       // two at a time
        int a = 0, rem = t;                         // t -> (a < c)
        while (rem >= nr - 1 - a) {
            rem -= nr - 1 - a;
            ++a;
        }
        const int c = a + 1 + rem;
        const double* fa = scr + a * na;
        const double* fc = scr + c * na;
        double m = -1e300;
        bool ok = true;
        for (int j = 0; j < na; ++j) {
            ok = ok && (fa[j] == fa[j]) && (fc[j] == fc[j]);
            m = fmax(m, fmin(fa[j], fc[j]));
        }
        for (int u = 0; u < na; ++u)
            for (int v = u + 1; v < na; ++v) {
                const double du = fa[u] - fc[u], dv = fa[v] - fc[v];
                if (du * dv < 0.0) {                // the edge (u, v) crosses f_a = f_c
                    // (v_rcp_f64 + two Newton steps, not an IEEE division: the crossing point
                    // enters a bound that is compared with a 1e-6 margin, a few ulp are nothing)
                    const double den = du - dv;
                    double rc = __builtin_amdgcn_rcp(den);
                    rc = fma(rc, fma(-den, rc, 1.0), rc);
                    rc = fma(rc, fma(-den, rc, 1.0), rc);
                    const double sx = du * rc;
                    m = fmax(m, fma(sx, fa[v] - fa[u], fa[u]));
                }
            }
        if (ok) b = fmin(b, m);
    }
    // wave minimum by DPP row shifts / row broadcasts (the reduction of ehm_ipm2.h: 20 instructions
    // where six ds_bpermute round trips took 60 and their LDS latency)
#define EHM_DPP_MIN_STEP(CTRL, RMASK)                                                          \
    {                                                                                          \
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(1e300), __double2loint(b),   \
                                                   CTRL, RMASK, 0xf, false);                   \
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(1e300), __double2hiint(b),   \
                                                   CTRL, RMASK, 0xf, false);                   \
        b = fmin(b, __hiloint2double(hi, lo));                                                 \
    }
    EHM_DPP_MIN_STEP(0x111, 0xf)
    EHM_DPP_MIN_STEP(0x112, 0xf)
    EHM_DPP_MIN_STEP(0x114, 0xf)
    EHM_DPP_MIN_STEP(0x118, 0xf)
    EHM_DPP_MIN_STEP(0x142, 0xa)
    EHM_DPP_MIN_STEP(0x143, 0xc)
#undef EHM_DPP_MIN_STEP
    const int lo = __builtin_amdgcn_readlane(__double2loint(b), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(b), 63);
    return __hiloint2double(hi, lo);
}

struct DevCounters {
    unsigned long long lp_solves;
    unsigned long long ipm_iters;
    unsigned long long stalled;
    unsigned long long min_margin_bits;   // |t*| as ordered uint64
    unsigned long long errors;
    unsigned long long slack_solves;      // LPs over a simplex (decide sweep)
    unsigned long long slack_iters;
    unsigned long long cert_closed;       // leaves closed by the cutting-plane bound, no LP
    unsigned long long wit_open;          // nodes proved open by their midpoint solve, no LP
    unsigned long long routed;            // decisions with |t*| < EHM_ROUTE_TOL (full-accuracy LP)
    unsigned long long wit_inherited;     // nodes proved open by an ancestor's witness, no LP
    unsigned long long mid_shared;        // midpoint optima taken from the table, no LP
    unsigned long long wit_table;         // nodes proved open by ANOTHER edge's midpoint optimum
                                          // found in the table (a neighbour had bisected it)
    // persistent frontier kernel, 100 MHz wall-clock ticks summed over the wavefronts: [0] resident,
    // [1] waiting for a queue slot to be filled, [2] waiting for a midpoint another wavefront is
    // solving, [3] inside midpoint solves, [4] inside suboptimality-test solves; [5] = number of
    // waits of kind [2]
    // [6] from the pop to the midpoint claim (record load, tangent-plane bound, inherited witness,
    // longest edge), [7] child records + queue pushes, [8] nodes put back because their midpoint
    // was being solved elsewhere
    unsigned long long prof[10];
    // builds with -DEHM2_PROF=1 only (ehm_ipm2.h): shader-clock cycles per phase of ipm_solve,
    // lane 0 of every wavefront, summed over the solves that ran to their end; [23] = their number
    unsigned long long phase[24];
    unsigned int ticket;                  // work distribution of the wide sweep kernels: next
    unsigned int ticket_pad;              // frontier position (zeroed before every launch)
};

__host__ __device__ inline int rec_off_vcost(int p) { return (p + 1) * p; }
__host__ __device__ inline int rec_off_vinput(int p) { return (p + 1) * p + (p + 1); }
__host__ __device__ inline int rec_doubles(int p, int n_u) {
    return (p + 1) * p + (p + 1) + (p + 1) * n_u;
}

enum { LP_POINT = 0, LP_FEAS = 1, LP_MIN_SIMPLEX = 2, LP_SLACK = 3, LP_FEAS_SIMPLEX = 4 };
enum { SX_MIN = 0, SX_SLACK = 1, SX_FEAS = 2 };

__host__ __device__ inline int lp_cols(const DevProblem& P, int kind) {
    switch (kind) {
        case LP_POINT: return P.n;
        case LP_FEAS: return P.n + 1;
        case LP_MIN_SIMPLEX: return P.n + P.p;
        case LP_FEAS_SIMPLEX: return P.n + P.p + 1;
        default: return P.n + P.p + 1;
    }
}
__host__ __device__ inline int lp_rows(const DevProblem& P, int kind) {
    switch (kind) {
        case LP_POINT: return P.m;
        case LP_FEAS: return P.m + 1;
        case LP_MIN_SIMPLEX: return P.m + P.p + 1;
        case LP_FEAS_SIMPLEX: return P.m + P.p + 2;
        default: return P.m + P.p + 3;
    }
}

// ---------------------------------------------------------------------------------------
// geometry (bit-exact restatement of lib/tools.py:224-257 arithmetic)
// ---------------------------------------------------------------------------------------
// Edge length = sqrt(fma-chain of squared differences): numpy evaluates la.norm(x) as
// sqrt(x.dot(x)) and OpenBLAS' ddot tail loop is a fused-multiply-add chain starting from
// 0 (oracle/csrc/geom_ref.c).  First maximal edge in itertools.combinations order wins.
__device__ inline void longest_edge(const double* R, int p, int& bi, int& bj) {
#pragma clang fp contract(off)
    double best = -1.0;
    bi = 0;
    bj = 1;
    for (int i = 0; i <= p; ++i)
        for (int j = i + 1; j <= p; ++j) {
            double s = 0.0;
            for (int k = 0; k < p; ++k) {
                const double df = R[i * p + k] - R[j * p + k];
                s = __fma_rn(df, df, s);
            }
            const double len = __dsqrt_rn(s);
            if (len > best) {
                best = len;
                bi = i;
                bj = j;
            }
        }
}

// The same decision with one edge per lane (every lane of a 64-wide wavefront calls it; p <= 8,
// so the (p+1) p / 2 <= 36 edges fit): identical arithmetic per edge, and the FIRST maximal edge in
// enumeration order is the lowest lane whose length equals the wave maximum.  The serial version
// costs every lane ten correctly rounded square roots at p = 4; this one costs each lane one.
__device__ inline void longest_edge_wave(const double* R, int p, int lane, int& bi, int& bj) {
#pragma clang fp contract(off)
    const int n_edges = (p + 1) * p / 2;
    int ea = 0, rem = (lane < n_edges) ? lane : 0;
    while (rem >= p - ea) {
        rem -= (p - ea);
        ++ea;
    }
    const int eb = ea + 1 + rem;
    double s = 0.0;
    for (int k = 0; k < p; ++k) {
        const double df = R[ea * p + k] - R[eb * p + k];
        s = __fma_rn(df, df, s);
    }
    const double len = (lane < n_edges) ? __dsqrt_rn(s) : -1.0;
    double mx = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(len == mx);
    const int src = hit ? __builtin_ctzll(hit) : 0;     // (NaN vertices: edge 0, like the serial loop)
    bi = __shfl(ea, src, 64);
    bj = __shfl(eb, src, 64);
}

// Status word of a batched solve: 0 = converged / accepted.  Otherwise 1 | (q << 8), q = the decade
// of the best merit the solve reached (merit <= 10^q; merit 1 = the tolerances of 1e-10): a solve
// that stalled at merit 10^q holds its optimum to about 10^(q - 10), relative.  Callers that need
// a SIGN, or that accept what the reference accepts as OPTIMAL_INACCURATE
// (lib/oracle.py:440-442), read the decade; everybody else sees "nonzero = not converged".
__device__ inline int ehm_status_word(int status, double merit) {
    if (status == 0) return 0;
    int q = 0;
    if (!(merit == merit)) {
        q = 99;
    } else {
        double m = merit;
        while (m > 1.0 && q < 99) { m *= 0.1; ++q; }
    }
    return 1 | (q << 8);
}

}  // namespace ehm
