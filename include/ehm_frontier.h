/*
 * ehm_frontier.h -- the partition driver for laws whose mode sequences cannot be enumerated
 * (BASELINE.json configs[4]: 4 modes, N = 8 -> 65 536 sequences), native (C-ABI, part of
 * libehmpc.so).
 *
 * The reference leaves its mixed-integer oracles to a branch-and-bound solver
 * (lib/oracle.py:42-46, 89-102, 347-350) and drives them node by node from Python
 * (Worker.ecc / Worker.lcss, lib/worker.py:241-417).  Here the whole round loop runs behind ONE
 * call: the pending cells of the tree are visited together, every step of every cell's search --
 * one level of V_R's lexicographic descent over mode prefixes, one expansion of bar_E's
 * best-first queue -- goes into the same batched launch of the table kernels
 * (ehm_point_idx_batch / ehm_simplex_idx_batch), the relaxation blocks of the prefixes are
 * condensed from the law's per-step matrices here (no interpreter, no numpy in the loop), and
 * the tree stays in flat arrays until it is exported.
 *
 *   ecc  (lib/worker.py:241-291)  V_R = first sequence in enumeration order feasible at every
 *        vertex (ehm_search.h descents) -> vertex optima -> the cell holds a commutation;
 *        none: feasibility at the barycentre (lib/worker.py:264-266), longest-edge bisection
 *        (ehm_split_batch), two ecc children
 *   lcss (lib/worker.py:293-417)  bar_E by best-first search over prefixes (ehm_search_bare_*;
 *        the parent's best-slack sequence is tried first): closed -> an epsilon-suboptimal leaf.
 *        Open: bar_D = the sequence of LARGEST slack among those feasible at every vertex
 *        (best-first for the value, a lexicographic walk for the first sequence that attains it),
 *        its vertex optima and the variability test (lib/oracle.py:220-283); the cell adopts it
 *        in place and is looked at again (lib/worker.py:396-401) or is bisected, the children
 *        inheriting the vertex costs with the midpoint's optimum in the new slot
 *        (lib/worker.py:356-365).
 *   Only a cell whose vertex solves FAIL (lib/oracle.py:214-218, 406-414: blacklist and retry)
 *   is handed back to the caller with its record (flag EHM_FR_OPEN); the caller's one-cell
 *   oracles (explicit_hybrid_mpc_amd/bnb.py) have that path.
 *
 * Every function returns 0 or a negative EHM_E_* code of ehmpc.h; message:
 * ehm_frontier_last_error() (thread-local).  A handle is not thread-safe.
 */
#ifndef EHM_FRONTIER_H
#define EHM_FRONTIER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libehmpc.so is built with -fvisibility=hidden: the entry points declared in this header (and
 * in the other two public headers) are its whole exported surface. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct ehm_frontier ehm_frontier;
struct ehm_problem;

/* The piecewise-affine law with infinity-norm stage costs (the pattern of
 * lib/mpc_library.py:521-552 with the commutation fixed per step):
 *   x_{k+1} = A_i x_k + B_i u_k + w_i  in mode i,  Hx_i x_k <= hx_i,
 *   Gx x_k <= gx (k = 1..N),  Gu u_k <= gu,  V = sum ||Q x_k||_inf + sum ||R u_k||_inf. */
typedef struct ehm_pwa_law {
    int32_t n_x, n_u, n_modes, N;
    const double* A;              /* [n_modes][n_x][n_x] */
    const double* B;              /* [n_modes][n_x][n_u] */
    const double* w;              /* [n_modes][n_x]      */
    const int32_t* region_rows;   /* [n_modes] rows of Hx_i (0 = the mode has no region) */
    const double* Hx;             /* [sum region_rows][n_x] */
    const double* hx;             /* [sum region_rows]      */
    int32_t n_gx; const double* Gx; const double* gx;
    int32_t n_gu; const double* Gu; const double* gu;
    int32_t n_q;  const double* Q;      /* [n_q][n_x] */
    int32_t n_r;  const double* R;      /* [n_r][n_u] */
} ehm_pwa_law;

/* The batched solvers the driver stands on, over (prefix code, point / simplex) PAIRS
 * (prefix codes: ehm_search.h).  The device form is built by ehm_frontier_create; a caller may
 * bring its own (tests: the CPU statement of the table, oracle/prefix_bb.py). */
typedef struct ehm_pair_solvers {
    void* user;
    /* J [n] (+inf: infeasible; with feasibility_only 0 / +inf), u0 [n][n_u].  known_feasible:
     * the caller holds a proof that every pair is feasible, phase one is skipped. */
    int (*points)(void* user, int64_t n, const uint64_t* code, const double* theta,
                  int32_t feasibility_only, int32_t known_feasible, double* J, double* u0);
    /* t [n]: optimum of the suboptimality test of the prefix's relaxation on simplex R[k] with
     * vertex costs Vbar[k] (-inf: infeasible on it; +inf: a relaxation whose solve stalled --
     * "no information").  known[k]: feasible somewhere on the simplex, no phase one. */
    int (*slack)(void* user, int64_t n, const uint64_t* code, const double* R, const double* Vbar,
                 const uint8_t* known, double* t, double* alpha /* [n][p+1] maximiser, or NULL */);
    /* J [n]: minimum over simplex R[k] of the optimal cost of the prefix's relaxation (+inf:
     * infeasible on it; -inf: a solve that stalled -- as a bound it prunes nothing). */
    int (*min)(void* user, int64_t n, const uint64_t* code, const double* R, const uint8_t* known,
               double* J);
    /* tools.split_along_longest_edge for a batch (ehm_split_batch). */
    int (*split)(void* user, int64_t n, const double* R, double* S1, double* S2, int32_t* ij);
} ehm_pair_solvers;

/* Device form: n_tables commutation tables on GPU `device`, one per horizon (ascending, the last
 * one N).  The relaxation of a prefix constrains and prices its own steps only, so as a block of
 * the law with a SHORTER horizon it is the same problem in fewer columns and rows
 * (PWAMPC.with_horizon; it requires u = 0 to be admissible, which the caller checks): table t holds
 * the prefixes of horizons[t-1]+1 .. horizons[t] steps -- short ones on the shared-block kernels,
 * the rest on the wide kernels at THEIR size -- and full sequences live in the last one.
 * slots[t]: blocks resident in table t, written on demand, the longest-unused ones replaced when
 * it is full (0 = one slot per prefix, nothing is ever replaced). */
int ehm_frontier_create(const ehm_pwa_law* law, int32_t n_tables, const int32_t* horizons,
                        const int32_t* slots, int device, double eps_a, double eps_r,
                        ehm_frontier** out);
/* The same driver on the caller's solvers (n_x, n_u, n_modes, N of the law). */
int ehm_frontier_create_custom(int32_t n_x, int32_t n_u, int32_t n_modes, int32_t N,
                               const ehm_pair_solvers* solvers, double eps_a, double eps_r,
                               ehm_frontier** out);
int ehm_frontier_destroy(ehm_frontier* f);
const char* ehm_frontier_last_error(void);
int ehm_frontier_set_eps(ehm_frontier* f, double eps_a, double eps_r);
/* Device table `index` (for ehm_stats), its horizon, slots and blocks replaced so far; index ==
 * number of tables returns *table = NULL. */
int ehm_frontier_table(ehm_frontier* f, int32_t index, struct ehm_problem** table, int32_t* horizon,
                       int32_t* slots, int64_t* evicted);

/* Drops the tree and everything the searches remember (point ids, phase-one verdicts, vertex
 * optima); the blocks loaded in the device tables are problem data and stay. */
int ehm_frontier_reset(ehm_frontier* f);
/* A root cell (p+1 vertices, [p+1][p]) to grow with 'ecc'. */
int ehm_frontier_add_root(ehm_frontier* f, const double* vertices);

typedef struct ehm_frontier_opts {
    int32_t round_cap;        /* cells visited together in one round (0 = 4096)                */
    int32_t launch_target;    /* problems a best-first step aims at per launch (0 = 65536)     */
    int64_t max_visits;       /* stop after that many cell visits (0 = none): open cells stay  */
    int64_t min_regions;      /* stop once that many leaves are closed (0 = run to completion) */
    int32_t speculate;        /* reserved (0)                                                  */
    int32_t max_depth;        /* cells at this depth (roots: 0) are not bisected: they stay
                               * open leaves flagged EHM_FR_DEPTH (0 = no limit).  A law whose
                               * optimal cost jumps across a mode boundary is refined without
                               * end along that boundary (so is the reference's partition)      */
} ehm_frontier_opts;

typedef struct ehm_frontier_stats {
    int64_t rounds, visits, ecc_visits, lcss_visits;
    int64_t regions;          /* closed leaves                                                 */
    int64_t open_cells;       /* cells handed back to the caller (EHM_FR_OPEN)                 */
    int64_t n_nodes;
    int64_t calls_v_r, calls_p_theta, calls_bar_e, calls_bar_d;
    int64_t depth_limited;    /* open leaves left at max_depth                                 */
    int64_t swaps;            /* cells that adopted bar_D's commutation in place (lib/worker.py:396-401) */
    int64_t witness_hits;
    int64_t prefixes_expanded, answered_without_a_problem;
    int64_t optima_asked, optima_solved;
    int64_t lp_solves;        /* device form: problems the tables solved                       */
    int64_t launches;         /* solver calls (each at most a few kernel launches)             */
    int64_t blocks_loaded;
    int64_t stalled, slivers;
    int32_t truncated;        /* stopped by max_visits / min_regions with work pending         */
    int32_t depth;
    double seconds_solvers;   /* wall time inside the pair solvers                             */
    double seconds_total;
} ehm_frontier_stats;

/* Grows every pending cell (the roots added since the last reset, or what an earlier call left
 * pending) until nothing is pending or a limit of `opts` is reached.
 * A call that FAILS inside a round (a solver error, EHM_E_NUMERIC, out of memory) leaves the handle
 * poisoned: the cells of that round are in no work list any more, so run / p_theta / add_root /
 * export answer EHM_E_INVALID ("reset first") until ehm_frontier_reset -- never a silently
 * incomplete tree.
 * Environment EHM_FR_TALLY=1: the slack problems asked so far by caller (incumbent seeds, the
 * suboptimality test's search, bar_D's search) go to stderr at the end of every call -- diagnostics
 * (tools/c5_tally.py). */
int ehm_frontier_run(ehm_frontier* f, const ehm_frontier_opts* opts, ehm_frontier_stats* stats);

/* Oracle.P_theta (lib/oracle.py:104-139) at n parameters ([n][p]) in lockstep: J [n] (+inf: no mode
 * sequence is feasible there), u0 [n][n_u], sequence [n][N] (modes; -1 where infeasible) -- the
 * canonical answer, the first sequence in enumeration order within the tie tolerance of the
 * optimum.  What examples.create_oracle's eps_a rule (lib/examples.py:42-46) and a batched
 * ImplicitMPC (lib/mpc_library.py:659) call.  u0 / sequence may be NULL. */
int ehm_frontier_p_theta(ehm_frontier* f, int64_t n, const double* theta, double* J, double* u0,
                         int32_t* sequence);

/* Take / give of pending cells between handles (the reference's task queue hands any leaf to any
 * worker, lib/scheduler.py:498-599, 633-639; here a handle that runs dry -- its rank's roots are
 * finished -- is fed from one that still holds work, e.g. one configs[4] root that costs minutes).
 *
 * ehm_frontier_pending: cells in this handle's work lists (after a run that max_visits /
 * min_regions truncated, or before the first run).
 * ehm_frontier_take: removes up to max_cells of them, the SHALLOWEST first (the largest sub-trees
 * still to grow; a fixed order, so a repeated run hands over the same cells), and writes their
 * records: node [max_cells] (index in this handle's tree, where the cell stays a leaf flagged
 * EHM_FR_REMOTE), vertices [.][p+1][p], sequence [.][N] (-1: the cell holds no commutation yet),
 * vertex_costs [.][p+1], vertex_inputs [.][p+1][n_u], depth [.].
 * ehm_frontier_give: the cells become ROOTS of this handle's forest (only before it has grown:
 * after create / reset / add_root), with their depths, so ehm_frontier_opts.max_depth cuts the same
 * cells in either handle; a cell with a sequence goes on with lcss from its record, one without
 * starts with ecc.  The taker's tree, grafted onto the giver's EHM_FR_REMOTE leaves in the order
 * of `node`, is the tree one handle would have grown (tests/test_gpu_frontier_native.py). */
int ehm_frontier_pending(const ehm_frontier* f, int64_t* n_pending);
int ehm_frontier_take(ehm_frontier* f, int64_t max_cells, int64_t* n_taken, int32_t* node,
                      double* vertices, int32_t* sequence, double* vertex_costs,
                      double* vertex_inputs, int32_t* depth);
int ehm_frontier_give(ehm_frontier* f, int64_t n, const double* vertices, const int32_t* sequence,
                      const double* vertex_costs, const double* vertex_inputs, const int32_t* depth);

/* Node flags of the export. */
#define EHM_FR_CLOSED     1   /* epsilon-suboptimal leaf                                       */
#define EHM_FR_HAS_RECORD 2   /* holds a commutation, vertex costs and vertex inputs           */
#define EHM_FR_OPEN       4   /* leaf handed back: bar_E left it open (lcss continues with the
                               * caller), or a vertex solve failed (lib/oracle.py:214-218)     */
#define EHM_FR_PENDING    8   /* leaf not visited yet (a truncated run)                        */
#define EHM_FR_NEEDS_ECC 16   /* with EHM_FR_OPEN: the cell has no record, the caller runs ecc */
#define EHM_FR_DEPTH     32   /* open leaf: not bisected, the run's depth limit                */
#define EHM_FR_REMOTE    64   /* leaf taken out by ehm_frontier_take: another handle grows it  */

int ehm_frontier_sizes(const ehm_frontier* f, int64_t* n_nodes, int64_t* n_roots);
/* vertices [n][p+1][p], left / right [n] (-1: leaf; the roots are nodes 0 .. n_roots-1, children
 * follow their parents), sequence [n][N] (modes, -1 without a record), vertex_costs [n][p+1],
 * vertex_inputs [n][p+1][n_u], flags [n]. */
int ehm_frontier_export(const ehm_frontier* f, double* vertices, int32_t* left, int32_t* right,
                        int32_t* sequence, double* vertex_costs, double* vertex_inputs,
                        uint8_t* flags);
/* Problems solved by (table, kind, prefix length): out [n_tables][5][N+1]; kinds
 * 0 point phase one, 1 point optimum, 2 simplex phase one, 3 minimum over a simplex, 4 slack. */
int ehm_frontier_lp_counts(const ehm_frontier* f, int64_t* out);
/* The relaxation block of a prefix as the device tables hold it (tests: against
 * PWAMPC.condense_prefix): horizon = N or short_len; G [m][n], w [m], S [m][n_x];
 * dims[0] = n, dims[1] = m are returned when G is NULL. */
int ehm_frontier_condense(const ehm_pwa_law* law, int32_t horizon, int32_t len,
                          const int32_t* prefix, int32_t dims[2], double* G, double* w, double* S);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
