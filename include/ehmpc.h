/*
 * ehmpc.h -- C-ABI of libehmpc.so, the MI355X (gfx950) implementation of the offline
 * parameter-space partitioning hot path of dmalyuta/explicit_hybrid_mpc.
 *
 * The reference is pure Python and has no FFI/plugin interface of its own
 * (SURVEY.md section 8b): its seam for this path is the Python `Oracle` object
 * (lib/oracle.py:18-474), the geometry helpers (lib/tools.py:134-257), the node types
 * (lib/tree.py:12-95) and the partition entry `alg_call(which_alg, branch, location)`
 * (lib/worker.py:180-185, 241-417).  Every entry point below states which of those it
 * replaces.  The Python package `explicit_hybrid_mpc_amd` binds this header with ctypes
 * and re-exposes the reference's Python signatures; INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions: plain C, every function returns 0 on success or a negative EHM_E_* code
 * (message via ehm_last_error(), thread-local).  All arrays are C-contiguous, caller
 * owned, float64 / int32 / uint8.  "host" entry points take host pointers and copy;
 * "_dev" entry points take device pointers valid on the problem's GPU and enqueue on
 * the problem's HIP stream without synchronising (use ehm_sync()).  A handle is bound to
 * one GPU and is not thread-safe (the reference's Oracle is not re-entrant either,
 * lib/oracle.py:271-280).  There is no CPU fallback: without a usable GPU
 * ehm_problem_create fails with EHM_E_NO_DEVICE.
 */
#ifndef EHMPC_H
#define EHMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libehmpc.so is built with -fvisibility=hidden: the entry points declared in this header (and
 * in the other two public headers) are its whole exported surface. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define EHM_OK              0
#define EHM_E_INVALID      -1   /* bad argument / unsupported dimension */
#define EHM_E_NO_DEVICE    -2   /* no HIP device, or HIP runtime error at setup */
#define EHM_E_HIP          -3   /* HIP runtime error during execution */
#define EHM_E_CAPACITY     -4   /* node pool exhausted (raise max_nodes) */
#define EHM_E_INFEASIBLE   -5   /* Theta contains infeasible regions (lib/worker.py:266) */
#define EHM_E_NUMERIC      -6   /* a vertex solve failed (lib/oracle.py:440-442) */

/* Limits of this build.  Up to 32 columns / 256 rows an LP is solved by one wavefront
 * (several LPs per workgroup around one LDS copy of the commutation's constant block);
 * wider LPs by one workgroup each, with the normal matrix formed on the matrix cores. */
#define EHM_MAX_N   64          /* LP columns incl. simplex weights and slack */
#define EHM_MAX_M   1024        /* LP rows   incl. simplex / cost rows */
#define EHM_MAX_P   8           /* parameter dimension */

/* Per-instance solve status (status arrays). */
#define EHM_ST_OPTIMAL  0       /* all relative optimality criteria met */
#define EHM_ST_STALLED  1       /* best iterate returned; merit > 1 */

/*
 * Canonical data of one MPC instance: for commutation index d (0..n_delta-1)
 *     J*(theta,d) = min_z c^T z   s.t.  G[d] z <= w[d] + S[d] theta ,   u0 = z[0:n_u].
 * Built on the host by explicit_hybrid_mpc_amd.mpc_library.PWAMPC.compile(); it carries
 * what the reference's Oracle.__init__ (lib/oracle.py:23-102) extracts from
 * mpc.make_constraints / mpc.cost.
 */
typedef struct ehm_problem_desc {
    int32_t n;          /* decision variables of P_theta_delta                       */
    int32_t m;          /* inequality rows                                           */
    int32_t p;          /* parameter dimension (= n_x; simplices have p+1 vertices)  */
    int32_t n_u;        /* inputs returned as u0                                     */
    int32_t n_delta;    /* admissible commutations                                   */
    int32_t delta_len;  /* length of one commutation vector (delta_size * N)         */
    const double*  G;       /* [n_delta][m][n] row-major                              */
    const double*  w;       /* [n_delta][m]                                           */
    const double*  S;       /* [n_delta][m][p]                                        */
    const double*  c;       /* [n]                                                    */
    const uint8_t* deltas;  /* [n_delta][delta_len] 0/1, enumeration order           */
    double eps_a;       /* absolute suboptimality tolerance (lib/oracle.py:38)       */
    double eps_r;       /* relative suboptimality tolerance (lib/oracle.py:39)       */
} ehm_problem_desc;

typedef struct ehm_problem ehm_problem;   /* opaque: device copies + stream + scratch */
typedef struct ehm_tree    ehm_tree;      /* opaque: grown partition, device resident */

/* Oracle.__init__ (lib/oracle.py:23-102). device = HIP ordinal. */
int ehm_problem_create(const ehm_problem_desc* desc, int device, ehm_problem** out);
int ehm_problem_destroy(ehm_problem* prob);
/* Quadratic part of the cost.  Every MPC law of the reference has a cvx.quad_form cost
 * (lib/mpc_library.py:180-183, :515-517): per commutation d
 *     V(z,theta) = c^T z + 1/2 z^T H[d] z + (f0[d] + F[d] theta)^T z
 *                  + 1/2 theta^T C[d] theta + c1[d]^T theta + c0[d]
 * H [n_delta][n][n], F [n_delta][n][p], f0 [n_delta][n], C [n_delta][p][p], c1 [n_delta][p],
 * c0 [n_delta]; H and C symmetric positive semidefinite (symmetrised on entry).  After this
 * call P_theta_delta is a convex QP, the suboptimality test (lib/oracle.py:89-97) a convex
 * QCQP with two quadratic rows, and every entry point below works on them (n+p+1 <= 32,
 * m+p+3 <= 256).  Both kernel generations carry the quadratic block; generation 2 is the
 * default, ehm_problem_set_solver(prob, 1) selects the one-wavefront kernels. */
int ehm_problem_set_quadratic(ehm_problem* prob, const double* H, const double* F,
                              const double* f0, const double* C, const double* c1,
                              const double* c0);
/* Re-set eps_a / eps_r (examples.create_oracle builds a second Oracle, lib/examples.py:43-46). */
int ehm_problem_set_eps(ehm_problem* prob, double eps_a, double eps_r);
/* Replaces the constant blocks of commutation slots [first, first + count) (G [count][m][n],
 * w [count][m], S [count][m][p], row-major like ehm_problem_desc) and rebuilds every image the
 * kernels read for them.  With it the commutation table is a cache of problems generated on the
 * fly: the prefix relaxations of a search over mode sequences -- what the reference leaves to the
 * MICP solver's branch-and-bound (lib/oracle.py:42-46, 89-102, lib/global_vars.py:25) --
 * explicit_hybrid_mpc_amd/sequences.py.  Not while a partition run is active. */
int ehm_problem_update_blocks(ehm_problem* prob, int32_t first, int32_t count, const double* G,
                              const double* w, const double* S);
/* Batched problems with the commutation given as a slot index of the table (not as a 0/1
 * vector): problems over a simplex (mode 0 = min, 1 = suboptimality-test slack, 2 = phase one;
 * the a5 / a7' problems of lib/oracle.py:74-79, 89-97) and P_theta_delta / its phase-one form
 * (lib/oracle.py:141-173). */
int ehm_simplex_idx_batch(ehm_problem* prob, int64_t n_inst, const double* R, const double* Vbar,
                          const int32_t* slot, int32_t mode, double* obj, double* alpha,
                          int32_t* status);
int ehm_point_idx_batch(ehm_problem* prob, int64_t n_inst, const double* theta,
                        const int32_t* slot, int32_t feas, double* J, double* u0,
                        int32_t* status);
/* Kernel generation used by this handle's launches: 2 (default) = one copy of the
 * commutation's constant LP block in LDS per workgroup, several wavefronts per workgroup;
 * 1 = one wavefront per workgroup with a private copy of the LP (first build, kept as an
 * on-device cross-check).  Environment override at create time: EHM_SOLVER=1|2. */
int ehm_problem_set_solver(ehm_problem* prob, int generation);
/* Named options: "solver" (1|2, as above); "decide_full" (0|1): by default the
 * suboptimality-test sweep of ehm_partition_run stops each LP as soon as the SIGN of its
 * optimum t* is certain (the reference's bar_E is a feasibility problem,
 * lib/oracle.py:285-309) and records a lower bound of |t*|; 1 = solve every test LP to full
 * accuracy (EHM_DECIDE_FULL=1 at create time does the same); "mid_first" (0|1, default 1): the
 * persistent frontier kernel solves a node's midpoint problem before its suboptimality test and
 * skips the test when the midpoint already proves the node open (42 % of the open nodes of the
 * bench tree; identical tree, tests/test_gpu_kernel_generations.py; EHM_MID_FIRST=0|1);
 * "inherit_witness" (0|1, default 1): a node the suboptimality-test LP finds open hands the
 * point that proved it (the LP iterate's parameter, with the cost of its decision vector as an
 * upper bound of the optimal cost there) to the child that contains it; the child is open
 * without an LP of its own while that point still beats its interpolated vertex costs by the
 * tolerance plus a safety margin (EHM_NO_WITNESS=1 disables);
 * "share_midpoints" (0|1, default 1): the persistent frontier kernel keeps a table of midpoint
 * optima in device memory -- the simplices around an edge all bisect it at the same point and,
 * with one commutation, solve the same problem there; the first to ask solves and publishes,
 * the others take the entry (csrc/ehm_midtable.h; identical tree, ehm_tree_info.midpoints_shared
 * counts the problems saved; EHM_NO_MIDTABLE=1 disables).  The same switch covers the table's
 * other uses: the level-synchronous wide kernels share their midpoint solves through it, the
 * persistent kernel looks up the midpoints of a node's OTHER edges as witnesses of openness
 * (ehm_tree_info.witness_table) and puts a node whose midpoint is being solved elsewhere back
 * into its queue instead of waiting, and the multi-commutation engine shares the results of its
 * point problems by (parameter, commutation, kind).  With the table on, WHICH valid witness a
 * node records (its own midpoint's, another edge's that a neighbour happened to have published,
 * an LP's) depends on the order wavefronts reach the table, so ehm_tree_info's solve and witness
 * counters, min_margin and the stored witnesses vary by a fraction of a percent between runs;
 * every verdict is taken with the same margin rule, and the tree (structure, vertices, vertex
 * costs, flags) does not vary.  "share_midpoints" = 0 gives run-to-run identical counters;
 * "any_admissible" (seed + 1, default 0 = off; multi-commutation handles): V_R and bar_D return a
 * draw among the ADMISSIBLE commutations -- feasible at every vertex; with t* >= 0 -- instead of
 * the canonical one (first in enumeration order; largest slack).  The reference poses both as
 * Minimize(0) (lib/oracle.py:201, 347): which admissible commutation comes back is its solver's
 * choice, and its published cwh_z tree sizes (lib/post_process.py:489, 526) carry that choice.
 * The draw is a hash of (seed, oracle, path code of the node), so a run repeats bit for bit and
 * the CPU oracle (oracle/oracle_cpu.py, rule 'hash') takes the same draws; values 2^30 + m are
 * deterministic extremes instead of draws (m bit 0: V_R returns the LAST commutation feasible at
 * every vertex, bit 1: bar_D the admissible commutation with the SMALLEST slack): the envelope of
 * what the choice can do to a tree (tools/cwh_jobs.py).  Not to be set while a run is active, and
 * ehm_partition_take / _give refuse to move nodes of a multi-commutation run under it (the
 * hand-over does not carry the path codes the draws read);
 * "budget_keep" (0|1, default 0): budgeted launches (ehm_partition_advance) let a wavefront keep
 * one child like unbudgeted ones.  Off by default: measured, the frontier such a launch leaves
 * costs the rebalancing rounds more than the kept children save.
 * "check_witness" (0|1, default 0): cross-check of the persistent kernel's two LP-free verdicts --
 * the tangent-plane bound is evaluated also for the nodes the inherited witness proves open; a
 * node it would close at the same time counts in ehm_tree_info.errors (a test option: it costs
 * the bound for every such node);
 * "work_first" (0|1, default 1): a wavefront of the persistent kernel that splits a node goes on
 * with one of the two children itself and queues the other (EHM_NO_WORKFIRST=1 disables);
 * "timing" (0|1, default 0): multi-commutation runs record an event pair and a counter snapshot
 * around every batched launch, so that ehm_tree_info carries kernel seconds and solves by problem
 * kind (bench.py sets it; ~25 extra stream commands per sweep otherwise spared). */
int ehm_problem_set_option(ehm_problem* prob, const char* name, double value);
/* Environment switches read by the library (experiments and A/B measurements; none is needed):
 *   EHM_SOLVER=1|2, EHM_DECIDE_FULL=1   at ehm_problem_create, as the options above;
 *   EHM_ENGINE=0|1     overrides ehm_run_opts.engine (1 = persistent frontier kernel);
 *   EHM_NO_KP=1        persistent kernel at one solver width even where a two-width instance
 *                      (ehm_kp.hip) is compiled;
 *   EHM_NO_CUTS=1      no vertex gradients: every leaf is closed by its suboptimality-test LP
 *                      (the tangent-plane bound is also off under decide_full = 1);
 *   EHM_KEEP_GOING=1   a run whose oracle solves failed returns its tree instead of
 *                      EHM_E_NUMERIC;  EHM_DUMP_FAIL=path  writes the failing instance there. */
int ehm_sync(ehm_problem* prob);
/* HIP stream the handle enqueues on (a hipStream_t), for event timing by the caller. */
void* ehm_stream(ehm_problem* prob);

/* ---- batched oracles (host buffers) ------------------------------------------------ */

/* Oracle.P_theta_delta(theta, delta) (lib/oracle.py:141-173) for n_inst instances.
 * delta: [n_inst][delta_len] 0/1.  Outputs J [n_inst], u0 [n_inst][n_u],
 * status/iters [n_inst] (may be NULL).  An infeasible instance reports EHM_ST_STALLED. */
int ehm_solve_ptd_batch(ehm_problem* prob, int64_t n_inst, const double* theta,
                        const uint8_t* delta, double* J, double* u0,
                        int32_t* status, int32_t* iters);

/* Oracle.P_theta_delta(theta, delta, check_feasibility=True) (lib/oracle.py:164-167):
 * feasible[k] = 1 iff the constraint set is non-empty; tau[k] (may be NULL) is the
 * optimal worst-row violation (<= 0 iff feasible). */
int ehm_feas_ptd_batch(ehm_problem* prob, int64_t n_inst, const double* theta,
                       const uint8_t* delta, uint8_t* feasible, double* tau);

/* Oracle.P_theta(theta) (lib/oracle.py:104-139): minimum over all commutations, lowest
 * index on ties.  delta_idx[k] = -1 and J = +inf when no commutation is feasible. */
int ehm_solve_pt_batch(ehm_problem* prob, int64_t n_inst, const double* theta,
                       double* J, double* u0, int32_t* delta_idx);

/* Oracle.V_R(R) (lib/oracle.py:175-218): first commutation feasible at every vertex of
 * R [n_inst][p+1][p]; delta_idx = -1 if none.  vJ [n_inst][p+1], vu0 [n_inst][p+1][n_u]. */
int ehm_vr_batch(ehm_problem* prob, int64_t n_inst, const double* R,
                 int32_t* delta_idx, double* vJ, double* vu0);

/* Slack of the epsilon-suboptimality test for ONE commutation per instance: the decision
 * form of Oracle.bar_E_delta_R's constraints (lib/oracle.py:89-97)
 *     t* = max t  s.t. MPC constraints at theta = sum alpha_i v_i,
 *                      sum alpha_i Vbar_i - V - eps_a >= t,  sum alpha_i Vbar_i - (1+eps_r) V >= t.
 * alpha [n_inst][p+1] (may be NULL) is the maximiser's simplex weights. */
int ehm_slack_batch(ehm_problem* prob, int64_t n_inst, const double* R, const double* Vbar,
                    const uint8_t* delta, double* tstar, double* alpha, int32_t* status);

/* Oracle.bar_E_delta_R(R, V_delta_R) (lib/oracle.py:285-309): closed[k] = 1 iff no
 * commutation has t* >= 0 (epsilon-suboptimal => close the leaf).  tbest = max_d t*(d). */
int ehm_bar_e_batch(ehm_problem* prob, int64_t n_inst, const double* R, const double* Vbar,
                    uint8_t* closed, double* tbest);

/* min over the simplex for a fixed commutation (lib/oracle.py:74-79, used by
 * in_variability_ball :276). */
int ehm_min_simplex_batch(ehm_problem* prob, int64_t n_inst, const double* R,
                          const uint8_t* delta, double* Jmin, int32_t* status);

/* Oracle.bar_D_delta_R(R, V_delta_R, delta_ref) incl. in_variability_ball
 * (lib/oracle.py:311-414, 220-283).  delta_idx = -1 when there is no better commutation
 * (the reference's (None,None,None,None)); otherwise theta_star [p], vJ [p+1],
 * vu0 [p+1][n_u], var_small. */
int ehm_bar_d_batch(ehm_problem* prob, int64_t n_inst, const double* R, const double* Vbar,
                    const uint8_t* delta_ref, int32_t* delta_idx, double* theta_star,
                    double* vJ, double* vu0, uint8_t* var_small);

/* Oracle.P_theta_delta(theta, d, check_feasibility=True) for EVERY commutation d:
 * feasible [n_inst][n_delta]. */
int ehm_feas_all_batch(ehm_problem* prob, int64_t n_inst, const double* theta, uint8_t* feasible);

/* One visit of Worker.lcss for a batch of nodes (lib/worker.py:367-401): bar_E_delta_R and, for
 * the nodes that stay open, bar_D_delta_R, sharing the slacks the two oracles have in common and
 * what the caller already knows:
 *   vfeas [n_inst][p+1][n_delta]  feasibility of every commutation at every vertex (a child
 *         inherits p of its p+1 vertices; ehm_feas_all_batch on the new midpoints only),
 *   cand  [n_inst][n_delta] or NULL  commutations feasible somewhere on the PARENT simplex.
 * Feasible at a vertex => feasible on the simplex; infeasible on the parent => infeasible on the
 * child; only the rest gets a phase-one problem.  Outputs: closed / tbest as ehm_bar_e_batch,
 * cand_out [n_inst][n_delta] (feasible on this simplex: the children's `cand`), and for open
 * nodes delta_idx / theta_star / vJ / vu0 / var_small as ehm_bar_d_batch (delta_idx = -1 for
 * closed nodes).  Same verdicts and optima as the two calls, ~2.5x fewer sub-problems. */
int ehm_lcss_batch(ehm_problem* prob, int64_t n_inst, const double* R, const double* Vbar,
                   const uint8_t* delta_ref, const uint8_t* vfeas, const uint8_t* cand,
                   uint8_t* closed, double* tbest, uint8_t* cand_out, int32_t* delta_idx,
                   double* theta_star, double* vJ, double* vu0, uint8_t* var_small);


/* ---- geometry ------------------------------------------------------------------------ */

/* tools.split_along_longest_edge (lib/tools.py:224-257), bit-exact incl. the first-max
 * tie rule.  R,S1,S2: [n][p+1][p]; ij: [n][2].  device = HIP ordinal. */
int ehm_split_batch(int device, int64_t n, int32_t p, const double* R,
                    double* S1, double* S2, int32_t* ij);
/* tools.simplex_volume (lib/tools.py:134-150). */
int ehm_volume_batch(int device, int64_t n, int32_t p, const double* R, double* vol);

/* ---- partition engine ---------------------------------------------------------------- */

typedef struct ehm_run_opts {
    int64_t max_nodes;      /* node pool capacity (0 = default)                          */
    int32_t max_depth;      /* stop splitting below this depth relative to roots (0 = none) */
    int32_t action;         /* 0 = 'ecc' then 'lcss' (lib/worker.py:241-291), 1 = 'lcss'   */
    int32_t engine;         /* 0 = level-synchronous sweeps (decide / scan / expand launches per
                             * tree level); 1 = persistent frontier kernel: ONE launch, every
                             * wavefront pops nodes from a device queue, tests them and pushes
                             * the children of the ones it splits -- used when the run goes to
                             * completion on one rank with the shared-block kernels, otherwise
                             * the sweeps run.  Node ids in device memory then follow the
                             * allocation order; ehm_tree_export relabels them breadth first,
                             * i.e. to the numbering of engine 0 (same tree either way). */
    /* Multi-GPU sharding of the live frontier: every rank grows the same top of the tree
     * until a sweep's frontier holds >= shard_min_frontier nodes, then keeps only the
     * frontier nodes whose position k satisfies k % shard_world == shard_rank (the rest are
     * flagged bit2 = "owned by another rank").  shard_world <= 1 disables it. */
    int32_t shard_rank;
    int32_t shard_world;
    int32_t skip_volume;    /* 1 = do not compute volume_closed in ehm_tree_info_get       */
    int64_t shard_min_frontier;
    /* > 0 (with shard_world > 1, engine 1): the whole run is ONE launch of the persistent
     * frontier kernel on every rank, from the roots; the tree above this depth is grown
     * identically everywhere (replicated), a node created AT this depth is pursued by rank
     * (path code % shard_world) only -- no sweeps, no host round trip, no collective.
     * 0 = deal the frontier of a sweep (shard_min_frontier), then one launch per share. */
    int32_t deal_depth;
    int32_t reserved0;
} ehm_run_opts;

/* Optional initial node data for action 1 ('lcss' roots already carry a commutation,
 * lib/scheduler.py:633-639). */
typedef struct ehm_node_init {
    const uint8_t* delta;   /* [n_roots][delta_len]   */
    const double*  vcost;   /* [n_roots][p+1]         */
    const double*  vinput;  /* [n_roots][p+1][n_u]    */
} ehm_node_init;

/* alg_call(which_alg, branch, location) (lib/worker.py:180-185) for a batch of root
 * simplices: grows every root until all leaves are epsilon-suboptimal.
 * root_vertices: [n_roots][p+1][p] host. */
int ehm_partition_run(ehm_problem* prob, int64_t n_roots, const double* root_vertices,
                      const ehm_node_init* init, const ehm_run_opts* opts, ehm_tree** out);

/* The same run in resumable pieces, for the multi-GPU driver (one process per GPU): every rank
 * runs a few sweeps, the ranks all-gather their frontier sizes (RCCL) and move node records
 * from the longest frontiers to the shortest.  The reference rebalances through its MPI
 * master/worker star (lib/scheduler.py:498-599); here only counts and the moved records travel.
 *   begin  : uploads the roots (and runs the 'ecc' vertex solves);
 *   step   : up to max_sweeps frontier sweeps (<= 0: until the frontier is empty);
 *            *frontier_size = live frontier nodes afterwards;
 *   take   : removes count frontier nodes -- every s-th entry of the frontier, s = size / count,
 *            a sample of all its depths (single-commutation runs; the newest count entries if
 *            s < 2, on multi-commutation runs, or with EHM_TAKE_NEWEST=1): node_ids [count], records
 *            [count][(p+1)p + (p+1) + (p+1)n_u] (vertices | vertex costs | vertex inputs),
 *            meta [count][2] = (commutation index, depth); the nodes get flag bit2;
 *   give   : appends count nodes produced by another rank's take to the pool (flag bit5)
 *            and to the live frontier; *first_id = node id of the first one;
 *   finish : totals for ehm_tree_info_get / ehm_tree_export (required before either).
 * The buffers of take (node_ids, records, meta) and give (records, meta) may be HOST or DEVICE
 * pointers (hipMemcpyDefault): the multi-GPU driver hands device-resident blocks -- torch tensors --
 * straight to ncclSend / ncclRecv, no staging through the host. */
int ehm_partition_begin(ehm_problem* prob, int64_t n_roots, const double* root_vertices,
                        const ehm_node_init* init, const ehm_run_opts* opts, ehm_tree** out);
int ehm_partition_step(ehm_tree* tree, int32_t max_sweeps, int64_t* frontier_size);
int ehm_partition_take(ehm_tree* tree, int64_t count, int32_t* node_ids, double* records,
                       int32_t* meta);
int ehm_partition_give(ehm_tree* tree, int64_t count, const double* records,
                       const int32_t* meta, int32_t* first_id);
int ehm_partition_finish(ehm_tree* tree);

/* Progress of a run in flight, between two ehm_partition_step calls: what the reference's
 * WorkerStatusPublisher accumulates per closed leaf (lib/worker.py:19-116, 374-375) -- volume
 * filled, simplex count -- read from the device node pool (one reduction kernel). */
typedef struct ehm_progress {
    int64_t n_nodes;        /* simplices generated so far (multi-GPU: the replicated top */
    int64_t n_closed;       /* of the tree is reported by rank 0 only); closed leaves    */
    int64_t frontier;       /* live frontier size                                       */
    int64_t sweeps;
    int64_t lp_solves;
    int64_t ipm_iters;
    int32_t depth;
    int32_t reserved;
    double  volume_closed;  /* sum of the closed leaves' volumes                        */
    int64_t n_splits;       /* splits performed by this rank: what the reference's status
                               publisher counts (simplex_count += 1 per split,
                               lib/worker.py:274,327,107-109)                           */
} ehm_progress;
/* Up to max_pops node visits of the persistent frontier kernel (<= 0: to completion); the
 * unprocessed part of its device queue -- a contiguous slice, the queue is consumed in order --
 * becomes the live frontier again, so ehm_partition_take / _give can move nodes between ranks
 * before the next call.  Single-commutation problems on the shared-block kernels.
 * Dynamic multi-GPU runs: ehm_run_opts.shard_min_frontier < 0 makes every rank but 0 start with
 * an EMPTY frontier (rank 0 owns the roots); the rebalancing rounds feed them. */
int ehm_partition_advance(ehm_tree* tree, int64_t max_pops, int64_t* frontier_size);
/* Pool occupancy of a run in progress, without touching the device: nodes allocated so far and
 * the capacity of this run (max_nodes).  The multi-GPU driver checks a receiver's free pool
 * before it plans a transfer (explicit_hybrid_mpc_amd/distributed.py). */
int ehm_partition_counts(const ehm_tree* tree, int64_t* n_nodes, int64_t* max_nodes);
/* What ehm_partition_take can hand over right now, and the width of one record in doubles:
 * single-commutation runs move any frontier node ([vertices | costs | inputs]); multi-commutation
 * runs move lcss nodes only, each with the engine's bit rows appended (feasibility per vertex,
 * candidates, inherited verdicts, blacklist: csrc/ehm_hybrid.h).  records of take / give are
 * [count][record_doubles]. */
int ehm_partition_movable(const ehm_tree* tree, int64_t* movable, int32_t* record_doubles);
int ehm_partition_progress(ehm_tree* tree, ehm_progress* out);

typedef struct ehm_tree_info {
    int64_t n_nodes;
    int64_t n_leaves;
    int64_t n_roots;
    int64_t n_closed;       /* leaves with is_epsilon_suboptimal                        */
    int64_t lp_solves;      /* LP sub-problems solved on the device                     */
    int64_t ref_solves;     /* reference-equivalent oracle calls (one MICP = 1)         */
    int64_t ipm_iters;      /* interior-point iterations summed over all solves         */
    int64_t sweeps;         /* frontier sweeps (levels)                                 */
    int32_t max_depth;
    int32_t truncated;      /* 1 if max_depth / capacity stopped the growth             */
    double  volume_closed;  /* sum of closed leaf volumes (lib/worker.py:374-375)       */
    double  min_margin;     /* min |t*| over all close/split decisions                  */
    double  device_seconds; /* GPU time of the whole run (HIP events on the stream)     */
    double  decide_seconds; /* summed duration of the suboptimality-test sweep kernel   */
    double  expand_seconds; /* summed duration of the split + midpoint-solve kernel     */
    int64_t decide_launches;
    int64_t expand_launches;
    int64_t decide_solves;  /* LPs solved by the decide kernel                          */
    int64_t decide_iters;   /* their interior-point iterations                          */
    /* multi-GPU sharding: work done before the frontier was dealt is identical on every rank */
    int64_t replicated_closed;
    int64_t replicated_nodes;
    int64_t replicated_solves;
    int64_t cert_closed;    /* leaves closed by the tangent-plane bound of t* (gradients of the
                               optimal cost at the vertices, from the multipliers of the vertex
                               solves) without solving their suboptimality-test LP            */
    int64_t witness_open;   /* nodes proved open by their midpoint solve (option "mid_first") */
    /* multi-commutation runs */
    int64_t swaps;          /* nodes that took a better commutation in place (lib/worker.py:396-401) */
    int64_t blacklisted;    /* commutations blacklisted after a failed vertex solve
                               (lib/oracle.py:198-218, 406-414) */
    /* solves / interior-point iterations by problem kind: [0] P_theta_delta, [1] its phase-one
     * form, [2] min over a simplex, [3] suboptimality test (slack), [4] phase one over a simplex.
     * Multi-commutation runs: decide_* / decide_seconds cover the simplex kinds (2-4),
     * expand_seconds the point kinds (0-1). */
    int64_t kind_solves[5];
    int64_t kind_iters[5];
    /* near-threshold routing: decisions whose |t*| is below 1e-6 (1 + |V_0|); each of them was
     * taken by the suboptimality-test problem solved to full accuracy, never by a shortcut
     * (sign-only stop, tangent-plane bound, midpoint witness, inherited verdict) */
    int64_t near_threshold;
    int64_t witness_inherited;  /* nodes proved open by the witness of an ancestor's
                                   suboptimality test (option "inherit_witness"), no LP     */
    int64_t midpoints_shared;   /* splits whose midpoint optimum another simplex around the same
                                   edge had solved already (option "share_midpoints"), no LP */
    /* persistent frontier kernel: where its wavefronts spent their time, in ticks of the 100 MHz
     * wall clock summed over all of them: [0] resident, [1] waiting for a queue slot to be filled
     * (starved), [2] waiting for a midpoint optimum another wavefront was solving, [3] in midpoint
     * solves, [4] in suboptimality-test solves; [5] = number of waits of kind [2]; [6] from the
     * pop of a node to its midpoint claim (record load, tangent-plane bound, inherited witness,
     * longest edge), [7] child records and queue pushes; [8] = nodes put back into the queue
     * because another wavefront was solving their midpoint; [9] unused */
    int64_t persist_ticks[10];
    int64_t witness_table;      /* nodes proved open by the optimum at the midpoint of one of their
                                   OTHER edges, left in the table of midpoint optima by a
                                   neighbouring simplex that had bisected that edge, no LP */
} ehm_tree_info;

int ehm_tree_info_get(const ehm_tree* tree, ehm_tree_info* out);
/* Flat export, node k: vertices [k][p+1][p], left[k] / right[k] child index or -1,
 * delta_idx[k] (-1 = none), vcost [k][p+1], vinput [k][p+1][n_u],
 * flags[k] bit0 = is_epsilon_suboptimal, bit1 = has commutation data,
 * bit2 = subtree owned by another rank (multi-GPU sharding / ehm_partition_take),
 * bit5 = node received from another rank (ehm_partition_give; it is a root of this part).
 * Nodes 0..n_roots-1 are the roots in input order.  Any pointer may be NULL. */
int ehm_tree_export(const ehm_tree* tree, double* vertices, int32_t* left, int32_t* right,
                    int32_t* delta_idx, double* vcost, double* vinput, uint8_t* flags,
                    double* tstar);
int ehm_tree_destroy(ehm_tree* tree);
/* Page-locked host memory for the arrays ehm_tree_export fills (the export then runs at the speed
 * of the host link; the branch is the worker's output, lib/worker.py:456-458).  Free with
 * ehm_host_free. */
int ehm_host_alloc(size_t bytes, void** out);
int ehm_host_free(void* ptr);


/* ---- batched evaluation of the explicit control law (the partition's consumer) --------- */

/* ExplicitMPC.setup + __call__ (lib/mpc_library.py:677-792) for batches of states.  The
 * partition comes as the flat export of ehm_tree_export (or any binary forest in that form):
 * nodes 0..n_roots-1 are the root simplices in tools.delaunay order (they hang off the
 * reference's right spine, lib/tools.py:152-189), left/right = child index or -1,
 * vertices [n_nodes][p+1][p], vinput [n_nodes][p+1][n_u].  create computes
 * inv([v1-v0 .. vp-v0]) for every node on the device (compute_simplex_basis_inverse, :685-712). */
typedef struct ehm_explicit ehm_explicit;
int ehm_explicit_create(int device, int64_t n_nodes, int32_t n_roots, int32_t p, int32_t n_u,
                        const int32_t* left, const int32_t* right, const double* vertices,
                        const double* vinput, ehm_explicit** out);
/* x [n][p] -> u [n][n_u] (barycentric interpolation in the containing leaf, :786-789);
 * leaf [n] = node id of that leaf, visited [n] = containment tests on the way (both may be
 * NULL); kernel_seconds (may be NULL) = device time of the evaluation kernel. */
int ehm_explicit_eval_batch(ehm_explicit* ex, int64_t n, const double* x, double* u,
                            int32_t* leaf, int32_t* visited, double* kernel_seconds);
int ehm_explicit_destroy(ehm_explicit* ex);
const char* ehm_explicit_last_error(void);

/* Cumulative counters of a problem handle (SURVEY.md section 5 "tracing"). */
typedef struct ehm_counters {
    int64_t lp_solves;
    int64_t ipm_iters;
    int64_t kernel_launches;
    int64_t stalled;
    int64_t fallbacks;      /* batch LPs the generation-2 kernels handed to generation 1 */
    int64_t slivers;        /* (simplex, commutation) pairs of the mixed-integer oracles whose
                               phase-one optimum is within 1e-7 of zero AND whose slack problem
                               found no interior: treated as infeasible on that simplex */
    /* batched oracles on the shared-block / wide kernels: device seconds of their kernels (HIP
     * events on the handle's stream around each launch) and launches; [0] problems at a point
     * (ehm_solve_ptd_batch, ehm_point_idx_batch, ...), [1] problems over a simplex */
    double  batch_seconds[2];
    int64_t batch_launches[2];
} ehm_counters;
int ehm_stats(ehm_problem* prob, ehm_counters* out);

/* sizeof of the public structs, in this order: ehm_problem_desc, ehm_run_opts, ehm_node_init,
 * ehm_progress, ehm_tree_info, ehm_counters.  A binding that mirrors the structs (the ctypes
 * classes of explicit_hybrid_mpc_amd/_capi.py) checks itself against the library it loaded: a
 * mirror that is too small would be overrun by the library's writes.  Returns the number of
 * sizes it knows; writes at most n of them.  Needs no device. */
int ehm_abi_sizes(int64_t* sizes, int32_t n);

/* Diagnostics of the shared-block solver (csrc/ehm_ipm2.h): shader-clock cycles spent in each
 * phase of the interior-point iteration, lane 0 of every wavefront, summed over all solves since
 * the handle was created; out[23] = the number of solves.  All zero unless the library was built
 * with -DEHM2_PROF=1 (an experimental build, tools/solver_phases.py).  No reference counterpart:
 * the reference times whole oracle calls (lib/worker.py:60-116). */
int ehm_solver_phase_ticks(ehm_problem* prob, int64_t out[24]);

/* What the shared-block solver eliminates from the Newton systems of this handle (csrc/ehm_ipm2.h,
 * DESIGN.md section 3.2b): out = { nd0, nE, LE, lda } -- the z-columns [nd0, nd0 + nE) are the
 * trailing range of which every MPC row holds at most one entry (the epigraph variables of the
 * infinity-norm cost, lib/mpc_library.py:530-560), LE = rows per such column (padded), lda = column
 * stride of the LDS image.  nE = 0: nothing is eliminated (no such range, a quadratic cost, or
 * EHM_SPARSE=0 in the environment when the handle was created).  Reporting only (bench.py prices
 * the executed flops with it); no reference counterpart. */
int ehm_problem_layout(ehm_problem* prob, int32_t out[4]);

/* Device self test of the wave-level primitives (DPP reductions, reciprocal) of every
 * compiled kernel instance: out[5*k .. 5*k+4] for instance k, expected
 * {1072, 99, 25, 1/3, -1}.  Test hook, not part of the reference's surface. */
int ehm_selftest(int device, double* out, int32_t max_instances, int32_t* n_instances);

const char* ehm_last_error(void);
const char* ehm_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* EHMPC_H */
