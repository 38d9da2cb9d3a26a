/*
 * ehm_search.h -- host-side bookkeeping of the searches over mode PREFIXES (C-ABI, part of
 * libehmpc.so; no GPU is touched by these entry points).
 *
 * The reference hands its mixed-integer oracles to a branch-and-bound solver
 * (lib/oracle.py:42-46, 89-102, lib/global_vars.py:25).  Here the search tree of that
 * branch-and-bound -- the tree of mode prefixes -- is walked by the caller, and the relaxations
 * are solved in batched launches of the table solvers (ehmpc.h: ehm_point_idx_batch,
 * ehm_simplex_idx_batch).  Between two launches the searches of thousands of partition nodes
 * advance in lockstep; what that costs on the host is bookkeeping, and this is it:
 *
 *   - parameter points get integer ids by value (the children of a partition node share all
 *     but one of its vertices);
 *   - phase-one verdicts are remembered per (prefix, point), a relaxation feasible at both ends
 *     of a bisected edge is feasible at its midpoint (its feasible parameters form a convex
 *     set), and the pairs nothing is known about are handed out ONCE per launch however many
 *     searches ask for them;
 *   - V_R's canonical answer (lib/oracle.py:175-218: any commutation feasible at every vertex;
 *     rule of this build: the first in enumeration order) is a lexicographic descent over
 *     prefixes; all descents of a round advance together and stop only where a launch is needed.
 *
 * Protocol of a round: the caller states the questions (ehm_search_query, or
 * ehm_search_descent_begin + ehm_search_descent_step), gets the pending pairs
 * (ehm_search_asks: distinct prefixes, one index per pair, the points' coordinates), solves
 * their phase-one problems in one launch and returns the verdicts (ehm_search_answer /
 * the next ehm_search_descent_step).
 *
 * A prefix d_0 .. d_{k-1} (k <= N, modes 0 .. n_modes-1) is the integer
 *     code = sum_i (d_i + 1) * (n_modes + 1)^i      (the empty prefix is 0);
 * (n_modes + 1)^N must stay below 2^26.  Every function returns 0 or a negative EHM_E_* code
 * of ehmpc.h (message: ehm_search_last_error(), thread-local).  A handle is not thread-safe.
 */
#ifndef EHM_SEARCH_H
#define EHM_SEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libehmpc.so is built with -fvisibility=hidden: the entry points declared in this header (and
 * in the other two public headers) are its whole exported surface. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct ehm_search ehm_search;

int ehm_search_create(int32_t p, int32_t n_modes, int32_t N, ehm_search** out);
int ehm_search_destroy(ehm_search* s);
const char* ehm_search_last_error(void);

/* Ids of n points ([n][p] float64), by value; new points get the next free id. */
int ehm_search_point_ids(ehm_search* s, int64_t n, const double* points, int64_t* ids);
/* mid[k] is the midpoint of a[k] and b[k] (the bisections of the partition,
 * lib/tools.py:191-257): a relaxation feasible at both ends needs no problem at the midpoint. */
int ehm_search_register_midpoints(ehm_search* s, int64_t n, const int64_t* mid, const int64_t* a,
                                  const int64_t* b);
/* Give up the pending launch (its solver failed): the pairs it was to decide are unknown again,
 * unfinished descents are dropped.  The handle is usable afterwards. */
int ehm_search_abandon(ehm_search* s);
/* Forget the verdicts (ids and midpoints stay valid). */
int ehm_search_forget(ehm_search* s);
/* counts[0] = verdicts held, [1] = points, [2] = pairs handed out so far, [3] = pairs that a
 * second search of the same launch asked for again (answered once). */
int ehm_search_counts(const ehm_search* s, int64_t counts[4]);

/* What is held about n (prefix, point) pairs: verdict[k] = 1 feasible (a held verdict, or both
 * ends of the bisected edge the point is the midpoint of), 0 infeasible, -1 nothing. */
int ehm_search_peek(ehm_search* s, int64_t n, const uint64_t* code, const int64_t* point_id,
                    int8_t* verdict);

/* n_sets questions "is the relaxation of prefix code[k] KNOWN to be feasible at one of the points
 * point_id[set_begin[k] .. set_begin[k+1])?" (then it is feasible on their convex hull): known[k]
 * = 1 at the first held verdict 1 -- the scan of the set stops there --, else 0 and
 * first_unknown[k] = position within the set of the first point nothing is held about (-1: every
 * point is known to be infeasible). */
int ehm_search_peek_any(ehm_search* s, int64_t n_sets, const uint64_t* code, const int64_t* set_begin,
                        const int64_t* point_id, uint8_t* known, int32_t* first_unknown);

/* n_sets questions "is the relaxation of prefix code[k] feasible at EVERY point
 * point_id[set_begin[k] .. set_begin[k+1])?".  flags[k] = 0 where a held verdict already says
 * no; the pairs that need a problem are pending afterwards (*n_ask of them over *n_prefix
 * distinct prefixes; 0 = flags are final). */
int ehm_search_query(ehm_search* s, int64_t n_sets, const uint64_t* code, const int64_t* set_begin,
                     const int64_t* point_id, uint8_t* flags, int64_t* n_ask, int64_t* n_prefix);
/* The pending pairs: prefix_code [n_prefix], prefix_index [n_ask] (into prefix_code),
 * theta [n_ask][p]. */
int ehm_search_asks(const ehm_search* s, uint64_t* prefix_code, int64_t* prefix_index,
                    double* theta);
/* Verdicts of the pending pairs of ehm_search_query (feasible[a] != 0: the phase-one optimum
 * is within tolerance); they are remembered and flags [n_sets] completed. */
int ehm_search_answer(ehm_search* s, const uint8_t* feasible, uint8_t* flags);

/* V_R for n point sets at once: for set j the first full sequence, in enumeration order, whose
 * relaxations -- prefix by prefix -- are feasible at every point of the set and which is not
 * one of excluded[excl_begin[j] .. excl_begin[j+1]) (codes of full sequences: the reference's
 * blacklist, lib/oracle.py:198; excl_begin may be NULL). */
int ehm_search_descent_begin(ehm_search* s, int64_t n, const int64_t* set_begin,
                             const int64_t* point_id, const int64_t* excl_begin,
                             const uint64_t* excluded);
/* Advances every descent as far as held verdicts carry it.  feasible = the verdicts of the pairs
 * the previous step left pending (NULL on the first call).  *n_ask == 0: all descents are done. */
int ehm_search_descent_step(ehm_search* s, const uint8_t* feasible, int64_t* n_ask,
                            int64_t* n_prefix);
/* sequence [n][N] (modes; -1 where none exists), steps = lockstep levels walked. */
int ehm_search_descent_result(ehm_search* s, int32_t* sequence, int64_t* steps);

/* ---- best-first queues of the suboptimality-test searches (bar_E for many nodes at once) ------
 * Per search: a queue of prefixes keyed by the upper bound t of their relaxation's slack (the
 * reference's bar_E in decision form, lib/oracle.py:89-97, 285-309).  A step pops the best `width`
 * prefixes of every running search and expands them; children whose value is known (a value
 * seeded by the caller; an inherited upper bound, kept only where it refutes: t < -guard) need no
 * problem, the rest is the ask list of ONE launch.  With the answers: t < 0 refutes a child, a
 * full sequence with t >= 0 proves the node open, any other child is queued; an empty queue
 * closes the node.  Expansion order, ask order and ties are those of heapq on (-t, prefix).
 * An answer holding a NaN is refused (EHM_E_INVALID, the step stays in flight): an unknown value
 * must never prune. */
typedef struct ehm_search_bare ehm_search_bare;
int ehm_search_bare_create(int32_t n, int32_t n_modes, int32_t N, const double* guard,
                           ehm_search_bare** out);
int ehm_search_bare_destroy(ehm_search_bare* b);
int ehm_search_bare_seed(ehm_search_bare* b, int32_t j, uint64_t code, double t, int32_t open);
int ehm_search_bare_bounds(ehm_search_bare* b, int32_t j, int64_t count, const uint64_t* code,
                           const double* tb);
int ehm_search_bare_step(ehm_search_bare* b, int32_t width, int64_t* n_ask, int64_t* n_active);
int ehm_search_bare_asks(const ehm_search_bare* b, uint64_t* code, int32_t* owner);
int ehm_search_bare_answer(ehm_search_bare* b, const double* t, int64_t* n_active);
int ehm_search_bare_result(const ehm_search_bare* b, int8_t* closed, double* margin,
                           int64_t counts[2]);
int ehm_search_bare_learned(const ehm_search_bare* b, int32_t j, int64_t* count, uint64_t* code,
                            double* t);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
