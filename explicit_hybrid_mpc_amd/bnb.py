"""
Mixed-integer oracles by branch-and-bound over mode PREFIXES, for instances whose commutations
cannot be enumerated (BASELINE.json configs[4]: 4 modes, N = 8 -> 65 536 sequences), and the
partition driver that uses them.

The reference hands P_theta, V_R, bar_E_delta_R and bar_D_delta_R to a mixed-integer solver
(lib/oracle.py:42-102, lib/global_vars.py:25: MOSEK's branch-and-bound).  Here the tree of that
search is the tree of mode prefixes; the relaxation of a prefix (``PWAMPC.condense_prefix``:
the rows and stage costs of the undecided steps dropped) is one more block of a device
commutation table (``sequences.PrefixTable``), so every node expansion is a handful of LPs in
one batched launch of the kernels the enumerating engine uses.  Bounds:

    optimal cost of a prefix  <=  optimal cost of each completion          (P_theta)
    slack t* of a prefix      >=  slack t* of each completion              (bar_E, bar_D)
    prefix infeasible at a point / on a simplex  =>  so is each completion (V_R, bar_D)

The answers are the CANONICAL ones of the enumerating oracles (``oracle.Oracle``, DESIGN.md
"canonical commutation rule"): first in enumeration order among optima within the tie
tolerance.  Each search therefore runs in two phases: best-first for the optimal VALUE, then a
lexicographic descent for the first sequence that attains it -- a plateau of tied sequences
(modes that are interchangeable in an overlap band) costs one dive, not its enumeration.

``grow`` runs the reference's ecc / lcss (lib/worker.py:241-417) with these oracles from the top
of the tree and hands every node whose region table fits the device engine
(``sequences.relevant_sequences``: at most 256 sequences can matter on it) to
``ehm_partition_run`` -- the subtree below it is grown without leaving the GPU.
"""

import heapq
import time
import numpy as np

from . import sequences
from .oracle import SolverError
from .sequences import TIE_TOL
from .tree import NodeData

PLATEAU = 1e-7           # values closer than this (relative) to the incumbent are not explored
                         # in phase one: a tenth of the tie tolerance
BATCH = 16               # prefixes expanded per launch in the best-first phases


def _rel(x):
    return 1. + abs(x)


class PrefixOracle:
    """
    The reference's ``Oracle`` surface (lib/oracle.py:18-474; same methods and return
    conventions as ``oracle.Oracle``) for an ``mpc`` with any number of mode sequences.
    ``table``: a ``sequences.PrefixTable`` (created if not given).
    """

    def __init__(self, mpc, eps_a, eps_r, slots=4096, device=0, table=None, split=None):
        self.mpc = mpc
        self.eps_a = eps_a
        self.eps_r = eps_r
        self._own = table is None
        self.table = table if table is not None else sequences.make_table(
            mpc, slots=slots, device=device, eps_a=eps_a, eps_r=eps_r, split=split)
        self.table.set_eps(eps_a, eps_r)
        self.last_margin = np.inf
        self.n_expanded = 0          # prefixes expanded (search-tree nodes)
        self.n_inherited = 0         # prefixes answered from what a node or its ancestors solved
        self.n_blacklisted = 0
        self.calls = dict(P_theta=0, V_R=0, bar_E=0, bar_D=0)

    def close(self):
        if self._own:
            self.table.close()

    # -- sequences <-> the reference's 0/1 vectors -------------------------------------------
    def delta_of(self, seq):
        return self.mpc.sequence_to_delta(seq)

    def sequence_of(self, delta):
        d = np.asarray(delta).astype(int).reshape(self.mpc.N, self.mpc.delta_size)
        if not np.all(d.sum(axis=1) == 1):
            raise ValueError('not an admissible commutation')
        return tuple(int(i) for i in d.argmax(axis=1))

    def _kids(self, prefixes):
        return [q + (i,) for q in prefixes for i in range(self.mpc.delta_size)]

    # -- lib/oracle.py:104-139 ---------------------------------------------------------------
    def P_theta(self, theta, check_feasibility=False):
        t0 = time.time()
        self.calls['P_theta'] += 1
        theta = np.asarray(theta, dtype=np.float64)
        if check_feasibility:
            return self.table.first_feasible(theta[None]) is not None
        N = self.mpc.N

        def bound(prefixes):
            J, u0 = self.table.solve_points(prefixes, np.tile(theta, (len(prefixes), 1)))
            return J, u0
        # phase one: the optimal value, best first on the relaxations' costs
        heap, best = [(0., ())], np.inf
        cut = lambda: best - PLATEAU * _rel(best) if np.isfinite(best) else np.inf
        while heap and heap[0][0] < cut():
            batch = []
            while heap and len(batch) < BATCH and heap[0][0] < cut():
                batch.append(heapq.heappop(heap)[1])
            kids = self._kids(batch)
            self.n_expanded += len(batch)
            J, _ = bound(kids)
            for q, j in zip(kids, J):
                if not np.isfinite(j):
                    continue
                if len(q) == N:
                    best = min(best, j)
                else:
                    heapq.heappush(heap, (j, q))
        if not np.isfinite(best):
            return None, None, None, time.time() - t0
        # phase two: the first sequence, in enumeration order, within the tie tolerance
        limit = best + TIE_TOL * _rel(best)
        stack = [()]
        while stack:
            q = stack.pop()
            kids = self._kids([q])
            self.n_expanded += 1
            J, u0 = bound(kids)
            good = [(k, j, u) for k, j, u in zip(kids, J, u0) if j <= limit]
            if good and len(good[0][0]) == N:
                k, j, u = good[0]
                return u.copy(), self.delta_of(k), float(j), time.time() - t0
            stack.extend(k for k, _, _ in reversed(good))
        raise SolverError('P_theta: the optimum found in phase one was not reproduced')

    # -- lib/oracle.py:141-173 ---------------------------------------------------------------
    def P_theta_delta(self, theta, delta, check_feasibility=False):
        t0 = time.time()
        seq = self.sequence_of(delta)
        J, u0 = self.table.solve_points([seq], np.asarray(theta, dtype=np.float64)[None],
                                        feasibility_only=check_feasibility)
        if check_feasibility:
            return bool(np.isfinite(J[0]))
        if not np.isfinite(J[0]):
            return None, None, time.time() - t0
        return u0[0].copy(), float(J[0]), time.time() - t0

    def _vertex_solves(self, R, seq):
        """lib/oracle.py:416-443: optimal input and cost of ``seq`` at every vertex."""
        J, u0 = self.table.solve_points([seq] * R.shape[0], R)
        if not np.all(np.isfinite(J)):
            raise SolverError('problem infeasible')
        return [(u0[i].copy(), float(J[i]), 0.) for i in range(R.shape[0])]

    # -- lib/oracle.py:175-218 ---------------------------------------------------------------
    def V_R(self, R):
        self.calls['V_R'] += 1
        R = np.asarray(R, dtype=np.float64)
        blacklist = set()
        while True:
            seq = self.table.first_feasible(R, exclude=blacklist)
            if seq is None:
                return None, None
            try:
                return self.delta_of(seq), self._vertex_solves(R, seq)
            except SolverError:                     # lib/oracle.py:214-218
                blacklist.add(seq)
                self.n_blacklisted += 1

    # -- lib/oracle.py:285-309 ---------------------------------------------------------------
    def _slack(self, prefixes, R, V):
        n = len(prefixes)
        return self.table.solve_slack(prefixes, np.tile(R, (n, 1, 1)), np.tile(V, (n, 1)))

    def bar_E_delta_R(self, R, V_delta_R):
        """True iff NO sequence has t* >= 0 (the feasibility problem of the reference is
        infeasible): best-first on the prefixes' slack bounds until a full sequence with
        t* >= 0 turns up or every prefix is refuted."""
        self.calls['bar_E'] += 1
        R = np.asarray(R, dtype=np.float64)
        V = np.asarray(V_delta_R, dtype=np.float64)
        N = self.mpc.N
        heap = [(-np.inf, ())]
        refuted = np.inf                            # smallest |bound| among refuted prefixes
        while heap:
            batch = [heapq.heappop(heap)[1] for _ in range(min(BATCH, len(heap)))]
            kids = self._kids(batch)
            self.n_expanded += len(batch)
            t, _ = self._slack(kids, R, V)
            for q, tq in zip(kids, t):
                if not tq >= 0.:
                    refuted = min(refuted, abs(tq))
                elif len(q) == N:
                    self.last_margin = abs(tq)      # a lower bound of the best slack
                    return False
                else:
                    heapq.heappush(heap, (-tq, q))
        self.last_margin = refuted
        return True

    # -- lib/oracle.py:220-283 ---------------------------------------------------------------
    def in_variability_ball(self, R, V_delta_R, delta_ref, delta_star, theta_star):
        R = np.asarray(R, dtype=np.float64)
        Jmin = self.table.solve_min([self.sequence_of(delta_ref)], R[None], exact=True)[0]
        J = self.table.solve_points([self.sequence_of(delta_star)],
                                    np.asarray(theta_star, dtype=np.float64)[None])[0][0]
        if not (np.isfinite(Jmin) and np.isfinite(J)):
            raise SolverError('problem infeasible')
        rhs = max(self.eps_a, self.eps_r * float(J))
        return bool(np.max(V_delta_R) - float(Jmin) < rhs)

    # -- lib/oracle.py:311-414 ---------------------------------------------------------------
    def bar_D_delta_R(self, R, V_delta_R, delta_ref):
        """
        The commutation with the LARGEST slack among those feasible at every vertex with
        t* >= 0 (canonical rule; ties: first in enumeration order); the four ``None`` when
        there is none or it is ``delta_ref``.
        """
        self.calls['bar_D'] += 1
        R = np.asarray(R, dtype=np.float64)
        V = np.asarray(V_delta_R, dtype=np.float64)
        N = self.mpc.N
        ref = self.sequence_of(delta_ref)
        blacklist = set()

        def bound(prefixes):
            """slack bound, or -inf where the prefix cannot be feasible at every vertex"""
            t, alpha = self._slack(prefixes, R, V)
            live = [k for k in range(len(prefixes)) if t[k] >= 0.
                    and prefixes[k] not in blacklist]
            ok = self.table.feasible_at_all([prefixes[k] for k in live], R)
            out = np.full(len(prefixes), -np.inf)
            for k, good in zip(live, ok):
                if good:
                    out[k] = t[k]
            return out, alpha
        while True:
            # phase one: the largest slack
            heap, best = [(-np.inf, ())], -np.inf
            floor = lambda: max(0., best + PLATEAU * _rel(best)) if np.isfinite(best) else 0.
            while heap and -heap[0][0] >= floor():
                batch = []
                while heap and len(batch) < BATCH and -heap[0][0] >= floor():
                    batch.append(heapq.heappop(heap)[1])
                kids = self._kids(batch)
                self.n_expanded += len(batch)
                t, _ = bound(kids)
                for q, tq in zip(kids, t):
                    if not tq >= 0.:
                        continue
                    if len(q) == N:
                        best = max(best, tq)
                    else:
                        heapq.heappush(heap, (-tq, q))
            if not np.isfinite(best):
                return None, None, None, None
            # phase two: the first sequence within the tie tolerance of it
            limit = max(0., best - TIE_TOL * _rel(best))
            stack, star = [()], None
            while stack and star is None:
                q = stack.pop()
                kids = self._kids([q])
                self.n_expanded += 1
                t, alpha = bound(kids)
                good = [(k, tk, a) for k, tk, a in zip(kids, t, alpha) if tk >= limit]
                if good and len(good[0][0]) == N:
                    star = good[0]
                else:
                    stack.extend(k for k, _, _ in reversed(good))
            if star is None:
                raise SolverError('bar_D: the slack found in phase one was not reproduced')
            seq, _, alpha = star
            if seq == ref:
                return None, None, None, None
            theta_star = alpha @ R
            delta_star = self.delta_of(seq)
            try:
                vx = self._vertex_solves(R, seq)
                small = self.in_variability_ball(R, V, delta_ref, delta_star, theta_star)
                return delta_star, theta_star, vx, small
            except SolverError:                     # lib/oracle.py:406-414
                blacklist.add(seq)
                self.n_blacklisted += 1


# ---------------------------------------------------------------------------------------------
# the partition driver
# ---------------------------------------------------------------------------------------------
def _set_record(data, delta, vx):
    data.commutation = delta
    data.vertex_costs = np.array([v[1] for v in vx])
    data.vertex_inputs = np.array([v[0] for v in vx])


def grow(oracle, branch, action='ecc', table_max=256, max_visits=None, handoff=True,
         engine_opts=None, log=None, split=None):
    """
    ecc / lcss (lib/worker.py:241-417) on ``branch`` (a ``tree.Tree`` whose data holds the
    simplex; for 'lcss' also commutation, vertex costs and vertex inputs), grown in place with
    the branch-and-bound oracles; a node whose region table has at most ``table_max`` sequences
    is handed to the device engine with that table (``handoff``).  Returns a dict of counts.
    ``max_visits``: stop after that many host visits (the tree is then incomplete -- the open
    nodes are leaves without ``is_epsilon_suboptimal``).  ``split``: the longest-edge bisection
    (default ``tools.split_along_longest_edge``, the device kernel every engine here uses).
    """
    from . import engine, partition, tools
    split_longest_edge = split or tools.split_along_longest_edge
    mpc = oracle.mpc
    stats = dict(host_visits=0, handoffs=0, handoff_nodes=0, handoff_leaves=0, table_sizes=[],
                 tables_too_large=0, no_incumbent=0, truncated=False)
    work = [(branch, action)]
    while work:
        if max_visits is not None and stats['host_visits'] >= max_visits:
            stats['truncated'] = True
            break
        node, act = work.pop()
        data = node.data
        R = np.asarray(data.vertices, dtype=np.float64)
        if handoff:
            extra = [oracle.sequence_of(data.commutation)] if act == 'lcss' else []
            try:
                seqs, info = sequences.relevant_sequences(mpc, R[None], max_sequences=table_max,
                                                          table=oracle.table, extra=extra)
                if len(seqs) > table_max:
                    raise sequences.TableTooLarge('with the node\'s commutation')
            except sequences.TableTooLarge:
                stats['tables_too_large'] += 1
                seqs = None
            except sequences.NoIncumbent:
                stats['no_incumbent'] += 1
                seqs = None
            if seqs is not None:
                sub = mpc.restrict(seqs)
                gp = engine.GpuProblem(sub.compile(), oracle.eps_a, oracle.eps_r,
                                       device=getattr(oracle.table, 'device', 0))
                try:
                    init = None
                    if act == 'lcss':
                        init = dict(delta=np.asarray(data.commutation, dtype=np.float64)[None],
                                    vertex_costs=np.asarray(data.vertex_costs)[None],
                                    vertex_inputs=np.asarray(data.vertex_inputs)[None])
                    flat = gp.partition(R[None], action=act, init=init, **(engine_opts or {}))
                finally:
                    gp.close()
                partition.graft_flat(flat, [node])
                stats['handoffs'] += 1
                stats['handoff_nodes'] += flat.n_nodes
                stats['handoff_leaves'] += int(flat.info['n_leaves'])
                stats['table_sizes'].append(len(seqs))
                if log:
                    log('handoff: %d sequences, %d nodes' % (len(seqs), flat.n_nodes))
                continue
        stats['host_visits'] += 1
        if act == 'ecc':                            # lib/worker.py:241-283
            delta_hat, vx = oracle.V_R(R)
            # lib/worker.py:264-266 checks the barycentre first; a sequence feasible at every
            # vertex is feasible there too, so only a cell V_R finds nothing for needs the check
            if delta_hat is None and not oracle.P_theta(theta=np.average(R, axis=0),
                                                        check_feasibility=True):
                raise RuntimeError('STOP, Theta contains infeasible regions')
            if delta_hat is None:
                S_1, S_2, v_idx = split_longest_edge(R)
                oracle.table.register_midpoints([S_1[v_idx[0]]], [R[v_idx[0]]], [R[v_idx[1]]])
                node.grow(NodeData(vertices=S_1), NodeData(vertices=S_2))
                work.append((node.right, 'ecc'))
                work.append((node.left, 'ecc'))
            else:
                _set_record(data, delta_hat, vx)
                work.append((node, 'lcss'))
            continue
        # lcss, lib/worker.py:340-417
        if oracle.bar_E_delta_R(R=R, V_delta_R=data.vertex_costs):
            data.is_epsilon_suboptimal = True
            continue
        delta_star, theta_star, new_vx, varies_little = oracle.bar_D_delta_R(
            R=R, V_delta_R=data.vertex_costs, delta_ref=data.commutation)
        feasible = delta_star is not None
        if feasible:
            new_costs = np.array([v[1] for v in new_vx])
            new_inputs = np.array([v[0] for v in new_vx])
        else:
            delta_star = data.commutation
            new_costs, new_inputs = data.vertex_costs, data.vertex_inputs
        if feasible and varies_little:
            data.commutation, data.vertex_costs, data.vertex_inputs = (delta_star, new_costs,
                                                                        new_inputs)
            work.append((node, 'lcss'))
            continue
        S_1, S_2, v_idx = split_longest_edge(R)
        v_mid = S_1[v_idx[0]]
        oracle.table.register_midpoints([v_mid], [R[v_idx[0]]], [R[v_idx[1]]])
        u_mid, V_mid = oracle.P_theta_delta(theta=v_mid, delta=delta_star)[:2]
        in_1, in_2 = new_inputs.copy(), new_inputs.copy()
        co_1, co_2 = new_costs.copy(), new_costs.copy()
        in_1[v_idx[0]] = u_mid
        in_2[v_idx[1]] = u_mid
        co_1[v_idx[0]] = V_mid
        co_2[v_idx[1]] = V_mid
        node.grow(NodeData(vertices=S_1, commutation=delta_star, vertex_costs=co_1,
                           vertex_inputs=in_1),
                  NodeData(vertices=S_2, commutation=delta_star, vertex_costs=co_2,
                           vertex_inputs=in_2))
        work.append((node.right, 'lcss'))
        work.append((node.left, 'lcss'))
    return stats
