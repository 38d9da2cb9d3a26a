"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

Branch-and-bound over mode PREFIXES for the mixed-integer oracles -- what the reference hands to
MOSEK's branch-and-bound (lib/oracle.py:42-46, 89-102, lib/global_vars.py:25) and what
BASELINE.json's configs[4] (4 modes, N = 8: 65 536 sequences) needs instead of enumeration.
DESIGN.md section 7c states the bounds; this file restates them on the uncondensed models and
tests/test_oracle_prefix_bounds.py checks them:

    for a prefix (d_0 .. d_{k-1}) drop the dynamics and the mode-region rows of the steps >= k.
    The later states are then free variables, the later stage costs can be made 0, so the
    relaxed problem's optimal cost is a LOWER bound of every completion's optimal cost
    (P_theta) and its suboptimality-test optimum t* an UPPER bound of every completion's
    (bar_E / bar_D) -- a prefix that is infeasible, or has t* < 0, kills all its completions.
"""

import heapq
import numpy as np
from scipy.optimize import linprog

from .lp_models import FixedCommutationModel
from .oracle_cpu import HIGHS_OPTIONS


class PrefixModel(FixedCommutationModel):
    """The relaxation of every mode sequence that starts with ``prefix`` (LP costs only)."""

    def __init__(self, mpc, prefix):
        prefix = tuple(prefix)
        k, N = len(prefix), mpc.N
        super().__init__(mpc, prefix + (0,) * (N - k))
        assert not self.quadratic
        self.prefix = prefix
        # dynamics of the steps >= k: gone (x_{k+1} .. x_N become free)
        self.A_dyn = self.A_dyn[:k * self.n_x]
        self.b_dyn = self.b_dyn[:k * self.n_x]
        # mode-region rows sit at the end of A_ub, in step order: drop those of the steps >= k
        drop = sum(mpc.regions[0][0].shape[0] if mpc.regions[0] is not None else 0
                   for _ in range(k, N))
        if drop:
            self.A_ub = self.A_ub[:-drop]
            self.b_ub = self.b_ub[:-drop]


def _solve(lp):
    return linprog(lp['c'], A_ub=lp['A_ub'], b_ub=lp['b_ub'], A_eq=lp['A_eq'], b_eq=lp['b_eq'],
                   bounds=lp.get('bounds', (None, None)), method='highs', options=HIGHS_OPTIONS)


def prefix_cost(mpc, prefix, theta):
    """Lower bound of min over completions of J*(theta, delta); +inf if the prefix is infeasible."""
    res = _solve(PrefixModel(mpc, prefix).lp_point(theta))
    return float(res.fun) if res.status == 0 else np.inf


def prefix_slack(mpc, prefix, R, V_bar, eps_a, eps_r):
    """Upper bound of max over completions of t*(delta); -inf if infeasible on the simplex."""
    res = _solve(PrefixModel(mpc, prefix).lp_bar_E(R, V_bar, eps_a, eps_r))
    return -float(res.fun) if res.status == 0 else -np.inf


def p_theta_bb(mpc, theta, tie_tol=1e-6):
    """
    P_theta by best-first search over prefixes.  Returns (J*, sequence, LPs solved).  The
    canonical tie rule of oracle_cpu (lowest enumeration index among optima within tie_tol)
    is kept by expanding ties in lexicographic order.
    """
    n_modes, N = mpc.delta_size, mpc.N
    heap = [(0., ())]
    n_lp = 0
    best = (np.inf, None)
    while heap:
        bound, prefix = heapq.heappop(heap)
        if bound > best[0] + tie_tol * (1. + abs(best[0])):
            break                                   # nothing left can beat (or tie) the incumbent
        if len(prefix) == N:
            if best[1] is None or bound < best[0] - tie_tol * (1. + abs(best[0])) or \
                    (abs(bound - best[0]) <= tie_tol * (1. + abs(best[0])) and prefix < best[1]):
                best = (bound, prefix)
            continue
        for i in range(n_modes):
            child = prefix + (i,)
            J = prefix_cost(mpc, child, theta)
            n_lp += 1
            if np.isfinite(J):
                heapq.heappush(heap, (J, child))
    return best[0], best[1], n_lp


def bar_e_bb(mpc, R, V_bar, eps_a, eps_r):
    """
    bar_E_delta_R by depth-first search: is there a sequence with t* >= 0?  Returns
    (closed, LPs solved).  A prefix whose relaxed t* is negative is never extended.
    """
    n_modes, N = mpc.delta_size, mpc.N
    stack = [()]
    n_lp = 0
    while stack:
        prefix = stack.pop()
        kids = []
        for i in range(n_modes):
            child = prefix + (i,)
            t = prefix_slack(mpc, child, R, V_bar, eps_a, eps_r)
            n_lp += 1
            if t >= 0.:
                if len(child) == N:
                    return False, n_lp               # a full sequence with t* >= 0: not closed
                kids.append((t, child))
        for t, child in sorted(kids):                # most promising child on top of the stack
            stack.append(child)
    return True, n_lp


def prefix_feasible_on(mpc, prefix, R):
    """Is the relaxation of ``prefix`` feasible somewhere on the simplex R?"""
    return _solve(PrefixModel(mpc, prefix).lp_min_over_simplex(R)).status == 0


def feasible_sequences(mpc, simplices):
    """
    The mode sequences that are feasible somewhere on the union of the simplices, by
    depth-first search over prefixes (the CPU statement of
    explicit_hybrid_mpc_amd/sequences.py).  Returns (sorted sequences, LPs solved).
    """
    n_modes, N = mpc.delta_size, mpc.N
    out, stack, n_lp = [], [()], 0
    while stack:
        prefix = stack.pop()
        for i in range(n_modes):
            child = prefix + (i,)
            ok = False
            for R in simplices:
                n_lp += 1
                if prefix_feasible_on(mpc, child, R):
                    ok = True
                    break
            if ok:
                (out if len(child) == N else stack).append(child)
    return sorted(out), n_lp


def prefix_min_on(mpc, prefix, R):
    """Minimum over the simplex of the prefix relaxation's optimal cost; +inf if infeasible."""
    res = _solve(PrefixModel(mpc, prefix).lp_min_over_simplex(R))
    return float(res.fun) if res.status == 0 else np.inf


def relevant_sequences(mpc, simplices, tie_tol=1e-6):
    """
    CPU statement of explicit_hybrid_mpc_amd/sequences.py ``relevant_sequences``: the sequences
    whose cost can be below U somewhere on the region, U = the largest vertex cost of the
    sequence a greedy dive finds, plus V_R's canonical answer on the region (the first sequence
    in enumeration order that is feasible at every vertex).
    Returns (sorted sequences, U, incumbent, LPs solved).
    """
    n_modes, N = mpc.delta_size, mpc.N
    n_lp = 0

    def cost(prefix):
        nonlocal n_lp
        n_lp += len(simplices)
        return min(prefix_min_on(mpc, prefix, R) for R in simplices)
    dive = ()
    for _ in range(N):
        kids = [dive + (i,) for i in range(n_modes)]
        dive = kids[int(np.argmin([cost(k) for k in kids]))]
    verts = np.unique(np.asarray(simplices).reshape(-1, np.asarray(simplices).shape[-1]), axis=0)
    U = max(prefix_cost(mpc, dive, v) for v in verts)
    n_lp += len(verts)
    if not np.isfinite(U):
        raise ValueError('incumbent infeasible at a vertex')
    bound = U + tie_tol * (1. + abs(U))
    alive = [()]
    for _ in range(N):
        alive = [pre + (i,) for pre in alive for i in range(n_modes)
                 if cost(pre + (i,)) <= bound]
    first = first_feasible_sequence(mpc, verts)
    return sorted(set(alive) | {first}), U, dive, n_lp


def first_feasible_sequence(mpc, points):
    """First sequence in enumeration order feasible at every point (depth-first, lexicographic)."""
    stack = [()]
    while stack:
        prefix = stack.pop()
        good = [prefix + (i,) for i in range(mpc.delta_size)
                if all(np.isfinite(prefix_cost(mpc, prefix + (i,), v)) for v in points)]
        if good and len(good[0]) == mpc.N:
            return good[0]
        stack.extend(reversed(good))
    return None


def _prefix_search_base():
    from explicit_hybrid_mpc_amd.sequences import PrefixSearch
    return PrefixSearch


class CpuPrefixTable(_prefix_search_base()):
    """
    The pair solvers of explicit_hybrid_mpc_amd/sequences.PrefixTable on the uncondensed
    relaxations with HiGHS: lets the CPU tests run the searches written on those solvers
    (sequences.PrefixSearch, bnb.PrefixOracle, bnb_frontier) without a device.
    """

    def __init__(self, mpc, eps_a=1., eps_r=1.):
        self.mpc = mpc
        self.eps_a, self.eps_r = eps_a, eps_r
        self.lp_solves = 0
        self._models = {}
        self.init_search()

    def set_eps(self, eps_a, eps_r):
        self.eps_a, self.eps_r = eps_a, eps_r

    def close(self):
        pass

    def _model(self, prefix):
        prefix = tuple(prefix)
        if prefix not in self._models:
            self._models[prefix] = PrefixModel(self.mpc, prefix)
        return self._models[prefix]

    def solve_points(self, prefixes, thetas, feasibility_only=False, known_feasible=False):
        # (known_feasible: the device table skips its phase-one launch; here one LP says both)
        thetas = np.asarray(thetas, dtype=np.float64).reshape(len(prefixes), -1)
        J = np.full(len(prefixes), np.inf)
        u0 = np.zeros((len(prefixes), self.mpc.n_u))
        for k, q in enumerate(prefixes):
            m = self._model(q)
            res = _solve(m.lp_point(thetas[k]))
            self.lp_solves += 1
            if res.status == 0:
                J[k] = 0. if feasibility_only else res.fun
                u0[k] = m.u0(res.x)
        return J, u0

    def solve_min(self, prefixes, simplices, known_feasible=None, exact=False):
        J = np.full(len(prefixes), np.inf)
        for k, q in enumerate(prefixes):
            res = _solve(self._model(q).lp_min_over_simplex(simplices[k]))
            self.lp_solves += 1
            if res.status == 0:
                J[k] = res.fun
        return J

    def solve_slack(self, prefixes, simplices, vbars, known_feasible=None):
        na = np.asarray(simplices).shape[1]
        t = np.full(len(prefixes), -np.inf)
        alpha = np.zeros((len(prefixes), na))
        for k, q in enumerate(prefixes):
            res = _solve(self._model(q).lp_bar_E(simplices[k], vbars[k], self.eps_a, self.eps_r))
            self.lp_solves += 1
            if res.status == 0:
                t[k] = -res.fun
                alpha[k] = res.x[-1 - na:-1]
        return t, alpha
