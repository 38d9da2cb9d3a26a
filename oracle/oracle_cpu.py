"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

CPU restatement of the reference's ``Oracle`` (lib/oracle.py:18-474): same six methods,
same argument meaning, same return conventions.  Every mixed-integer problem of the
reference is solved by enumerating the admissible commutations and solving one LP per
commutation with SciPy HiGHS -- one solver call per sub-problem, like the reference's
one ``Problem.solve`` per oracle.

Canonical commutation rule (the reference returns whatever feasible commutation MOSEK
happens to find, lib/oracle.py:201,347 -- ``Minimize(0)``):

* ``P_theta``      : commutation with the smallest optimal cost; costs within
  ``TIE_TOL*(1+|J|)`` of the minimum count as tied and the lowest index wins.
* ``V_R``          : first commutation (enumeration order) feasible at every vertex.
* ``bar_E_delta_R``: feasible  <=>  max over commutations of the slack t* is >= 0.
* ``bar_D_delta_R``: among commutations feasible at every vertex of R, the one with the
  largest slack t* (same tie rule), provided t* >= 0;  theta* is that LP's maximiser.
"""

import time
import numpy as np
from scipy.optimize import linprog

from .lp_models import FixedCommutationModel
from . import qp_numpy


# HiGHS defaults to 1e-7 feasibility tolerances; the parity bar on optimal costs is 1e-7
# relative, so the checker itself is run tighter.
HIGHS_OPTIONS = dict(primal_feasibility_tolerance=1e-10, dual_feasibility_tolerance=1e-10)


# Two commutations often give mathematically identical optima (e.g. both modes admissible in
# the guard band); which one a solver reports as "smaller" is then rounding noise.  Values
# closer than this (relative to 1+|value|) are ties, broken by enumeration order.
TIE_TOL = 1e-6



def hash_draw(seed, salt, code):
    """csrc/ehm_hybrid.h::hy_draw with draw = seed + 1: splitmix64 of (4 seed + salt, path code),
    upper 32 bits.  salt 1 = V_R, 2 = bar_D."""
    mask = (1 << 64) - 1
    z = ((((seed * 4 + salt) & 0xffffffff) << 32) | (code & 0xffffffff)) & mask
    z = (z + 0x9e3779b97f4a7c15) & mask
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & mask
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & mask
    z ^= z >> 31
    return z >> 32


class SolverError(RuntimeError):
    """Stands in for cvx.SolverError (lib/oracle.py:442)."""


class OracleCPU:
    def __init__(self, mpc, eps_a, eps_r):
        """Same signature and attributes as lib/oracle.py:23-39."""
        self.mpc = mpc
        self.eps_a = eps_a
        self.eps_r = eps_r
        self.sequences = mpc.mode_sequences()
        self.deltas = [mpc.sequence_to_delta(s) for s in self.sequences]
        # a model class may bring its own uncondensed restatement (oracle/satellite_cpu.py)
        make = getattr(mpc, 'fixed_commutation_model', None)
        self.models = [make(s) if make else FixedCommutationModel(mpc, s)
                       for s in self.sequences]
        self.n_solves = 0          # LP solver calls (one per commutation sub-problem)
        self.memoize = False       # cache point solves by (theta, commutation): children
        self._memo = {}            # share p of their p+1 vertices with the parent
        self.last_margin = np.inf  # |t*| of the last feasibility decision
        # test hook for the numerical-failure retry loops (lib/oracle.py:198-218, 406-414):
        # vertex solves of this commutation index raise like a failed MOSEK call would
        self.fail_vertex_solves_of = None
        self.n_blacklisted = 0
        # 'best' = the canonical rule above; 'first' = what a feasibility MICP solved by a
        # depth-first branch-and-bound tends to return: the FIRST commutation (enumeration
        # order) that satisfies the constraints (bar_D: feasible at every vertex and t* >= 0)
        self.bar_d_rule = 'best'
        # STUDY options for the reference's published leaf counts (lib/post_process.py:489,526;
        # tools/cwh_leafcount_study.py) -- never used by the parity tests:
        #   'random' rule: V_R and bar_D return ANY commutation that satisfies their constraints
        #   (the reference minimises 0, lib/oracle.py:201,347: its solver's choice), drawn with
        #   ``self.rng``;  ``verdict_tol`` tau: the feasibility problems count as feasible from
        #   t* >= -tau (1 + |V_0|) on, what a solver that accepts constraint violations of its
        #   feasibility tolerance does
        self.rng = None
        # rule 'hash': the draw of the device's option "any_admissible" (csrc/ehm_hybrid.h,
        # hy_draw): hash of (seed, oracle, path code of the node) -- PartitionCPU sets node_code
        self.hash_seed = 0
        self.node_code = 0
        self.verdict_tol = 0.
        # a QP solve that stalls is accepted below this residual / gap (the reference accepts
        # OPTIMAL_INACCURATE, lib/oracle.py:440-442); the studies loosen it
        self.qp_accept = 1e-9

    # -- helpers ---------------------------------------------------------------------
    def delta_index(self, delta):
        key = np.asarray(delta).astype(int)  # lib/oracle.py:384-385 comparison
        for d, ref in enumerate(self.deltas):
            if np.array_equal(ref.astype(int), key):
                return d
        raise ValueError('not an admissible commutation')

    def _solve(self, lp):
        self.n_solves += 1
        bounds = lp.get('bounds', (None, None))
        if lp.get('P') is not None or lp.get('quad'):
            return self._solve_quadratic(lp, bounds)
        res = linprog(lp['c'], A_ub=lp['A_ub'], b_ub=lp['b_ub'], A_eq=lp['A_eq'],
                      b_eq=lp['b_eq'], bounds=bounds, method='highs',
                      options=HIGHS_OPTIONS)
        return res

    def _solve_quadratic(self, lp, bounds):
        """
        Quadratic cost (every law of lib/mpc_library.py: ``cvx.quad_form``): feasibility of
        the linear rows is HiGHS' verdict (the quadratic rows of the suboptimality test can
        always be met by lowering t), the optimum comes from oracle/qp_numpy.py.
        """
        class _Res:
            pass
        res = _Res()
        n = lp['c'].size
        feas = linprog(np.zeros(n), A_ub=lp['A_ub'], b_ub=lp['b_ub'], A_eq=lp['A_eq'],
                       b_eq=lp['b_eq'], bounds=bounds, method='highs', options=HIGHS_OPTIONS)
        if feas.status != 0:
            res.status, res.x, res.fun = 2, None, None
            return res
        A_ub, b_ub = lp['A_ub'], lp['b_ub']
        if isinstance(bounds, list):         # lower bounds 0 of the simplex weights as rows
            extra = [j for j, (lo, hi) in enumerate(bounds) if lo is not None]
            rows = np.zeros((len(extra), n))
            for r, j in enumerate(extra):
                rows[r, j] = -1.
            A_ub = np.vstack([A_ub, rows])
            b_ub = np.concatenate([b_ub, np.zeros(len(extra))])
        out = qp_numpy.solve(lp['c'], A_ub, b_ub, lp['A_eq'], lp['b_eq'], P=lp.get('P'),
                             quad=lp.get('quad', ()))
        if out.status != 0 and max(out.res_p, out.res_d, out.gap) > self.qp_accept:
            raise SolverError('QP oracle did not converge (%g %g %g)' %
                              (out.res_p, out.res_d, out.gap))
        res.status, res.x, res.fun = 0, out.x, float(out.fun)
        return res

    def _point(self, theta, d):
        """(feasible, u0, J) of P_theta_delta for commutation index d."""
        key = None
        if self.memoize:
            key = (np.asarray(theta, dtype=np.float64).tobytes(), d)
            if key in self._memo:
                return self._memo[key]
        res = self._solve(self.models[d].lp_point(theta))
        if res.status != 0:
            out = (False, None, None)
        else:
            out = (True, self.models[d].u0(res.x), float(res.fun))
        if key is not None:
            self._memo[key] = out
        return out

    # -- lib/oracle.py:104-139 -----------------------------------------------------------
    def P_theta(self, theta, check_feasibility=False):
        t0 = time.time()
        cand = []
        for d in range(len(self.models)):
            ok, u, J = self._point(theta, d)
            if not ok:
                continue
            if check_feasibility:
                return True
            cand.append((d, u, J))
        if check_feasibility:
            return False
        if not cand:
            return None, None, None, time.time() - t0
        J_min = min(c[2] for c in cand)
        d, u, J = next(c for c in cand if c[2] <= J_min + TIE_TOL * (1. + abs(J_min)))
        return u, self.deltas[d].copy(), J, time.time() - t0

    # -- lib/oracle.py:141-173 -----------------------------------------------------------
    def P_theta_delta(self, theta, delta, check_feasibility=False):
        t0 = time.time()
        ok, u, J = self._point(theta, self.delta_index(delta))
        if check_feasibility:
            return ok
        if not ok:
            return None, None, time.time() - t0
        return u, J, time.time() - t0

    # -- lib/oracle.py:416-443 -----------------------------------------------------------
    def lexicographic_u0(self, theta, delta, tol=1e-6):
        """
        The lexicographically smallest first input over the tol-optimal face of P_theta_delta
        (oracle/lp_models.py: lp_point_lexicographic; the rule of
        explicit_hybrid_mpc_amd/lexicographic.py restated on the uncondensed model with HiGHS).
        """
        d = self.delta_index(delta)
        model = self.models[d]
        if model.quadratic:
            raise ValueError('quadratic cost: the optimum is a point')
        ok, _, J = self._point(theta, d)
        if not ok:
            raise SolverError('P_theta_delta infeasible')
        V_cap = J + tol * (1. + abs(J))
        caps, u = [], []
        for j in range(model.n_u):
            res = self._solve(model.lp_point_lexicographic(theta, j, V_cap, caps))
            if res.status != 0:
                raise SolverError('lexicographic stage %d failed' % (1 + j))
            u.append(float(res.fun))
            caps.append(u[-1] + tol * (1. + abs(u[-1])))
        return np.array(u)

    def _compute_vx_inputs_and_costs(self, R, delta):
        out = []
        if (self.fail_vertex_solves_of is not None and
                self.delta_index(delta) == self.fail_vertex_solves_of):
            raise SolverError('forced failure (test hook)')
        for vertex in R:
            u, J, t = self.P_theta_delta(theta=vertex, delta=delta)
            if u is None:
                raise SolverError('problem infeasible')
            out.append((u, J, t))
        return out

    def _feasible_on_vertices(self, R, d):
        return all(self._point(v, d)[0] for v in R)

    # -- lib/oracle.py:175-218 -----------------------------------------------------------
    def V_R(self, R):
        blacklist = set()                      # lib/oracle.py:198 delta_neq_other_deltas
        while True:
            found = None
            if self.bar_d_rule in ('random', 'hash'):
                ok = [d for d in range(len(self.models))
                      if d not in blacklist and self._feasible_on_vertices(R, d)]
                if not ok:
                    found = None
                elif self.bar_d_rule == 'hash':
                    found = ok[hash_draw(self.hash_seed, 1, self.node_code) % len(ok)]
                else:
                    found = ok[int(self.rng.integers(len(ok)))]
            else:
                for d in range(len(self.models)):
                    if d not in blacklist and self._feasible_on_vertices(R, d):
                        found = d
                        break
            if found is None:
                return None, None
            delta = self.deltas[found].copy()
            try:
                return delta, self._compute_vx_inputs_and_costs(R, delta)
            except SolverError:                # lib/oracle.py:214-218
                blacklist.add(found)
                self.n_blacklisted += 1

    # -- lib/oracle.py:285-309 -----------------------------------------------------------
    def slack(self, R, V_delta_R, d):
        """t*(d) and its maximiser (alpha) of the bar_E decision LP."""
        res = self._solve(self.models[d].lp_bar_E(R, V_delta_R, self.eps_a, self.eps_r))
        if res.status != 0:
            return -np.inf, None
        na = np.asarray(R).shape[0]
        return -float(res.fun), np.array(res.x[-1 - na:-1])

    def bar_E_delta_R(self, R, V_delta_R):
        t_best = max(self.slack(R, V_delta_R, d)[0] for d in range(len(self.models)))
        self.last_margin = abs(t_best)
        return not (t_best >= -self.verdict_tol * (1. + abs(float(np.asarray(V_delta_R)[0]))))

    # -- lib/oracle.py:220-283 -----------------------------------------------------------
    def in_variability_ball(self, R, V_delta_R, delta_ref, delta_star, theta_star):
        d_ref = self.delta_index(delta_ref)
        res = self._solve(self.models[d_ref].lp_min_over_simplex(R))
        if res.status != 0:
            raise SolverError('min over simplex failed')
        min_lhs = float(res.fun)
        max_lhs = np.max(V_delta_R)
        V_delta_theta = self.P_theta_delta(theta=theta_star, delta=delta_star)[1]
        rhs = max(self.eps_a, self.eps_r * V_delta_theta)
        return bool(max_lhs - min_lhs < rhs)

    # -- lib/oracle.py:311-414 -----------------------------------------------------------
    def bar_D_delta_R(self, R, V_delta_R, delta_ref):
        R = np.asarray(R, dtype=np.float64)
        cand = []
        for d in range(len(self.models)):
            if not self._feasible_on_vertices(R, d):
                continue
            t, alpha = self.slack(R, V_delta_R, d)
            if t >= -self.verdict_tol * (1. + abs(float(np.asarray(V_delta_R)[0]))):
                cand.append((t, d, alpha))
        blacklist = set()                      # lib/oracle.py:345 delta_blacklist
        while True:
            live = [c for c in cand if c[1] not in blacklist]
            if not live:
                return None, None, None, None
            t_max = max(c[0] for c in live)
            if self.bar_d_rule == 'first':
                best = live[0]
            elif self.bar_d_rule == 'random':
                best = live[int(self.rng.integers(len(live)))]
            elif self.bar_d_rule == 'hash':
                best = live[hash_draw(self.hash_seed, 2, self.node_code) % len(live)]
            else:
                best = next(c for c in live if c[0] >= t_max - TIE_TOL * (1. + abs(t_max)))
            delta_star = self.deltas[best[1]].copy()
            if np.array_equal(delta_star.astype(int), np.asarray(delta_ref).astype(int)):
                return None, None, None, None
            theta_star = best[2] @ R
            try:
                vx = self._compute_vx_inputs_and_costs(R, delta_star)
                var_small = self.in_variability_ball(R, V_delta_R, delta_ref, delta_star,
                                                     theta_star)
                return delta_star, theta_star, vx, var_small
            except SolverError:                # lib/oracle.py:406-414
                blacklist.add(best[1])
                self.n_blacklisted += 1
