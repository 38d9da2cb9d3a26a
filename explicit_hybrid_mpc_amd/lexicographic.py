"""
Lexicographic second stage for the vertex inputs of LP-cost laws.

What it is for.  The partition stores, at every vertex of every cell, the first input of the
fixed-commutation optimum there (lib/worker.py:356-365, 406-414) and the explicit law
interpolates those inputs (lib/mpc_library.py:786-789).  With an infinity-norm (LP) cost the
optimum of ``P_theta_delta`` need not be a point in u_0: the reference then stores whatever vertex
of the optimal face its solver stops at, and an interior-point method stops somewhere else on the
same face.  Both are optimal, both give an epsilon-suboptimal law, but the stored inputs are not a
function of the parameter any more.  This module makes them one, the same way on the device and in
the CPU oracle (``oracle/oracle_cpu.py``: ``OracleCPU.lexicographic_u0``):

    stage 0        V*    = min V(z)      s.t. the rows of P_theta_delta         (already solved)
    stage 1+j      u*_j  = min u_0[j]    s.t. the same rows,  V(z) <= V* + tol (1 + |V*|),
                                              u_0[i] <= u*_i + tol (1 + |u*_i|)   for i < j

so u* is the lexicographically smallest first input over the (tol-)optimal face -- unique by
construction.  Every stage is again a parametric LP in the canonical form the library takes
(``ehm_problem_create``): the caps are extra rows whose right-hand sides are extra PARAMETERS,

    G_j = [G ; c' ; e_0' .. e_{j-1}']    w_j = [w ; 0]    S_j = [S 0 ; 0 I]    c_j = e_j
    theta_j = (theta, V* + dV, u*_0 + du_0, ..)

and is solved in batches by ``ehm_solve_ptd_batch`` on its own problem handle: no kernel is
special to it and nothing runs on the CPU.  The parameter dimension grows by one per stage, so
the stage exists for  p + n_u <= EHM_MAX_P (= 8):  BASELINE.json's configs[0..2] (p = 4, n_u = 2;
the double integrator p = 2, n_u = 1).  Quadratic-cost laws are strictly convex in u: their
optimum is a point and they need no second stage.

    lex = LexicographicInputs(can)                 # can = mpc.compile()
    U = lex.solve(theta, delta_idx, J)             # (K, n_u)
    lex.refine(flat)                               # rewrites flat.vertex_inputs in place
"""

import numpy as np

from . import _capi
from .mpc_library import CanonicalLP

TOL = 1e-7          # relative width of the optimal face the later stages stay on: a hundred
#                     solver tolerances (1e-9).  Where the optimum IS a point the cap moves u_0 by
#                     (sensitivity of the LP) x TOL -- 9e-5 at most on the headline instance.
INACCURATE_MAX_DECADE = 6   # a stage solve that stalls on the thin face is taken up to merit 1e6
#                     (csrc/ehm_dev.h, ehm_status_word: its value is good to ~1e-4), as the
#                     reference takes OPTIMAL_INACCURATE (lib/oracle.py:440-442); n_inaccurate and
#                     worst_decade say how many there were and how far the worst one got
MAX_WIDENINGS = 3


def stage_problem(can, j):
    """Canonical data of stage 1+j (see the module docstring)."""
    nd, m, n, p = can.n_delta, can.m, can.n, can.p
    extra = 1 + j
    G = np.zeros((nd, m + extra, n))
    w = np.zeros((nd, m + extra))
    S = np.zeros((nd, m + extra, p + extra))
    G[:, :m] = can.G
    w[:, :m] = can.w
    S[:, :m, :p] = can.S
    G[:, m] = can.c[None, :]
    for i in range(j):
        G[:, m + 1 + i, i] = 1.
    for i in range(extra):
        S[:, m + i, p + i] = 1.
    c = np.zeros(n)
    c[j] = 1.
    return CanonicalLP(G, w, S, c, can.deltas, can.n_u, can.N, can.delta_size)


class LexicographicInputs:
    """The n_u stage problems of one canonical instance, resident on one GPU."""

    def __init__(self, can, device=0, tol=TOL):
        from .engine import GpuProblem
        if getattr(can, 'quadratic', False):
            raise ValueError('a quadratic cost is strictly convex in u: its optimum is a point '
                             'and needs no second stage')
        if can.p + can.n_u > _capi.EHM_MAX_P:
            raise ValueError('the lexicographic stage adds one parameter per input: p + n_u = '
                             '%d exceeds EHM_MAX_P = %d' % (can.p + can.n_u, _capi.EHM_MAX_P))
        self.can = can
        self.tol = float(tol)
        self.stages = [GpuProblem(stage_problem(can, j), 1., 1., device=device)
                       for j in range(can.n_u)]
        self.n_inaccurate = 0       # stage solves accepted below full accuracy so far
        self.worst_decade = 0
        self.n_widened = 0          # points solved again on a wider face

    def close(self):
        for gp in self.stages:
            gp.close()
        self.stages = []

    def _stages(self, theta, delta, J, tol):
        """All stages for one batch, face width ``tol`` per point.  Returns (U, failed)."""
        caps = [J + tol * (1. + np.abs(J))]
        U = np.empty((theta.shape[0], self.can.n_u))
        failed = np.zeros(theta.shape[0], dtype=bool)
        for j, gp in enumerate(self.stages):
            val, _, status, _ = gp.solve_ptd(np.column_stack([theta] + caps), delta)
            decade = (status >> 8) & 0xff
            stalled = status != 0
            bad = stalled & ((decade > INACCURATE_MAX_DECADE) | ~np.isfinite(val))
            failed |= bad
            good = stalled & ~bad
            self.n_inaccurate += int(good.sum())
            if good.any():
                self.worst_decade = max(self.worst_decade, int(decade[good].max()))
            val = np.where(np.isfinite(val), val, 0.)
            U[:, j] = val
            # (a stalled value may lie BELOW the true minimum by its error bar: the cap the next
            # stage stays under is widened by it, or that stage would be infeasible)
            width = np.where(stalled, np.maximum(tol, 10. ** (np.minimum(decade, 8) - 8.)), tol)
            caps.append(val + width * (1. + np.abs(val)))
        return U, failed

    def solve(self, theta, delta_idx, J):
        """
        theta (K, p), delta_idx (K,) commutation index of each point, J (K,) its optimal cost
        (stage 0).  Returns the lexicographically smallest optimal first inputs (K, n_u).

        A point at which a stage solve breaks down on the thin face (no usable iterate: a handful
        of the 189 k distinct vertices of the headline tree) is solved again from stage 1 with
        the face ten times wider, up to ``MAX_WIDENINGS`` times (``n_widened`` counts them).
        """
        theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
        J = np.asarray(J, dtype=np.float64).reshape(-1)
        delta = self.can.deltas[np.asarray(delta_idx, dtype=np.int64).reshape(-1)]
        K = theta.shape[0]
        U = np.empty((K, self.can.n_u))
        todo = np.arange(K)
        tol = np.full(K, self.tol)
        for attempt in range(1 + MAX_WIDENINGS):
            U[todo], failed = self._stages(theta[todo], delta[todo], J[todo], tol[todo])
            todo = todo[failed]
            if todo.size == 0:
                return U
            tol[todo] *= 10.
            self.n_widened += int(todo.size) if attempt < MAX_WIDENINGS else 0
        raise _capi.EhmError(_capi.EHM_E_NUMERIC,
                             'lexicographic stage failed at point %d (theta %s) on faces up to '
                             '%g wide' % (int(todo[0]), theta[todo[0]].tolist(),
                                          self.tol * 10. ** MAX_WIDENINGS))

    def refine(self, flat):
        """
        Rewrite ``flat.vertex_inputs`` (engine.FlatTree) with the lexicographic inputs of every
        node that carries data.  A vertex shared by many cells is solved once per commutation.
        Returns the number of distinct (vertex, commutation) pairs solved.
        """
        has = (flat.flags & 2) != 0 if len(self.can.deltas) > 1 else flat.delta_idx >= 0
        idx = np.nonzero(has)[0]
        if idx.size == 0:
            return 0
        p = self.can.p
        nv = p + 1
        V = flat.vertices[idx].reshape(-1, p)
        D = np.repeat(flat.delta_idx[idx].astype(np.int64), nv)
        C = flat.vertex_costs[idx].reshape(-1)
        key = np.empty((V.shape[0], p + 1))
        key[:, :p] = V
        key[:, p] = D
        # distinct rows by a 64-bit mix of their bit patterns (a sort of 8 M integers instead of
        # 8 M records); the grouping is verified and falls back to the record sort on a collision
        bits = key.view(np.uint64)
        mix = np.zeros(bits.shape[0], dtype=np.uint64)
        for q in range(p + 1):
            mix = (mix ^ bits[:, q]) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(q + 1)
            mix ^= mix >> np.uint64(29)
        _, first, inverse = np.unique(mix, return_index=True, return_inverse=True)
        if not np.array_equal(bits[first][inverse], bits):
            rows = key.view(np.dtype((np.void, key.dtype.itemsize * (p + 1)))).reshape(-1)
            _, first, inverse = np.unique(rows, return_index=True, return_inverse=True)
        U = self.solve(V[first], D[first], C[first])
        flat.vertex_inputs[idx] = U[inverse].reshape(idx.size, nv, self.can.n_u)
        return int(first.size)
