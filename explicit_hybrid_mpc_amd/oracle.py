"""
``Oracle`` with the reference's Python surface (lib/oracle.py:18-474), evaluated by the
HIP kernels of libehmpc.so.  Drop-in for the partitioning pipeline: same constructor,
same six methods, same return conventions (tuples / ``None`` / ``bool``), so
``Worker.ecc`` / ``Worker.lcss`` style callers, ``examples.create_oracle`` and
``ImplicitMPC`` keep working unchanged.

Differences a caller can observe:
* ``mpc`` is a ``mpc_library.PWAMPC`` (plain arrays) instead of a CVXPY model;
* the mixed-integer problems return the canonical commutation documented in DESIGN.md
  instead of "whatever the solver finds" (lib/oracle.py:201,347 are ``Minimize(0)``);
* every method also has a batched twin on ``Oracle.gpu`` (``engine.GpuProblem``) -- one
  kernel launch for thousands of parameters / simplices.
There is no CPU fallback: constructing an Oracle without a GPU raises.
"""

import time
import numpy as np

from .engine import GpuProblem


class SolverError(RuntimeError):
    """Raised where the reference raises cvx.SolverError (lib/oracle.py:440-442)."""


class Oracle:
    def __init__(self, mpc, eps_a, eps_r, device=0):
        """
        mpc : mpc_library.PWAMPC;  eps_a / eps_r : absolute / relative suboptimality
        tolerances (lib/oracle.py:23-39).
        """
        self.mpc = mpc
        self.eps_a = eps_a
        self.eps_r = eps_r
        self.canonical = mpc.compile()
        self.gpu = GpuProblem(self.canonical, eps_a, eps_r, device=device)

    def close(self):
        self.gpu.close()

    def _delta_of(self, idx):
        return self.canonical.deltas[int(idx)].copy()

    # -- lib/oracle.py:104-139 -----------------------------------------------------------
    def P_theta(self, theta, check_feasibility=False):
        t0 = time.time()
        J, u0, didx = self.gpu.solve_pt(np.asarray(theta, dtype=np.float64)[None])
        if check_feasibility:
            return bool(didx[0] >= 0)
        if didx[0] < 0:
            return None, None, None, time.time() - t0
        return u0[0], self._delta_of(didx[0]), float(J[0]), time.time() - t0

    # -- lib/oracle.py:141-173 -----------------------------------------------------------
    def P_theta_delta(self, theta, delta, check_feasibility=False):
        t0 = time.time()
        theta = np.asarray(theta, dtype=np.float64)[None]
        if check_feasibility:
            feas, _ = self.gpu.feasible_ptd(theta, delta)
            return bool(feas[0])
        feas, _ = self.gpu.feasible_ptd(theta, delta)
        if not feas[0]:
            return None, None, time.time() - t0
        J, u0, status, _ = self.gpu.solve_ptd(theta, delta)
        if status[0] != 0:
            return None, None, time.time() - t0
        return u0[0], float(J[0]), time.time() - t0

    @staticmethod
    def _vx_list(vJ, vu):
        return [(vu[i].copy(), float(vJ[i]), 0.) for i in range(vJ.shape[0])]

    # -- lib/oracle.py:175-218 -----------------------------------------------------------
    def V_R(self, R):
        didx, vJ, vu = self.gpu.v_r(np.asarray(R, dtype=np.float64)[None])
        if didx[0] < 0:
            return None, None
        return self._delta_of(didx[0]), self._vx_list(vJ[0], vu[0])

    # -- lib/oracle.py:285-309 -----------------------------------------------------------
    def bar_E_delta_R(self, R, V_delta_R):
        closed, _ = self.gpu.bar_e(np.asarray(R, dtype=np.float64)[None],
                                   np.asarray(V_delta_R, dtype=np.float64)[None])
        return bool(closed[0])

    # -- lib/oracle.py:220-283 -----------------------------------------------------------
    def in_variability_ball(self, R, V_delta_R, delta_ref, delta_star, theta_star):
        R = np.asarray(R, dtype=np.float64)[None]
        Jmin, st = self.gpu.min_simplex(R, delta_ref)
        J, _, st2, _ = self.gpu.solve_ptd(np.asarray(theta_star, dtype=np.float64)[None],
                                          delta_star)
        if st[0] != 0 or st2[0] != 0:
            raise SolverError('problem infeasible')
        rhs = max(self.eps_a, self.eps_r * float(J[0]))
        return bool(np.max(V_delta_R) - float(Jmin[0]) < rhs)

    # -- lib/oracle.py:311-414 -----------------------------------------------------------
    def bar_D_delta_R(self, R, V_delta_R, delta_ref):
        didx, ths, vJ, vu, vs = self.gpu.bar_d(np.asarray(R, dtype=np.float64)[None],
                                               np.asarray(V_delta_R, dtype=np.float64)[None],
                                               delta_ref)
        if didx[0] < 0:
            return None, None, None, None
        return (self._delta_of(didx[0]), ths[0].copy(), self._vx_list(vJ[0], vu[0]),
                bool(vs[0]))
