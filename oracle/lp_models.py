"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

Uncondensed LP models of the oracle problems, straight from the reference's problem
statements in ``lib/oracle.py:23-102`` -- states stay decision variables and the
dynamics are equality constraints, like the CVXPY problems the reference builds from
``mpc.make_constraints(theta,x,u,delta)`` (contract lib/mpc_library.py:43-59; PWA pattern
:521-552 with the commutation fixed).  This is deliberately NOT the condensed form the
HIP kernels use, so that agreement between the two also checks the condensation.

Variable order:  v = [x_0 .. x_N | u_0 .. u_{N-1} | ex_1 .. ex_N | eu_0 .. eu_{N-1}]
followed, for the simplex problems, by [alpha_0 .. alpha_p] and, for bar_E, [t].
"""

import numpy as np


class FixedCommutationModel:
    """Constraint blocks of one MPC instance for one mode sequence."""

    def __init__(self, mpc, seq):
        self.mpc = mpc
        self.seq = tuple(seq)
        n_x, n_u, N = mpc.n_x, mpc.n_u, mpc.N
        self.n_x, self.n_u, self.N = n_x, n_u, N
        self.ox = 0
        self.ou = n_x * (N + 1)
        self.quadratic = getattr(mpc, 'cost_type', 'inf') == 'quadratic'
        self.oex = self.ou + n_u * N
        self.oeu = self.oex + (0 if self.quadratic else N)
        self.nv = self.oeu + (0 if self.quadratic else N)
        nv = self.nv
        # equalities: dynamics (x_0 = theta is added by the caller)
        Aeq, beq = [], []
        for k in range(N):
            i = self.seq[k]
            row = np.zeros((n_x, nv))
            row[:, self.ox + (k + 1) * n_x:self.ox + (k + 2) * n_x] = np.eye(n_x)
            row[:, self.ox + k * n_x:self.ox + (k + 1) * n_x] = -mpc.A[i]
            row[:, self.ou + k * n_u:self.ou + (k + 1) * n_u] = -mpc.B[i]
            Aeq.append(row)
            beq.append(mpc.w[i])
        self.A_dyn = np.vstack(Aeq)
        self.b_dyn = np.concatenate(beq)
        # x_0 selector
        self.A_x0 = np.zeros((n_x, nv))
        self.A_x0[:, :n_x] = np.eye(n_x)
        # inequalities
        Aub, bub = [], []
        for k in range(1, N + 1):
            row = np.zeros((mpc.Gx.shape[0], nv))
            row[:, self.ox + k * n_x:self.ox + (k + 1) * n_x] = mpc.Gx
            Aub.append(row)
            bub.append(mpc.gx)
        for k in range(N):
            row = np.zeros((mpc.Gu.shape[0], nv))
            row[:, self.ou + k * n_u:self.ou + (k + 1) * n_u] = mpc.Gu
            Aub.append(row)
            bub.append(mpc.gu)
        for k in (() if self.quadratic else range(1, N + 1)):
            for sgn in (1., -1.):
                row = np.zeros((mpc.Q.shape[0], nv))
                row[:, self.ox + k * n_x:self.ox + (k + 1) * n_x] = sgn * mpc.Q
                row[:, self.oex + k - 1] = -1.
                Aub.append(row)
                bub.append(np.zeros(mpc.Q.shape[0]))
        for k in (() if self.quadratic else range(N)):
            for sgn in (1., -1.):
                row = np.zeros((mpc.R.shape[0], nv))
                row[:, self.ou + k * n_u:self.ou + (k + 1) * n_u] = sgn * mpc.R
                row[:, self.oeu + k] = -1.
                Aub.append(row)
                bub.append(np.zeros(mpc.R.shape[0]))
        for k in range(N):
            r = mpc.regions[self.seq[k]]
            if r is not None:
                row = np.zeros((r[0].shape[0], nv))
                row[:, self.ox + k * n_x:self.ox + (k + 1) * n_x] = r[0]
                Aub.append(row)
                bub.append(r[1])
        self.A_ub = np.vstack(Aub)
        self.b_ub = np.concatenate(bub)
        # cost V = sum ex + sum eu,  or (cost_type 'quadratic', the form of
        # lib/mpc_library.py:515-517)  V = 1/2 v'P v = sum u'Ru + sum_{k<N} x'Qx + x_N'P x_N
        self.cost = np.zeros(nv)
        self.P = None
        if self.quadratic:
            Pm = np.zeros((nv, nv))
            for k in range(N):
                Pm[self.ou + k * n_u:self.ou + (k + 1) * n_u,
                   self.ou + k * n_u:self.ou + (k + 1) * n_u] = 2. * mpc.R
            for k in range(1, N + 1):
                Pm[self.ox + k * n_x:self.ox + (k + 1) * n_x,
                   self.ox + k * n_x:self.ox + (k + 1) * n_x] = 2. * (mpc.P if k == N else mpc.Q)
            self.P = Pm
        else:
            self.cost[self.oex:] = 1.

    def u0(self, v):
        return np.array(v[self.ou:self.ou + self.n_u])

    # -- P_theta_delta (lib/oracle.py:141-173) ---------------------------------------
    def lp_point(self, theta):
        A_eq = np.vstack([self.A_dyn, self.A_x0])
        b_eq = np.concatenate([self.b_dyn, np.asarray(theta, dtype=np.float64)])
        lp = dict(c=self.cost, A_ub=self.A_ub, b_ub=self.b_ub, A_eq=A_eq, b_eq=b_eq)
        if self.quadratic:
            lp['P'] = self.P
        return lp

    def lp_point_lexicographic(self, theta, j, V_cap, u_caps):
        """
        Stage 1+j of the lexicographic tie-break of the first input (no counterpart in the
        reference, which stores the vertex its solver stops at, lib/worker.py:356-365; the
        device form is explicit_hybrid_mpc_amd/lexicographic.py): min u_0[j] over the rows of
        P_theta_delta with V <= V_cap and u_0[i] <= u_caps[i] for i < j.
        """
        lp = self.lp_point(theta)
        rows = [self.cost[None, :]]
        rhs = [float(V_cap)]
        for i in range(j):
            r = np.zeros((1, self.nv))
            r[0, self.ou + i] = 1.
            rows.append(r)
            rhs.append(float(u_caps[i]))
        c = np.zeros(self.nv)
        c[self.ou + j] = 1.
        return dict(c=c, A_ub=np.vstack([lp['A_ub']] + rows),
                    b_ub=np.concatenate([lp['b_ub'], rhs]), A_eq=lp['A_eq'], b_eq=lp['b_eq'])

    # -- problems over a simplex (lib/oracle.py:70-79, 89-97) -------------------------
    def _simplex_blocks(self, R, extra_cols):
        """x_0 = sum_i alpha_i R[i], sum alpha = 1, alpha >= 0 (as bounds)."""
        R = np.asarray(R, dtype=np.float64)
        na = R.shape[0]
        nv, n_x = self.nv, self.n_x
        ntot = nv + na + extra_cols
        A_eq = np.zeros((self.A_dyn.shape[0] + n_x + 1, ntot))
        A_eq[:self.A_dyn.shape[0], :nv] = self.A_dyn
        r0 = self.A_dyn.shape[0]
        A_eq[r0:r0 + n_x, :nv] = self.A_x0
        A_eq[r0:r0 + n_x, nv:nv + na] = -R.T
        A_eq[r0 + n_x, nv:nv + na] = 1.
        b_eq = np.concatenate([self.b_dyn, np.zeros(n_x), [1.]])
        A_ub = np.zeros((self.A_ub.shape[0], ntot))
        A_ub[:, :nv] = self.A_ub
        bounds = [(None, None)] * nv + [(0., None)] * na + [(None, None)] * extra_cols
        return A_eq, b_eq, A_ub, self.b_ub.copy(), bounds, ntot, na

    def lp_min_over_simplex(self, R):
        A_eq, b_eq, A_ub, b_ub, bounds, ntot, na = self._simplex_blocks(R, 0)
        c = np.zeros(ntot)
        c[:self.nv] = self.cost
        lp = dict(c=c, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=b_eq, bounds=bounds)
        if self.quadratic:
            lp['P'] = np.zeros((ntot, ntot))
            lp['P'][:self.nv, :self.nv] = self.P
        return lp

    def lp_bar_E(self, R, V_bar, eps_a, eps_r):
        """
        Decision form of lib/oracle.py:89-97:  maximise t subject to the MPC constraints
        at theta = sum alpha_i v_i and
            sum alpha_i V_i - V - eps_a        >= t
            sum alpha_i V_i - (1+eps_r) V      >= t .
        The reference's feasibility problem is feasible iff t* >= 0.
        """
        A_eq, b_eq, A_ub, b_ub, bounds, ntot, na = self._simplex_blocks(R, 1)
        nv = self.nv
        V_bar = np.asarray(V_bar, dtype=np.float64)
        if self.quadratic:
            # two convex quadratic rows  kappa V(v) - sum alpha_i V_i + t + eps <= 0
            Pbig = np.zeros((ntot, ntot))
            Pbig[:nv, :nv] = self.P
            quad = []
            for kappa, r0 in ((1., eps_a), (1. + eps_r, 0.)):
                q = np.zeros(ntot)
                q[nv:nv + na] = -V_bar
                q[-1] = 1.
                quad.append((kappa * Pbig, q, r0))
            c = np.zeros(ntot)
            c[-1] = -1.
            return dict(c=c, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=b_eq, bounds=bounds,
                        quad=quad)
        rows = np.zeros((2, ntot))
        rows[0, :nv] = self.cost
        rows[0, nv:nv + na] = -V_bar
        rows[0, -1] = 1.
        rows[1, :nv] = (1. + eps_r) * self.cost
        rows[1, nv:nv + na] = -V_bar
        rows[1, -1] = 1.
        A_ub = np.vstack([A_ub, rows])
        b_ub = np.concatenate([b_ub, [-eps_a, 0.]])
        c = np.zeros(ntot)
        c[-1] = -1.
        return dict(c=c, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=b_eq, bounds=bounds)
