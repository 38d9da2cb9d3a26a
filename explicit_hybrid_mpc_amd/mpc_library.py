"""
MPC problem definitions for the partitioning hot path, and their compilation into
the dense canonical form the HIP kernels consume.

Role in the reference: ``lib/mpc_library.py`` builds CVXPY variables/constraints for
each MPC law and hands ``make_constraints``, ``cost``, ``V``, ``x0``, ``delta``,
``delta_size``, ``N``, ``n_x`` to the Oracle (contract at lib/mpc_library.py:26-39,
constraint pattern at :186-216 and, for big-M PWA dynamics, :521-552).  Here the same
information is carried as plain arrays (a piecewise-affine system, polyhedral state and
input sets, an infinity-norm stage cost), and ``PWAMPC.compile()`` condenses the states
out for every admissible mode sequence ("commutation") so that each oracle sub-problem
becomes

    J*(theta, d) = min_z  c^T z   s.t.   G_d z <= w_d + S_d theta ,      u0 = z[:n_u]

with z = [u_0 .. u_{N-1}, ex_1 .. ex_N, eu_0 .. eu_{N-1}] (inputs, state-cost epigraphs,
input-cost epigraphs).  A fixed commutation turns the reference's big-M rows
(lib/mpc_library.py:530-534) into exact mode dynamics, which is what the condensation
encodes.

Quadratic costs.  Every MPC law of the reference has a ``cvx.quad_form`` cost
(lib/mpc_library.py:180-183, :515-517).  With ``cost='quadratic'`` (and for ``SatelliteZ``)
the compiled form additionally carries, per commutation,

    V(z, theta) = c^T z + 1/2 z^T H_d z + (f0_d + F_d theta)^T z
                  + 1/2 theta^T C_d theta + c1_d^T theta + c0_d ,

so P_theta_delta is a convex QP and the suboptimality test (lib/oracle.py:89-97) a convex
QCQP with two quadratic rows; the infinity-norm epigraph variables are then absent.

The commutation vector layout follows the reference: ``delta[delta_size*k+i] == 1`` iff
mode ``i`` is active at step ``k`` (lib/mpc_library.py:160, lib/oracle.py:463-470); it is a
float64 0/1 vector of length ``delta_size*N``.
"""

import itertools
import numpy as np


class CanonicalLP:
    """
    Dense canonical data for all commutations of one MPC instance.

    Attributes
    ----------
    n, m, p, n_u, N, delta_size : int
        Decision dimension, inequality rows, parameter dimension, inputs, horizon,
        modes per step.
    n_delta : int
        Number of admissible commutations (mode sequences).
    G : (n_delta, m, n) float64     inequality matrix per commutation.
    w : (n_delta, m)    float64     right-hand-side offset.
    S : (n_delta, m, p) float64     right-hand-side parameter gain.
    c : (n,)            float64     linear cost (shared by all commutations).
    deltas : (n_delta, delta_size*N) float64   the 0/1 commutation vectors, in
        enumeration order (this order is the canonical tie-break order).
    """

    def __init__(self, G, w, S, c, deltas, n_u, N, delta_size, quad=None):
        self.G = np.ascontiguousarray(G, dtype=np.float64)
        self.w = np.ascontiguousarray(w, dtype=np.float64)
        self.S = np.ascontiguousarray(S, dtype=np.float64)
        self.c = np.ascontiguousarray(c, dtype=np.float64)
        self.deltas = np.ascontiguousarray(deltas, dtype=np.float64)
        self.n_delta, self.m, self.n = self.G.shape
        self.p = self.S.shape[2]
        self.n_u = int(n_u)
        self.N = int(N)
        self.delta_size = int(delta_size)
        # quadratic part of the cost (None for a pure LP cost): dict with
        #   H (n_delta,n,n), F (n_delta,n,p), f0 (n_delta,n), C (n_delta,p,p), c1 (n_delta,p),
        #   c0 (n_delta,)
        self.quadratic = quad is not None
        if quad is not None:
            for key in ('H', 'F', 'f0', 'C', 'c1', 'c0'):
                setattr(self, key, np.ascontiguousarray(quad[key], dtype=np.float64))
            assert self.H.shape == (self.n_delta, self.n, self.n)
            assert self.F.shape == (self.n_delta, self.n, self.p)

    def cost_value(self, d, z, theta):
        """V(z, theta) of commutation index d."""
        z = np.asarray(z, dtype=np.float64)
        theta = np.asarray(theta, dtype=np.float64)
        v = float(self.c @ z)
        if self.quadratic:
            v += 0.5 * z @ self.H[d] @ z + (self.f0[d] + self.F[d] @ theta) @ z
            v += 0.5 * theta @ self.C[d] @ theta + self.c1[d] @ theta + self.c0[d]
        return float(v)

    def delta_index(self, delta):
        """
        Map a commutation vector to its enumeration index.  The comparison is the
        reference's ``astype(int)`` one (lib/oracle.py:384-385).
        """
        key = np.asarray(delta).astype(int)
        hit = np.nonzero((self.deltas.astype(int) == key[None, :]).all(axis=1))[0]
        if hit.size == 0:
            raise ValueError('commutation %s is not an admissible mode sequence' %
                             (key.tolist(),))
        return int(hit[0])


class PWAMPC:
    """
    Piecewise-affine (hybrid) MPC with polyhedral constraints and an infinity-norm
    stage cost.  A single-mode instance is an ordinary linear MPC.

        x_{k+1} = A_i x_k + B_i u_k + w_i     if mode i is active at step k,
        Hx_i x_k <= hx_i                      (mode region, may be empty),
        Gx x_k <= gx , k = 1..N ;  Gu u_k <= gu , k = 0..N-1 ;  x_0 = theta ,
        V = sum_{k=1..N} ||Q x_k||_inf + sum_{k=0..N-1} ||R u_k||_inf          (cost='inf')
        V = sum_{k=0..N-1} u_k'R u_k + sum_{k=1..N-1} x_k'Q x_k + x_N'P x_N     (cost='quadratic',
            the form of lib/mpc_library.py:515-517; Q, R, P symmetric positive semidefinite,
            P defaults to Q).

    Exposes the attribute names of the reference's MPC contract
    (lib/mpc_library.py:26-39): ``N``, ``n_x``, ``n_u``, ``delta_size``.
    """

    def __init__(self, A, B, w, regions, Gx, gx, Gu, gu, Q, R, N, name='pwa_mpc',
                 cost='inf', P=None):
        self.A = [np.asarray(a, dtype=np.float64) for a in A]
        self.B = [np.asarray(b, dtype=np.float64) for b in B]
        self.w = [np.asarray(v, dtype=np.float64) for v in w]
        # regions[i] = (Hx_i, hx_i) or None
        self.regions = [None if r is None else
                        (np.asarray(r[0], dtype=np.float64),
                         np.asarray(r[1], dtype=np.float64)) for r in regions]
        self.Gx = np.asarray(Gx, dtype=np.float64)
        self.gx = np.asarray(gx, dtype=np.float64)
        self.Gu = np.asarray(Gu, dtype=np.float64)
        self.gu = np.asarray(gu, dtype=np.float64)
        self.Q = np.asarray(Q, dtype=np.float64)
        self.R = np.asarray(R, dtype=np.float64)
        self.N = int(N)
        self.n_x = self.A[0].shape[0]
        self.n_u = self.B[0].shape[1]
        self.delta_size = len(self.A)
        self.name = name
        if cost not in ('inf', 'quadratic'):
            raise ValueError("cost must be 'inf' or 'quadratic'")
        self.cost_type = cost
        self.P = self.Q if P is None else np.asarray(P, dtype=np.float64)
        if cost == 'quadratic':
            for M_, dim in ((self.Q, self.n_x), (self.P, self.n_x), (self.R, self.n_u)):
                if M_.shape != (dim, dim) or not np.allclose(M_, M_.T):
                    raise ValueError('quadratic cost weights must be symmetric and square')
        self._canonical = None

    # -- commutations ---------------------------------------------------------------
    def mode_sequences(self):
        """
        The admissible mode sequences, lexicographic with step 0 most significant: all of them,
        or the subset a caller has restricted the instance to (``restrict``).
        """
        if getattr(self, '_sequences', None) is not None:
            return list(self._sequences)
        return list(itertools.product(range(self.delta_size), repeat=self.N))

    def with_horizon(self, N):
        """The same law over the first ``N`` steps (infinity-norm cost: the stage costs are a plain
        sum, so the relaxation of a mode prefix of at most ``N`` steps -- the undecided steps cost
        nothing and constrain nothing PROVIDED u = 0 is admissible, G_u 0 <= g_u, which
        ``sequences.short_horizon`` checks before it offers the split -- is the same problem in
        either horizon)."""
        if self.cost_type != 'inf':
            raise ValueError('with_horizon: the quadratic cost has a terminal term')
        return PWAMPC(self.A, self.B, self.w, self.regions, self.Gx, self.gx, self.Gu, self.gu,
                      self.Q, self.R, N, name='%s_first%d' % (self.name, N), cost='inf')

    def restrict(self, sequences):
        """
        A copy of this instance whose commutation set is ``sequences`` (sorted into enumeration
        order): what a search over mode prefixes found feasible on a region
        (``sequences.feasible_sequences``) -- the sequences outside cannot be feasible there, so
        every oracle restricted to the region returns what it would with all of them.
        """
        import copy
        out = copy.copy(self)
        out._sequences = sorted(tuple(int(i) for i in s) for s in sequences)
        out._canonical = None
        out._prefix_template = None
        out._prefix_pred = {}
        out.name = '%s_restricted%d' % (self.name, len(out._sequences))
        return out

    def sequence_to_delta(self, seq):
        """Mode sequence -> reference-layout 0/1 vector (lib/mpc_library.py:160)."""
        d = np.zeros(self.delta_size * self.N)
        for k, i in enumerate(seq):
            d[self.delta_size * k + i] = 1.
        return d

    # -- condensation ---------------------------------------------------------------
    def _prediction(self, seq):
        """x_k = Phi[k] theta + Gam[k] U + om[k],  U = [u_0;..;u_{N-1}]."""
        n_x, n_u, N = self.n_x, self.n_u, self.N
        Phi = [np.eye(n_x)]
        Gam = [np.zeros((n_x, N * n_u))]
        om = [np.zeros(n_x)]
        for k in range(N):
            i = seq[k]
            A, B, w = self.A[i], self.B[i], self.w[i]
            Phi.append(A @ Phi[k])
            G_next = A @ Gam[k]
            G_next[:, k * n_u:(k + 1) * n_u] += B
            Gam.append(G_next)
            om.append(A @ om[k] + w)
        return Phi, Gam, om

    def n_rows_per_sequence(self, seq):
        m = self.N * (self.Gx.shape[0] + self.Gu.shape[0])
        if self.cost_type == 'inf':
            m += 2 * self.N * (self.Q.shape[0] + self.R.shape[0])
        for k in range(self.N):
            r = self.regions[seq[k]]
            if r is not None:
                m += r[0].shape[0]
        return m

    def _condense(self, seq, m_pad):
        n_x, n_u, N = self.n_x, self.n_u, self.N
        nU = N * n_u
        n = nU + (2 * N if self.cost_type == 'inf' else 0)
        Phi, Gam, om = self._prediction(seq)
        rows_G, rows_w, rows_S = [], [], []

        def add(Gz, wv, Sv):
            rows_G.append(Gz)
            rows_w.append(wv)
            rows_S.append(Sv)

        def pad(M_u):
            out = np.zeros((M_u.shape[0], n))
            out[:, :nU] = M_u
            return out
        # state constraints, k = 1..N :  Gx (Phi th + Gam U + om) <= gx
        for k in range(1, N + 1):
            add(pad(self.Gx @ Gam[k]), self.gx - self.Gx @ om[k], -self.Gx @ Phi[k])
        # input constraints
        for k in range(N):
            M_u = np.zeros((self.Gu.shape[0], nU))
            M_u[:, k * n_u:(k + 1) * n_u] = self.Gu
            add(pad(M_u), self.gu.copy(), np.zeros((self.Gu.shape[0], n_x)))
        # state-cost epigraphs:  +-Q x_k <= ex_k
        for k in (range(1, N + 1) if self.cost_type == 'inf' else ()):
            for sgn in (1., -1.):
                Gz = pad(sgn * self.Q @ Gam[k])
                Gz[:, nU + (k - 1)] = -1.
                add(Gz, -sgn * self.Q @ om[k], -sgn * self.Q @ Phi[k])
        # input-cost epigraphs:  +-R u_k <= eu_k
        for k in (range(N) if self.cost_type == 'inf' else ()):
            for sgn in (1., -1.):
                M_u = np.zeros((self.R.shape[0], nU))
                M_u[:, k * n_u:(k + 1) * n_u] = sgn * self.R
                Gz = pad(M_u)
                Gz[:, nU + N + k] = -1.
                add(Gz, np.zeros(self.R.shape[0]), np.zeros((self.R.shape[0], n_x)))
        # mode regions, k = 0..N-1 :  Hx_i x_k <= hx_i
        for k in range(N):
            r = self.regions[seq[k]]
            if r is not None:
                Hx, hx = r
                add(pad(Hx @ Gam[k]), hx - Hx @ om[k], -Hx @ Phi[k])
        G = np.vstack(rows_G)
        w = np.concatenate(rows_w)
        S = np.vstack(rows_S)
        m = G.shape[0]
        if m < m_pad:
            # pad with trivially satisfied rows  0 <= 1  so every commutation shares m
            G = np.vstack([G, np.zeros((m_pad - m, n))])
            w = np.concatenate([w, np.ones(m_pad - m)])
            S = np.vstack([S, np.zeros((m_pad - m, n_x))])
        return G, w, S

    def condense_prefix(self, prefix):
        """
        ``_condense_prefix_reference`` (see there for what the block is), built directly: the
        rows that do not depend on the modes come from a template, the prediction matrices of
        a prefix from its parent's (cached), and only the rows of the decided steps are
        computed.  Bit-identical to the reference form (tests/test_host_logic.py).
        """
        if self.cost_type != 'inf':
            raise ValueError('prefix relaxations are implemented for the infinity-norm cost')
        prefix = tuple(int(i) for i in prefix)
        k, N, n_u, n_x = len(prefix), self.N, self.n_u, self.n_x
        nU = N * n_u
        nGx, nGu, nQ, nR = self.Gx.shape[0], self.Gu.shape[0], self.Q.shape[0], self.R.shape[0]
        tpl = getattr(self, '_prefix_template', None)
        if tpl is None:
            G, w, S = self._condense_prefix_reference(())
            tpl = self._prefix_template = (G, w, S)
            self._prefix_pred = {(): (np.eye(n_x), np.zeros((n_x, nU)), np.zeros(n_x))}
        G, w, S = tpl[0].copy(), tpl[1].copy(), tpl[2].copy()
        pred = self._prefix_pred
        if len(pred) > 200000:
            pred.clear()
            pred[()] = (np.eye(n_x), np.zeros((n_x, nU)), np.zeros(n_x))
        states = [pred[()]]                              # (Phi_j, Gam_j, om_j), j = 0..k
        for j in range(k):
            q = prefix[:j + 1]
            st = pred.get(q)
            if st is None:
                Phi, Gam, om = states[j]
                A, B, wv = self.A[q[j]], self.B[q[j]], self.w[q[j]]
                G_next = A @ Gam
                G_next[:, j * n_u:(j + 1) * n_u] += B
                st = pred[q] = (A @ Phi, G_next, A @ om + wv)
            states.append(st)
        off_q = N * nGx + N * nGu
        off_r = off_q + 2 * N * nQ + 2 * N * nR
        row_r = off_r
        for kk in range(1, k + 1):
            Phi, Gam, om = states[kk]
            r0 = (kk - 1) * nGx                          # state constraints of x_kk
            G[r0:r0 + nGx, :nU] = self.Gx @ Gam
            w[r0:r0 + nGx] = self.gx - self.Gx @ om
            S[r0:r0 + nGx] = -self.Gx @ Phi
            for b, sgn in enumerate((1., -1.)):          # +-Q x_kk <= ex_kk
                r0 = off_q + (2 * (kk - 1) + b) * nQ
                G[r0:r0 + nQ, :nU] = sgn * self.Q @ Gam
                w[r0:r0 + nQ] = -sgn * self.Q @ om
                S[r0:r0 + nQ] = -sgn * self.Q @ Phi
        for kk in range(k):                              # mode regions of x_kk, kk < k
            reg = self.regions[prefix[kk]]
            if reg is not None:
                Hx, hx = reg
                Phi, Gam, om = states[kk]
                nr = Hx.shape[0]
                G[row_r:row_r + nr, :nU] = Hx @ Gam
                w[row_r:row_r + nr] = hx - Hx @ om
                S[row_r:row_r + nr] = -Hx @ Phi
                row_r += nr
        return G, w, S

    def _condense_prefix_reference(self, prefix):
        """
        (G, w, S) of the RELAXATION shared by every mode sequence that starts with ``prefix``
        (k = len(prefix) steps fixed): the rows that involve the states x_j, j > k -- state
        constraints, mode regions of the steps >= k -- become 0 <= 1, the epigraph rows of the
        stage costs of x_j, j > k, become e_j >= 0.  Same variables and the same m rows as a
        full sequence, so it is one more block of the commutation table.  Its optimal cost is a
        lower bound, its suboptimality-test optimum an upper bound, and its feasibility a
        necessary condition for every completion (the uncondensed
        statement is oracle/prefix_bb.py).  Infinity-norm costs.
        """
        if self.cost_type != 'inf':
            raise ValueError('prefix relaxations are implemented for the infinity-norm cost')
        prefix = tuple(int(i) for i in prefix)
        k, N, n_u = len(prefix), self.N, self.n_u
        seq = prefix + (0,) * (N - k)
        widest = max(range(self.delta_size),
                     key=lambda i: 0 if self.regions[i] is None else self.regions[i][0].shape[0])
        m_pad = self.n_rows_per_sequence((widest,) * N)      # the m of the compiled table
        G, w, S = self._condense(seq, m_pad)
        nU = N * n_u
        nGx, nGu, nQ, nR = self.Gx.shape[0], self.Gu.shape[0], self.Q.shape[0], self.R.shape[0]
        row = 0
        for kk in range(1, N + 1):                       # state constraints of x_kk
            if kk > k:
                G[row:row + nGx] = 0.
                w[row:row + nGx] = 1.
                S[row:row + nGx] = 0.
            row += nGx
        row += N * nGu                                   # input constraints: kept
        for kk in range(1, N + 1):                       # +-Q x_kk <= ex_kk
            for _ in range(2):
                if kk > k:
                    G[row:row + nQ] = 0.
                    G[row:row + nQ, nU + (kk - 1)] = -1.
                    w[row:row + nQ] = 0.
                    S[row:row + nQ] = 0.
                row += nQ
        row += 2 * N * nR                                # input-cost epigraphs: kept
        for kk in range(N):                              # mode regions of x_kk
            r = self.regions[seq[kk]]
            if r is not None:
                nr = r[0].shape[0]
                if kk >= k:
                    G[row:row + nr] = 0.
                    w[row:row + nr] = 1.
                    S[row:row + nr] = 0.
                row += nr
        return G, w, S

    def _quadratic_cost(self, seq):
        """
        V as a function of (U, theta) for one mode sequence (x_k = Phi_k theta + Gam_k U + om_k):
        1/2 U'H U + (f0 + F theta)'U + 1/2 theta'C theta + c1'theta + c0.
        """
        n_x, n_u, N = self.n_x, self.n_u, self.N
        nU = N * n_u
        Phi, Gam, om = self._prediction(seq)
        H = np.zeros((nU, nU))
        F = np.zeros((nU, n_x))
        f0 = np.zeros(nU)
        C = np.zeros((n_x, n_x))
        c1 = np.zeros(n_x)
        c0 = 0.
        for k in range(N):
            H[k * n_u:(k + 1) * n_u, k * n_u:(k + 1) * n_u] += 2. * self.R
        for k in range(1, N + 1):
            W = self.P if k == N else self.Q
            H += 2. * Gam[k].T @ W @ Gam[k]
            F += 2. * Gam[k].T @ W @ Phi[k]
            f0 += 2. * Gam[k].T @ W @ om[k]
            C += 2. * Phi[k].T @ W @ Phi[k]
            c1 += 2. * Phi[k].T @ W @ om[k]
            c0 += om[k] @ W @ om[k]
        return dict(H=0.5 * (H + H.T), F=F, f0=f0, C=0.5 * (C + C.T), c1=c1, c0=c0)

    def compile(self):
        """Condense every admissible commutation.  Returns a CanonicalLP."""
        if self._canonical is not None:
            return self._canonical
        seqs = self.mode_sequences()
        m_pad = max(self.n_rows_per_sequence(s) for s in seqs)
        Gs, ws, Ss, ds = [], [], [], []
        for s in seqs:
            G, w, S = self._condense(s, m_pad)
            Gs.append(G)
            ws.append(w)
            Ss.append(S)
            ds.append(self.sequence_to_delta(s))
        nU = self.N * self.n_u
        quad = None
        if self.cost_type == 'quadratic':
            c = np.zeros(nU)
            parts = [self._quadratic_cost(s) for s in seqs]
            quad = {key: np.stack([q[key] for q in parts]) for key in parts[0]}
        else:
            c = np.concatenate([np.zeros(nU), np.ones(2 * self.N)])
        self._canonical = CanonicalLP(np.stack(Gs), np.stack(ws), np.stack(Ss), c,
                                      np.stack(ds), self.n_u, self.N, self.delta_size,
                                      quad=quad)
        return self._canonical


def satellite_parameters():
    """Common CWH parameters, lib/mpc_library.py:794-823 (same names and values)."""
    pars = {'mu': 3.986004418e14,   # [m^3*s^-2] standard gravitational parameter
            'R_E': 6378137.,        # [m] Earth mean radius
            'h_E': 415e3,           # [m] orbit height
            'T_s': 100,             # [s] silent time
            'pos_err_max': 10e-2,   # [m]
            'vel_err_max': 1e-3,    # [m/s]
            'delta_v_max': 2e-3,    # [m/s]
            'w_max': 50e-9,         # [m/s^2] exogenous acceleration
            'sigma_fix': 1e-6,      # [m/s] fixed input error
            'sigma_pos': 2e-2,      # position estimate error growth slope
            'sigma_vel': 1e-3,      # velocity estimate error growth slope
            'input_ang_err': 2.,    # [deg] input error cone opening angle
            'p_max': 0.4e-2,        # [m] position estimate error
            'v_max': 4e-6}          # [m/s] velocity estimate error
    pars['a'] = pars['h_E'] + pars['R_E']
    pars['wo'] = np.sqrt(pars['mu'] / pars['a'] ** 3)
    pars['delta_v_min'] = pars['delta_v_max'] * 0.01
    pars['sigma_rcs'] = np.tan(np.deg2rad(pars['input_ang_err']) / 2.)
    return pars


class SatelliteZ:
    """
    The reference's ``SatelliteZ`` law ("cwh_z", lib/mpc_library.py:221-272 on top of
    ``MPC.setup_RMPC`` :61-219): CWH out-of-plane dynamics, input either off or on with a
    minimum impulse (U_ext minus U_int = two intervals, so ``delta_size`` = 2 and a step may
    also have no active piece), state constraints tightened against box-bounded process /
    estimation noise and against noise whose bound grows with |position|, |velocity| and
    |input|, quadratic cost  sum_k (u_k/dv_max)^2 + 1e-2 sum_{k=1..N} |D_x^-1 x_k|^2.

    With one position dimension every norm of the uncertainty model is an absolute value, so
    for a fixed commutation the law is a convex QP; ``compile()`` condenses it to

        z = [u_0..u_{N-1} | apos_0, avel_0, .., apos_{N-1}, avel_{N-1}]      (n = 3N)

    with ``a >= |x|`` epigraph rows (|u_k| needs none: its sign is fixed by the active
    input piece).  A step with no active piece has u_k = 0: its column is removed from the
    constraints and from the state part of the cost, and the input weight keeps the
    variable at zero.  Commutation sequences are per-step values 0 (off) or 1 + piece.

    The two- and three-axis laws (lib/mpc_library.py:274-405) carry genuine second-order
    cone rows (2-norms of 2-/3-vectors) and are outside this LP/QP path.
    """

    def __init__(self, N=4, name='cwh_z'):
        import scipy.linalg as sla
        pars = satellite_parameters()
        self.pars = pars
        self.N = int(N)
        self.n_x, self.n_u = 2, 1
        self.delta_size = 2
        self.T_s = pars['T_s']
        self.name = name
        A_c = np.array([[0., 1.], [-pars['wo'] ** 2, 0.]])
        B_c = np.array([[0.], [1.]])
        self.A = sla.expm(A_c * self.T_s)                         # :260-261
        self.B = self.A @ B_c
        Mx = np.block([[A_c, B_c], [np.zeros((1, 3))]])
        self.E = sla.expm(Mx * self.T_s)[:2, 2:]                  # :262-263
        self.Gx = np.array([[1., 0.], [-1., 0.], [0., 1.], [0., -1.]])
        self.gx = np.array([pars['pos_err_max']] * 2 + [pars['vel_err_max']] * 2)
        self.u_pieces = [(-pars['delta_v_max'], -pars['delta_v_min']),
                         (pars['delta_v_min'], pars['delta_v_max'])]
        self._canonical = None
        self._tightening()

    def _tightening(self):
        """sum_sigma and the dependent-noise coefficients, lib/mpc_library.py:107-138, 197-204."""
        pars, N, A = self.pars, self.N, self.A
        # disturbance gain, lib/plant.py:93-115: p = (process, state est., input, state, state, input)
        D = np.hstack([self.E, -A, self.B, -A, -A, self.B])       # 2 x 9
        ind = D[:, :3]                                             # independent, box-bounded
        ub = np.array([pars['w_max'], pars['p_max'], pars['v_max']])
        dep = [D[:, 3:4], D[:, 4:5], D[:, 7:8], D[:, 8:9]]         # D L_l, one column each
        n_g = self.gx.size
        self.sum_sigma = np.zeros((N + 1, n_g))
        self.coef = np.zeros((N + 1, N, 4, n_g))
        Apow = [np.linalg.matrix_power(A, k) for k in range(N + 1)]
        for k in range(1, N + 1):
            for i in range(k):
                GA = self.Gx @ Apow[k - 1 - i]
                self.sum_sigma[k] += np.abs(GA @ ind) @ ub
                for l in range(4):
                    self.coef[k, i, l] = np.abs(GA @ dep[l])[:, 0]
        self.sigma = np.array([pars['sigma_fix'], pars['sigma_pos'], pars['sigma_vel'],
                               pars['sigma_rcs']])

    # -- commutations ---------------------------------------------------------------
    def mode_sequences(self):
        """Per step 0 = no piece active, 1 + i = input piece i; step 0 most significant."""
        return list(itertools.product(range(self.delta_size + 1), repeat=self.N))

    def sequence_to_delta(self, seq):
        d = np.zeros(self.delta_size * self.N)
        for k, s in enumerate(seq):
            if s > 0:
                d[self.delta_size * k + (s - 1)] = 1.
        return d

    def box_vertices(self):
        """Vertices of the partitioned set (lib/examples.py:77-79)."""
        pe, ve = self.pars['pos_err_max'], self.pars['vel_err_max']
        return np.array(list(itertools.product([-pe, pe], [-ve, ve])))

    def _condense(self, seq):
        N = self.N
        n = 3 * N
        oa = N                                   # apos_i at oa + 2 i, avel_i at oa + 2 i + 1
        on = [s > 0 for s in seq]
        Phi = [np.eye(2)]
        Gam = [np.zeros((2, N))]
        for k in range(N):
            Phi.append(self.A @ Phi[k])
            Gn = self.A @ Gam[k]
            if on[k]:
                Gn[:, k] += self.B[:, 0]
            Gam.append(Gn)
        sign_u = [0. if s == 0 else (-1. if s == 1 else 1.) for s in seq]
        rows_G, rows_w, rows_S = [], [], []
        n_g = self.gx.size
        # tightened state rows, k = 1..N
        for k in range(1, N + 1):
            Gz = np.zeros((n_g, n))
            Gz[:, :N] = self.Gx @ Gam[k]
            const = self.sum_sigma[k].copy()
            for i in range(k):
                const += self.coef[k, i, 0] * self.sigma[0]
                Gz[:, oa + 2 * i] += self.coef[k, i, 1] * self.sigma[1]
                Gz[:, oa + 2 * i + 1] += self.coef[k, i, 2] * self.sigma[2]
                Gz[:, i] += self.coef[k, i, 3] * self.sigma[3] * sign_u[i]     # |u_i| = sign u_i
            rows_G.append(Gz)
            rows_w.append(self.gx - const)
            rows_S.append(-self.Gx @ Phi[k])
        # epigraphs  +-x_i[c] - a_{i,c} <= 0 ,  i = 0..N-1
        for i in range(N):
            for cidx in range(2):
                for sgn in (1., -1.):
                    Gz = np.zeros((1, n))
                    Gz[0, :N] = sgn * Gam[i][cidx]
                    Gz[0, oa + 2 * i + cidx] = -1.
                    rows_G.append(Gz)
                    rows_w.append(np.zeros(1))
                    rows_S.append(-sgn * Phi[i][cidx:cidx + 1])
        # input pieces (an inactive step keeps two trivially true rows so m is shared)
        for k in range(N):
            Gz = np.zeros((2, n))
            if on[k]:
                lo, hi = self.u_pieces[seq[k] - 1]
                Gz[0, k], Gz[1, k] = 1., -1.
                rows_w.append(np.array([hi, -lo]))
            else:
                rows_w.append(np.ones(2))
            rows_G.append(Gz)
            rows_S.append(np.zeros((2, 2)))
        G = np.vstack(rows_G)
        w = np.concatenate(rows_w)
        S = np.vstack(rows_S)
        # cost, lib/mpc_library.py:178-184
        H = np.zeros((n, n))
        F = np.zeros((n, 2))
        C = np.zeros((2, 2))
        du = self.pars['delta_v_max']
        H[:N, :N] += 2. * np.eye(N) / du ** 2
        Dxi = np.diag([1. / self.pars['pos_err_max'], 1. / self.pars['vel_err_max']])
        W = 1e-2 * Dxi @ Dxi
        for k in range(1, N + 1):
            H[:N, :N] += 2. * Gam[k].T @ W @ Gam[k]
            F[:N] += 2. * Gam[k].T @ W @ Phi[k]
            C += 2. * Phi[k].T @ W @ Phi[k]
        return G, w, S, dict(H=0.5 * (H + H.T), F=F, f0=np.zeros(n), C=0.5 * (C + C.T),
                             c1=np.zeros(2), c0=0.)

    def compile(self):
        if self._canonical is not None:
            return self._canonical
        seqs = self.mode_sequences()
        Gs, ws, Ss, qs, ds = [], [], [], [], []
        for s in seqs:
            G, w, S, q = self._condense(s)
            Gs.append(G)
            ws.append(w)
            Ss.append(S)
            qs.append(q)
            ds.append(self.sequence_to_delta(s))
        quad = {key: np.stack([q[key] for q in qs]) for key in qs[0]}
        self._canonical = CanonicalLP(np.stack(Gs), np.stack(ws), np.stack(Ss),
                                      np.zeros(3 * self.N), np.stack(ds), self.n_u, self.N,
                                      self.delta_size, quad=quad)
        return self._canonical
