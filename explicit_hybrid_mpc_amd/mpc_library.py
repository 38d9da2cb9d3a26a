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

    def __init__(self, G, w, S, c, deltas, n_u, N, delta_size):
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
        V = sum_{k=1..N} ||Q x_k||_inf + sum_{k=0..N-1} ||R u_k||_inf .

    Exposes the attribute names of the reference's MPC contract
    (lib/mpc_library.py:26-39): ``N``, ``n_x``, ``n_u``, ``delta_size``.
    """

    def __init__(self, A, B, w, regions, Gx, gx, Gu, gu, Q, R, N, name='pwa_mpc'):
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
        self._canonical = None

    # -- commutations ---------------------------------------------------------------
    def mode_sequences(self):
        """All mode sequences, lexicographic with step 0 most significant."""
        return list(itertools.product(range(self.delta_size), repeat=self.N))

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
        m += 2 * self.N * (self.Q.shape[0] + self.R.shape[0])
        for k in range(self.N):
            r = self.regions[seq[k]]
            if r is not None:
                m += r[0].shape[0]
        return m

    def _condense(self, seq, m_pad):
        n_x, n_u, N = self.n_x, self.n_u, self.N
        nU = N * n_u
        n = nU + 2 * N
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
        for k in range(1, N + 1):
            for sgn in (1., -1.):
                Gz = pad(sgn * self.Q @ Gam[k])
                Gz[:, nU + (k - 1)] = -1.
                add(Gz, -sgn * self.Q @ om[k], -sgn * self.Q @ Phi[k])
        # input-cost epigraphs:  +-R u_k <= eu_k
        for k in range(N):
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
        c = np.concatenate([np.zeros(nU), np.ones(2 * self.N)])
        self._canonical = CanonicalLP(np.stack(Gs), np.stack(ws), np.stack(Ss), c,
                                      np.stack(ds), self.n_u, self.N, self.delta_size)
        return self._canonical
