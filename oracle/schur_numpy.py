"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

The block elimination the generation-4 solver (csrc/ehm_ipm2.h, "sparse columns") applies to the
normal equations of one interior-point iteration, written in numpy without any wave-level
tricks.  It changes how the Newton system is SOLVED, not the system: the iterates of the
interior-point method are the same (to rounding), which ``tests/test_oracle_schur.py`` checks
against the dense solve of ``oracle/ipm_numpy.py``.

Setting.  The LP columns of every oracle problem are  [z | beta | t]; the z-columns of an
infinity-norm MPC law end with its epigraph variables (lib/mpc_library.py:530-560 builds them
as ``ex_k``, ``eu_k``), and every MPC row holds AT MOST ONE of them.  Call that trailing range E
(nE columns), everything else D.  With the rows split into

    A0 = MPC rows + simplex rows   (singleton on E; the simplex rows touch D only)
    X  = the k <= 3 dense extra rows (suboptimality rows, phase-one bound)

the normal matrix is   M = A0' D0 A0 + X' Gam X   and its E x E block of the first term is
DIAGONAL:  Delta_e = sum_{i in rows(e)} d_i a_ie^2.  Eliminating E from the first term and
carrying the dense rows along as an augmented unknown  y = Gam X dx  gives (derivation in
DESIGN.md section 3.2b):

    G     = (A0' D0 A0)_DE                      nr x nE,  g_re = sum_{i in rows(e)} d_i a_ir a_ie
    S0    = (A0' D0 A0)_DD - G Delta^-1 G'      Schur complement, positive semidefinite
    Xh    = X_D - X_E Delta^-1 G'               the dense rows, reduced            (k x nr)
    Gh    = Gam^-1 + X_E Delta^-1 X_E'          k x k, positive definite, = L L'
    Xt    = L^-1 Xh
    S     = S0 + Xt' Xt                         what is factorised (nr x nr)

and for a right-hand side r = (r_D, r_E):

    rh    = r_D - G Delta^-1 r_E
    rho   = L^-1 (X_E Delta^-1 r_E)
    x_D   = S^-1 (rh - Xt' rho)
    y     = L^-T (Xt x_D + rho)
    x_E   = Delta^-1 (r_E - G' x_D - X_E' y)

Every term that is added is positive semidefinite (no cancellation beyond that of a Schur
complement of a positive definite matrix, which is benign).
"""

import numpy as np


def singleton_tail(G_all, max_rows=16):
    """
    Largest trailing range [nd0, n) of the z-columns such that, in EVERY commutation, no row
    holds more than one of them and no column more than ``max_rows`` entries.
    G_all: (n_delta, m, n).  Returns nd0 (= n when there is no such range).
    """
    nz = np.abs(np.asarray(G_all)) > 0.
    n = nz.shape[2]
    nd0 = n
    for cand in range(n - 1, -1, -1):
        tail = nz[:, :, cand:]
        if (tail.sum(axis=2) <= 1).all() and (tail.sum(axis=1) <= max_rows).all() and \
                (tail.sum(axis=1) >= 1).all():
            nd0 = cand
        else:
            break
    return nd0


def reduced_factor(A0, d0, X, gam, cols_D, cols_E):
    """Everything the solves need; see the module docstring."""
    AD, AE = A0[:, cols_D], A0[:, cols_E]
    assert ((np.abs(AE) > 0).sum(axis=1) <= 1).all(), 'E is not a singleton block'
    Delta = (d0[:, None] * AE * AE).sum(axis=0)
    G = AD.T @ (d0[:, None] * AE)
    S0 = AD.T @ (d0[:, None] * AD) - (G / Delta) @ G.T
    k = X.shape[0]
    XD, XE = X[:, cols_D], X[:, cols_E]
    Xh = XD - (XE / Delta) @ G.T
    Gh = np.diag(1. / gam) + (XE / Delta) @ XE.T if k else np.zeros((0, 0))
    L = np.linalg.cholesky(Gh) if k else np.zeros((0, 0))
    Xt = np.linalg.solve(L, Xh) if k else Xh
    S = S0 + Xt.T @ Xt
    return dict(Delta=Delta, G=G, S=S, Xt=Xt, L=L, XE=XE, cols_D=cols_D, cols_E=cols_E)


def reduced_solve(F, r, solve_S=np.linalg.solve):
    cols_D, cols_E = F['cols_D'], F['cols_E']
    rD, rE = r[cols_D], r[cols_E]
    Delta, G, Xt, L, XE = F['Delta'], F['G'], F['Xt'], F['L'], F['XE']
    k = Xt.shape[0]
    rh = rD - G @ (rE / Delta)
    rho = np.linalg.solve(L, XE @ (rE / Delta)) if k else np.zeros(0)
    xD = solve_S(F['S'], rh - Xt.T @ rho)
    y = np.linalg.solve(L.T, Xt @ xD + rho) if k else np.zeros(0)
    xE = (rE - G.T @ xD - XE.T @ y) / Delta
    x = np.zeros(len(r))
    x[cols_D] = xD
    x[cols_E] = xE
    return x


def solve_lp_reduced(c, A, b, n_dense_rows, cols_E, max_iter=40, tol_res=1e-10, tol_gap=1e-10,
                     step_frac=0.999):
    """
    oracle.ipm_numpy.solve_lp with the Newton systems solved through the reduction (single
    attempt at ``step_frac``, no stall bookkeeping: a check of the algebra on whole solves).
    The last ``n_dense_rows`` rows of A are the dense extra rows.
    Returns (x, obj, iterations, converged).
    """
    from .ipm_numpy import guarded_cholesky, chol_solve
    m, n = A.shape
    cols_E = np.asarray(cols_E, dtype=int)
    cols_D = np.array([j for j in range(n) if j not in set(cols_E.tolist())], dtype=int)
    m0 = m - n_dense_rows
    x = np.zeros(n)
    s = np.maximum(b - A @ x, 1.)
    lam = np.ones(m)
    bnorm = 1. + np.max(np.abs(b))
    cnorm = 1. + np.max(np.abs(c))

    def solve_S(S, rhs):
        return chol_solve(guarded_cholesky(S), rhs)

    for it in range(max_iter + 1):
        r_p = A @ x + s - b
        r_d = A.T @ lam + c
        mu = s @ lam / m
        pobj, dobj = c @ x, -b @ lam
        merit = max(np.max(np.abs(r_p)) / bnorm / tol_res, np.max(np.abs(r_d)) / cnorm / tol_res,
                    abs(pobj - dobj) / (1. + abs(pobj)) / tol_gap)
        if merit <= 1.:
            return x, float(pobj), it, True
        if it == max_iter:
            break
        dvec = lam / s
        F = reduced_factor(A[:m0], dvec[:m0], A[m0:], dvec[m0:], cols_D, cols_E)

        def solve(rc):
            rhs = -r_d + A.T @ ((rc - lam * r_p) / s)
            dx = reduced_solve(F, rhs, solve_S)
            ds = -r_p - A @ dx
            dl = -(rc + lam * ds) / s
            return dx, ds, dl

        def max_step(v, dv):
            neg = dv < 0
            return np.min(-v[neg] / dv[neg]) if neg.any() else 1e300
        dx_a, ds_a, dl_a = solve(s * lam)
        ap = min(1., max_step(s, ds_a))
        ad = min(1., max_step(lam, dl_a))
        sigma = ((s + ap * ds_a) @ (lam + ad * dl_a) / m / mu) ** 3
        dx, ds, dl = solve(s * lam + ds_a * dl_a - sigma * mu)
        ap = min(1., step_frac * max_step(s, ds))
        ad = min(1., step_frac * max_step(lam, dl))
        x = x + ap * dx
        s = s + ap * ds
        lam = lam + ad * dl
    return x, float(c @ x), max_iter, False
