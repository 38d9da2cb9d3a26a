"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

Second, independent CPU solver path used to cross-check the HiGHS oracle: the condensed
canonical LPs (``explicit_hybrid_mpc_amd.mpc_library.CanonicalLP``) assembled in numpy and
solved with a dense Mehrotra predictor-corrector interior-point method.  It is the
algorithm the HIP kernels implement (DESIGN.md "wave-local LP"), written without any
wave-level tricks, so a disagreement between kernel and oracle can be localised to
either the algorithm (this file disagrees with HiGHS) or the kernel (it does not).

Published algorithm: S. Mehrotra, "On the implementation of a primal-dual interior point
method", SIAM J. Optim. 2(4), 1992; form for  min c^T x  s.t.  A x + s = b, s >= 0  as in
Nocedal & Wright, Numerical Optimization, Alg. 14.3.
"""

import numpy as np


def assemble_point(can, d, theta):
    """P_theta_delta (lib/oracle.py:141-173) in condensed form."""
    return can.c.copy(), can.G[d].copy(), can.w[d] + can.S[d] @ np.asarray(theta)


def _simplex_rows(can, d, R):
    R = np.asarray(R, dtype=np.float64)
    p = can.p
    n, m = can.n, can.m
    Dv = (R[1:] - R[0]).T                      # (p, p), theta = R[0] + Dv beta
    A = np.zeros((m + p + 1, n + p))
    A[:m, :n] = can.G[d]
    A[:m, n:] = -can.S[d] @ Dv
    A[m:m + p, n:] = -np.eye(p)                # beta >= 0
    A[m + p, n:] = 1.                          # sum beta <= 1
    b = np.concatenate([can.w[d] + can.S[d] @ R[0], np.zeros(p), [1.]])
    return A, b


def assemble_min_simplex(can, d, R):
    """min over the simplex for a fixed commutation (lib/oracle.py:74-79)."""
    A, b = _simplex_rows(can, d, R)
    c = np.concatenate([can.c, np.zeros(can.p)])
    return c, A, b


def assemble_bar_E(can, d, R, V_bar, eps_a, eps_r):
    """Decision form of lib/oracle.py:89-97; objective is -t, so t* = -optimum."""
    A0, b0 = _simplex_rows(can, d, R)
    V_bar = np.asarray(V_bar, dtype=np.float64)
    n, p = can.n, can.p
    A = np.zeros((A0.shape[0] + 2, n + p + 1))
    A[:A0.shape[0], :n + p] = A0
    dV = V_bar[1:] - V_bar[0]
    A[-2, :n] = can.c
    A[-2, n:n + p] = -dV
    A[-2, -1] = 1.
    A[-1, :n] = (1. + eps_r) * can.c
    A[-1, n:n + p] = -dV
    A[-1, -1] = 1.
    b = np.concatenate([b0, [V_bar[0] - eps_a, V_bar[0]]])
    c = np.zeros(n + p + 1)
    c[-1] = -1.
    return c, A, b


def assemble_feasibility(can, d, theta):
    """Phase-one form:  min tau  s.t.  G z - tau <= h, tau >= -1;  feasible iff tau* <= 0."""
    n, m = can.n, can.m
    A = np.zeros((m + 1, n + 1))
    A[:m, :n] = can.G[d]
    A[:m, n] = -1.
    A[m, n] = -1.
    b = np.concatenate([can.w[d] + can.S[d] @ np.asarray(theta), [1.]])
    c = np.zeros(n + 1)
    c[n] = 1.
    return c, A, b


class IPMResult:
    __slots__ = ('x', 'obj', 'status', 'iters', 'lam', 'res', 'merit')


PROX_REL = 0.         # optional relative proximal term on the normal-matrix diagonal (off: it
                      # slows the tail convergence on these LPs; kept for experiments)
PIVOT_REL = 1e-13     # a pivot below PIVOT_REL * (its original diagonal) is a dependent column
PIVOT_BIG = 1e128     # replacing it by this zeroes the corresponding solution component
ACCEPT_MERIT = 1e3    # stalled but within 1000x of the tolerances: accepted (OPTIMAL_INACCURATE)
STALL_ZONE = 1e4      # non-improving iterations count as a stall only this close to the tolerances


def guarded_cholesky(M):
    """
    Lower Cholesky factor with the dependent-pivot guard of LIPSOL/PCx: when a pivot
    collapses relative to its original diagonal entry the column is frozen.
    """
    n = M.shape[0]
    L = np.tril(M).copy()
    d0 = np.diag(M).copy()
    for k in range(n):
        piv = L[k, k]
        if not (piv > PIVOT_REL * d0[k]) or not (piv > 0.):
            piv = PIVOT_BIG
        piv = np.sqrt(piv)
        L[k, k] = piv
        L[k + 1:, k] /= piv
        for j in range(k + 1, n):
            L[j:, j] -= L[j:, k] * L[j, k]
    return L


def chol_solve(L, rhs):
    n = L.shape[0]
    y = rhs.copy()
    for k in range(n):
        y[k] /= L[k, k]
        y[k + 1:] -= L[k + 1:, k] * y[k]
    for k in range(n - 1, -1, -1):
        y[k] /= L[k, k]
        y[:k] -= L[k, :k] * y[k]
    return y


STEP_FRAC = 0.999       # generation-2 kernels (ehm_ipm2.h); generation 1 uses STEP_FRAC_SAFE
STEP_FRAC_SAFE = 0.99   # a solve that stalls with STEP_FRAC is repeated with this one,
STEP_FRAC_LAST = 0.9    # and once more with this one (hybrid instances, a few per 1e7)


def solve_lp(c, A, b, max_iter=40, tol_res=1e-10, tol_gap=1e-10, step_frac=None,
             prox=PROX_REL):
    """
    min c^T x  s.t.  A x <= b.
    status: 0 optimal (all three relative criteria met), 1 stalled / iteration limit
    (best iterate returned, its merit in ``.merit``).
    step_frac=None: STEP_FRAC first, STEP_FRAC_SAFE if that stalls (ipm_solve_retry).
    """
    if step_frac is None:
        it = 0
        for sf in (STEP_FRAC, STEP_FRAC_SAFE, STEP_FRAC_LAST):
            out = solve_lp(c, A, b, max_iter, tol_res, tol_gap, sf, prox)
            it += out.iters
            if out.status == 0:
                break
        out.iters = it
        return out
    m, n = A.shape
    x = np.zeros(n)
    s = np.maximum(b - A @ x, 1.)
    lam = np.ones(m)
    bnorm = 1. + np.max(np.abs(b))
    cnorm = 1. + np.max(np.abs(c))
    out = IPMResult()
    status = 1
    best = None
    stall = 0
    it = 0
    for it in range(max_iter + 1):
        r_p = A @ x + s - b
        r_d = A.T @ lam + c
        mu = s @ lam / m
        pobj = c @ x
        dobj = -b @ lam
        e_p = np.max(np.abs(r_p)) / bnorm
        e_d = np.max(np.abs(r_d)) / cnorm
        e_g = abs(pobj - dobj) / (1. + abs(pobj))
        merit = max(e_p / tol_res, e_d / tol_res, e_g / tol_gap)
        if best is None or merit < best[0]:
            best = (merit, x.copy(), lam.copy(), it)
            stall = 0
        elif best[0] < STALL_ZONE:
            stall += 1
        if merit <= 1.:
            status = 0
            break
        if stall >= 3 or it == max_iter:
            break
        dvec = lam / s
        M = A.T @ (dvec[:, None] * A)
        # proximal (primal) regularisation relative to each column's own scale: bounds the
        # condition number of the scaled normal matrix by 1/prox on degenerate LPs
        M[np.diag_indices(n)] *= (1. + prox)
        L = guarded_cholesky(M)

        def solve(rc):
            rhs = -r_d + A.T @ ((rc - lam * r_p) / s)
            dx = chol_solve(L, rhs)
            ds = -r_p - A @ dx
            dl = -(rc + lam * ds) / s
            return dx, ds, dl

        def max_step(v, dv):
            neg = dv < 0
            if not neg.any():
                return 1e300
            return np.min(-v[neg] / dv[neg])
        dx_a, ds_a, dl_a = solve(s * lam)
        ap = min(1., max_step(s, ds_a))
        ad = min(1., max_step(lam, dl_a))
        mu_aff = (s + ap * ds_a) @ (lam + ad * dl_a) / m
        sigma = (mu_aff / mu) ** 3
        dx, ds, dl = solve(s * lam + ds_a * dl_a - sigma * mu)
        ap = min(1., step_frac * max_step(s, ds))
        ad = min(1., step_frac * max_step(lam, dl))
        x = x + ap * dx
        s = s + ap * ds
        lam = lam + ad * dl
    merit, x, lam, _ = best
    if status != 0 and merit <= ACCEPT_MERIT:
        status = 0
    out.x, out.obj, out.status, out.iters, out.lam = x, float(c @ x), status, it, lam
    out.merit = merit
    out.res = (float(np.max(np.abs(np.minimum(b - A @ x, 0.)))),
               float(np.max(np.abs(A.T @ lam + c))))
    return out
