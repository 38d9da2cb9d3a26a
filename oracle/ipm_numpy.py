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


# =========================================================================================
# Quadratic costs: the convex programs the QP-capable HIP kernels (ehm_ipm.h with a
# quadratic block) solve, in their condensed form, and the kernel's algorithm in numpy.
#
#   minimise    c_lin'y + kappa0 V(y)
#   subject to  A y <= b                                  (rows not in `iq`)
#               kappa_i V(y) + a_i'y <= b_i               (rows iq[i]; A[iq[i]] holds a_i)
#   V(y) = 1/2 y'Q y + q'y            (its constant v0 is folded into b_i and reported cost)
# =========================================================================================
class QuadProgram:
    __slots__ = ('c_lin', 'A', 'b', 'Q', 'q', 'v0', 'kappa0', 'iq', 'kap')


def _quad_blocks(can, d, v0, E):
    """Q, q, constant of V over (z, beta) with theta = v0 + E beta (E = None: theta = v0)."""
    n, p = can.n, can.p
    H, F, f0, C, c1, c0 = can.H[d], can.F[d], can.f0[d], can.C[d], can.c1[d], can.c0[d]
    qz = can.c + f0 + F @ v0
    const = 0.5 * v0 @ C @ v0 + c1 @ v0 + c0
    if E is None:
        return H.copy(), qz, const
    Q = np.zeros((n + p, n + p))
    Q[:n, :n] = H
    Q[:n, n:] = F @ E
    Q[n:, :n] = (F @ E).T
    Q[n:, n:] = E.T @ C @ E
    q = np.concatenate([qz, E.T @ (C @ v0 + c1)])
    return Q, q, const


def assemble_point_quad(can, d, theta):
    theta = np.asarray(theta, dtype=np.float64)
    pr = QuadProgram()
    pr.Q, pr.q, pr.v0 = _quad_blocks(can, d, theta, None)
    pr.c_lin = np.zeros(can.n)
    pr.A = can.G[d].copy()
    pr.b = can.w[d] + can.S[d] @ theta
    pr.kappa0, pr.iq, pr.kap = 1., [], []
    return pr


def assemble_min_simplex_quad(can, d, R):
    R = np.asarray(R, dtype=np.float64)
    A, b = _simplex_rows(can, d, R)
    pr = QuadProgram()
    pr.Q, pr.q, pr.v0 = _quad_blocks(can, d, R[0], (R[1:] - R[0]).T)
    pr.c_lin = np.zeros(can.n + can.p)
    pr.A, pr.b = A, b
    pr.kappa0, pr.iq, pr.kap = 1., [], []
    return pr


def assemble_bar_E_quad(can, d, R, V_bar, eps_a, eps_r):
    """Objective -t; rows  kappa_i V - dVbar'beta + t <= Vbar_0 - eps_i  (- kappa_i v0)."""
    R = np.asarray(R, dtype=np.float64)
    V_bar = np.asarray(V_bar, dtype=np.float64)
    A0, b0 = _simplex_rows(can, d, R)
    n, p = can.n, can.p
    Q, q, v0 = _quad_blocks(can, d, R[0], (R[1:] - R[0]).T)
    pr = QuadProgram()
    pr.Q = np.zeros((n + p + 1, n + p + 1))
    pr.Q[:n + p, :n + p] = Q
    pr.q = np.concatenate([q, [0.]])
    pr.v0 = v0
    m0 = A0.shape[0]
    A = np.zeros((m0 + 2, n + p + 1))
    A[:m0, :n + p] = A0
    dV = V_bar[1:] - V_bar[0]
    for r in (m0, m0 + 1):
        A[r, n:n + p] = -dV
        A[r, -1] = 1.
    pr.kap = [1., 1. + eps_r]
    pr.iq = [m0, m0 + 1]
    pr.A = A
    pr.b = np.concatenate([b0, [V_bar[0] - eps_a - pr.kap[0] * v0, V_bar[0] - pr.kap[1] * v0]])
    pr.c_lin = np.zeros(n + p + 1)
    pr.c_lin[-1] = -1.
    pr.kappa0 = 0.
    return pr


def solve_cp(pr, max_iter=60, tol_res=1e-10, tol_gap=1e-10, step_frac=STEP_FRAC_SAFE):
    """
    The kernel's algorithm (ehm_ipm.h, quadratic block on) for a QuadProgram: Mehrotra
    predictor-corrector on the normal equations  (w Q + J'DJ) dy = rhs,  J = the linear rows
    and the CURRENT gradients of the quadratic rows, w = kappa0 + sum_i kappa_i lambda_i,
    common primal/dual step, residuals of the quadratic rows re-evaluated every iterate,
    gap measured by s'lambda.  ``obj`` = c_lin'y + kappa0 (V(y) + v0).
    """
    A = pr.A.copy()
    b, Q, q = pr.b, pr.Q, pr.q
    m, n = A.shape
    a_lin = [pr.A[i].copy() for i in pr.iq]
    y = np.zeros(n)
    s = np.maximum(b, 1.)
    lam = np.ones(m)
    bnorm = 1. + np.max(np.abs(b))
    out = IPMResult()
    status = 1
    best = None
    stall = 0
    it = 0
    for it in range(max_iter + 1):
        gV = Q @ y + q
        yQy = y @ (gV - q)
        for i, r in enumerate(pr.iq):
            A[r] = pr.kap[i] * gV + a_lin[i]
        r_p = A @ y + s - b
        for i, r in enumerate(pr.iq):
            r_p[r] -= 0.5 * pr.kap[i] * yQy
        grad = pr.c_lin + pr.kappa0 * gV
        r_d = grad + A.T @ lam
        mu = s @ lam / m
        pobj = pr.c_lin @ y + pr.kappa0 * (0.5 * yQy + q @ y)
        cnorm = 1. + max(np.max(np.abs(pr.c_lin)), pr.kappa0 * np.max(np.abs(gV)))
        e_p = np.max(np.abs(r_p)) / bnorm
        # dual residual in the metric of the Hessian's diagonal, as in the kernels (ehm_ipm.h)
        e_d = np.max(np.abs(r_d) / np.sqrt(1. + np.diag(Q))) / cnorm
        e_g = (s @ lam) / (1. + abs(pobj + pr.kappa0 * pr.v0))
        merit = max(e_p / tol_res, e_d / tol_res, e_g / tol_gap)
        if best is None or merit < best[0]:
            best = (merit, y.copy(), lam.copy(), it, pobj)
            stall = 0
        elif best[0] < STALL_ZONE:
            stall += 1
        if merit <= 1.:
            status = 0
            break
        if stall >= 3 or it == max_iter:
            break
        dvec = lam / s
        wq = pr.kappa0 + sum(pr.kap[i] * lam[r] for i, r in enumerate(pr.iq))
        M = wq * Q + A.T @ (dvec[:, None] * A)
        L = guarded_cholesky(M)

        def newton(rc):
            rhs = -r_d + A.T @ ((rc - lam * r_p) / s)
            dy = chol_solve(L, rhs)
            ds = -r_p - A @ dy
            dl = -(rc + lam * ds) / s
            return dy, ds, dl

        def max_step(v, dv):
            neg = dv < 0
            return np.min(-v[neg] / dv[neg]) if neg.any() else 1e300
        dy_a, ds_a, dl_a = newton(s * lam)
        a = min(1., max_step(s, ds_a), max_step(lam, dl_a))
        mu_aff = (s + a * ds_a) @ (lam + a * dl_a) / m
        sigma = (mu_aff / mu) ** 3
        dy, ds, dl = newton(s * lam + ds_a * dl_a - sigma * mu)
        a = min(1., step_frac * min(max_step(s, ds), max_step(lam, dl)))
        y = y + a * dy
        s = s + a * ds
        lam = lam + a * dl
    merit, y, lam, _, pobj = best
    if status != 0 and merit <= ACCEPT_MERIT:
        status = 0
    out.x, out.obj, out.status, out.iters, out.lam = y, float(pobj + pr.kappa0 * pr.v0), status, it, lam
    out.merit = merit
    out.res = (0., 0.)
    return out
