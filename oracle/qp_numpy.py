"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

Dense primal-dual interior-point solver for small convex quadratic programs with
(optionally) convex quadratic constraints -- the CPU stand-in for the reference's
``Problem.solve(solver='MOSEK')`` on quadratic-cost MPC laws (every model of
lib/mpc_library.py has a ``cvx.quad_form`` cost, :180-183, :515-517; with it the
suboptimality test lib/oracle.py:89-97 is a convex QCQP):

    minimise    1/2 x'P x + c'x
    subject to  A_ub x <= b_ub ,   A_eq x = b_eq ,
                1/2 x'P_i x + q_i'x + r_i <= 0        (i = 1..)

It works on the UNCONDENSED models (equality rows kept, Newton step from the full
symmetric quasi-definite augmented system, LU with partial pivoting), i.e. deliberately not the
normal-equation algorithm of the HIP kernels.  Published algorithm: Mehrotra
predictor-corrector for convex programs, Nocedal & Wright, Numerical Optimization 2nd ed.,
sections 16.6 and 19.3 (common primal/dual step length, residuals of the nonlinear rows re-evaluated
at every iterate).  The result carries the KKT residuals so tests can check optimality
independently of this code; SciPy's SLSQP is used as a second opinion in the tests.
"""

import numpy as np
import scipy.linalg as sla


class QPResult:
    __slots__ = ('x', 'fun', 'status', 'iters', 'lam', 'nu', 'res_p', 'res_d', 'gap')


def solve(c, A_ub, b_ub, A_eq=None, b_eq=None, P=None, quad=(), tol=1e-11, max_iter=80,
          equilibrate=True):
    """
    ``equilibrate``: solve in scaled variables x = D xs (D diagonal, every column of the
    stacked data brought to unit size) with every linear row scaled to unit infinity norm --
    the MPC models mix metres, millimetres per second and 1/dv_max^2 weights, and the
    indefinite KKT solve is not scale invariant.  The returned x / fun / lam are unscaled.
    """
    if equilibrate:
        return _solve_scaled(c, A_ub, b_ub, A_eq, b_eq, P, quad, tol, max_iter)
    c = np.asarray(c, dtype=np.float64)
    n = c.size
    A = np.asarray(A_ub, dtype=np.float64).reshape(-1, n)
    b = np.asarray(b_ub, dtype=np.float64)
    ml = A.shape[0]
    if A_eq is None:
        A_eq, b_eq = np.zeros((0, n)), np.zeros(0)
    A_eq = np.asarray(A_eq, dtype=np.float64).reshape(-1, n)
    b_eq = np.asarray(b_eq, dtype=np.float64)
    me = A_eq.shape[0]
    P = np.zeros((n, n)) if P is None else np.asarray(P, dtype=np.float64)
    quad = list(quad)
    mq = len(quad)
    m = ml + mq

    def g_and_J(x):
        g = np.empty(m)
        J = np.empty((m, n))
        g[:ml] = A.dot(x) - b
        J[:ml] = A
        for i, (Pi, qi, ri) in enumerate(quad):
            Px = Pi.dot(x)
            g[ml + i] = 0.5 * x.dot(Px) + qi.dot(x) + ri
            J[ml + i] = Px + qi
        return g, J

    x = np.zeros(n)
    if me:
        x = np.linalg.lstsq(A_eq, b_eq, rcond=None)[0]
    g, J = g_and_J(x)
    s = np.maximum(-g, 1.0)
    lam = np.ones(m)
    nu = np.zeros(me)
    scale_p = 1.0 + max(np.abs(b).max(initial=0.), np.abs(b_eq).max(initial=0.))
    res = QPResult()
    res.status = 1
    best_merit, stall = np.inf, 0
    for it in range(max_iter + 1):
        g, J = g_and_J(x)
        W = P.copy()
        for i, (Pi, _, _) in enumerate(quad):
            W += lam[ml + i] * Pi
        grad = P.dot(x) + c
        r_d = grad + J.T.dot(lam) + A_eq.T.dot(nu)
        r_p = g + s
        r_e = A_eq.dot(x) - b_eq
        mu = s.dot(lam) / m
        fun = 0.5 * x.dot(P.dot(x)) + c.dot(x)
        scale_d = 1.0 + max(np.abs(grad).max(initial=0.), np.abs(c).max(initial=0.))
        e_p = max(np.abs(r_p).max(initial=0.), np.abs(r_e).max(initial=0.)) / scale_p
        e_d = np.abs(r_d).max(initial=0.) / scale_d
        e_g = s.dot(lam) / (1.0 + abs(fun))
        merit = max(e_p, e_d, e_g)
        if merit < best_merit:
            best_merit, stall = merit, 0
            res.x, res.fun, res.iters = x.copy(), fun, it
            res.lam, res.nu = lam.copy(), nu.copy()
            res.res_p, res.res_d, res.gap = e_p, e_d, e_g
        elif best_merit < 1e-6:
            stall += 1          # (the merit of an infeasible-start method is not monotone early on)
        if merit <= tol:
            res.status = 0
            break
        if it == max_iter or stall >= 5:
            break
        # augmented (quasi-definite) system: unlike the reduced matrix W + J'(lam/s)J its
        # conditioning stays bounded as s/lam -> 0 on the active rows
        #   [ W    J'      Aeq' ] [dx ]   [ -r_d                 ]
        #   [ J   -S/Lam   0    ] [dl ] = [ -r_p + r_c / lam     ]
        #   [ Aeq  0       0    ] [dnu]   [ -r_e                 ]
        K = np.zeros((n + m + me, n + m + me))
        K[:n, :n] = W + 1e-13 * np.eye(n)
        K[:n, n:n + m] = J.T
        K[n:n + m, :n] = J
        K[n:n + m, n:n + m] = -np.diag(s / lam)
        K[:n, n + m:] = A_eq.T
        K[n + m:, :n] = A_eq
        K[n + m:, n + m:] = -1e-13 * np.eye(me)
        lu = sla.lu_factor(K)

        def newton(r_c):
            rhs = np.concatenate([-r_d, -r_p + r_c / lam, -r_e])
            sol = sla.lu_solve(lu, rhs)
            dx, dl, dnu = sol[:n], sol[n:n + m], sol[n + m:]
            ds = -r_p - J.dot(dx)
            return dx, dnu, ds, dl

        dx, dnu, ds, dl = newton(s * lam)
        a = min(1.0, _ratio(s, ds), _ratio(lam, dl))
        mu_aff = (s + a * ds).dot(lam + a * dl) / m
        sigma = (mu_aff / mu) ** 3
        dx, dnu, ds, dl = newton(s * lam + ds * dl - sigma * mu)
        a = min(1.0, 0.99 * min(_ratio(s, ds), _ratio(lam, dl)))
        x = x + a * dx
        nu = nu + a * dnu
        s = s + a * ds
        lam = lam + a * dl
    return res


def _ratio(v, dv):
    neg = dv < 0
    return float(np.min(-v[neg] / dv[neg])) if neg.any() else 1e300


def _solve_scaled(c, A_ub, b_ub, A_eq, b_eq, P, quad, tol, max_iter):
    c = np.asarray(c, dtype=np.float64)
    n = c.size
    A = np.asarray(A_ub, dtype=np.float64).reshape(-1, n)
    b = np.asarray(b_ub, dtype=np.float64)
    Ae = np.zeros((0, n)) if A_eq is None else np.asarray(A_eq, dtype=np.float64).reshape(-1, n)
    be = np.zeros(0) if b_eq is None else np.asarray(b_eq, dtype=np.float64)
    Pm = np.zeros((n, n)) if P is None else np.asarray(P, dtype=np.float64)
    size = np.maximum(np.abs(A).max(axis=0, initial=0.), np.abs(Ae).max(axis=0, initial=0.))
    size = np.maximum(size, np.sqrt(np.abs(np.diag(Pm))))
    for (Pi, qi, _) in quad:
        size = np.maximum(size, np.sqrt(np.abs(np.diag(Pi))))
    D = 1.0 / np.where(size > 0, size, 1.0)
    As, Aes = A * D, Ae * D
    ru = np.abs(As).max(axis=1, initial=0.)
    ru = np.where(ru > 0, ru, 1.0)
    re = np.abs(Aes).max(axis=1, initial=0.)
    re = np.where(re > 0, re, 1.0)
    quad_s = [((Pi * D).T * D, qi * D, ri) for (Pi, qi, ri) in quad]
    res = solve(c * D, As / ru[:, None], b / ru, Aes / re[:, None], be / re,
                P=(Pm * D).T * D, quad=quad_s, tol=tol, max_iter=max_iter, equilibrate=False)
    res.x = res.x * D
    ml = A.shape[0]
    res.lam = np.concatenate([res.lam[:ml] / ru, res.lam[ml:]])
    res.nu = res.nu / re
    return res
