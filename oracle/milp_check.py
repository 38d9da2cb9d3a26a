"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

P_theta as ONE mixed-integer LP, solved by HiGHS' branch-and-bound (scipy.optimize.milp): an
independent statement of what the reference gives its MICP solver (lib/oracle.py:42-46 with the
mode logic of lib/mpc_library.py:521-552) -- binary mode indicators, big-M disaggregation of the
piecewise-affine dynamics -- against which the enumerating oracle and the prefix search are
pinned on instances small enough for it (tests/test_oracle_prefix_bounds.py).

    x_{k+1} = sum_i z_{k,i},   |z_{k,i}| <= M d_{k,i},
    |z_{k,i} - (A_i x_k + B_i u_k + w_i)| <= M (1 - d_{k,i}),
    Hx_i x_k <= hx_i + M (1 - d_{k,i}),   sum_i d_{k,i} = 1,   d binary,
    Gx x_k <= gx (k >= 1),  Gu u_k <= gu,  +-Q x_k <= ex_k 1,  +-R u_k <= eu_k 1,
    minimise sum ex_k + sum eu_k          (infinity-norm stage costs).
"""

import numpy as np
from scipy.optimize import milp, LinearConstraint, Bounds


def _model(mpc, big_m, extra_cols=0):
    """Rows of the mixed-integer model WITHOUT the x_0 condition; ``extra_cols`` more columns."""
    assert mpc.cost_type == 'inf'
    N, nx, nu, nm = mpc.N, mpc.n_x, mpc.n_u, mpc.delta_size
    ox = 0                              # x_0 .. x_N
    ou = ox + (N + 1) * nx              # u_0 .. u_{N-1}
    oz = ou + N * nu                    # z_{k,i}
    od = oz + N * nm * nx               # d_{k,i}
    oex = od + N * nm                   # ex_1 .. ex_N
    oeu = oex + N                       # eu_0 .. eu_{N-1}
    nv = oeu + N + extra_cols
    X = lambda k: slice(ox + k * nx, ox + (k + 1) * nx)
    U = lambda k: slice(ou + k * nu, ou + (k + 1) * nu)
    Z = lambda k, i: slice(oz + (k * nm + i) * nx, oz + (k * nm + i + 1) * nx)
    D = lambda k, i: od + k * nm + i
    rows, lo, hi = [], [], []

    def add(row, lb, ub):
        rows.append(row)
        lo.append(lb)
        hi.append(ub)

    def blank(n):
        return np.zeros((n, nv))
    for k in range(N):
        r = blank(1)
        for i in range(nm):
            r[0, D(k, i)] = 1.
        add(r, [1.], [1.])
        r = blank(nx)                                   # x_{k+1} = sum_i z_{k,i}
        r[:, X(k + 1)] = np.eye(nx)
        for i in range(nm):
            r[:, Z(k, i)] -= np.eye(nx)
        add(r, np.zeros(nx), np.zeros(nx))
        for i in range(nm):
            for sgn in (1., -1.):
                r = blank(nx)                           # +-z <= M d
                r[:, Z(k, i)] = sgn * np.eye(nx)
                r[:, D(k, i)] = -big_m
                add(r, np.full(nx, -np.inf), np.zeros(nx))
                r = blank(nx)                           # +-(z - A x - B u - w) <= M (1 - d)
                r[:, Z(k, i)] = sgn * np.eye(nx)
                r[:, X(k)] = -sgn * mpc.A[i]
                r[:, U(k)] = -sgn * mpc.B[i]
                r[:, D(k, i)] = big_m
                add(r, np.full(nx, -np.inf), big_m + sgn * mpc.w[i])
            if mpc.regions[i] is not None:
                Hx, hx = mpc.regions[i]
                r = blank(Hx.shape[0])
                r[:, X(k)] = Hx
                r[:, D(k, i)] = big_m
                add(r, np.full(Hx.shape[0], -np.inf), hx + big_m)
        r = blank(mpc.Gx.shape[0])
        r[:, X(k + 1)] = mpc.Gx
        add(r, np.full(mpc.Gx.shape[0], -np.inf), mpc.gx)
        r = blank(mpc.Gu.shape[0])
        r[:, U(k)] = mpc.Gu
        add(r, np.full(mpc.Gu.shape[0], -np.inf), mpc.gu)
        for sgn in (1., -1.):
            r = blank(mpc.Q.shape[0])
            r[:, X(k + 1)] = sgn * mpc.Q
            r[:, oex + k] = -1.
            add(r, np.full(mpc.Q.shape[0], -np.inf), np.zeros(mpc.Q.shape[0]))
            r = blank(mpc.R.shape[0])
            r[:, U(k)] = sgn * mpc.R
            r[:, oeu + k] = -1.
            add(r, np.full(mpc.R.shape[0], -np.inf), np.zeros(mpc.R.shape[0]))
    cost = np.zeros(nv)
    cost[oex:oeu + N] = 1.
    integrality = np.zeros(nv)
    integrality[od:oex] = 1
    lb = np.full(nv, -np.inf)
    ub = np.full(nv, np.inf)
    lb[od:oex], ub[od:oex] = 0., 1.
    return dict(rows=rows, lo=lo, hi=hi, cost=cost, integrality=integrality, lb=lb, ub=ub,
                nv=nv, x0=X(0), d=slice(od, oex), tail=oeu + N, add=add, blank=blank)


def _solve(M, c):
    res = milp(c, constraints=LinearConstraint(np.vstack(M['rows']), np.concatenate(M['lo']),
                                               np.concatenate(M['hi'])),
               integrality=M['integrality'], bounds=Bounds(M['lb'], M['ub']),
               options=dict(mip_rel_gap=1e-10))
    return res


def p_theta_milp(mpc, theta, big_m=50.):
    """(optimal cost, mode sequence) of the mixed-integer LP; (inf, None) if infeasible."""
    M = _model(mpc, big_m)
    r = M['blank'](mpc.n_x)
    r[:, M['x0']] = np.eye(mpc.n_x)
    M['add'](r, np.asarray(theta, dtype=float), np.asarray(theta, dtype=float))
    res = _solve(M, M['cost'])
    if res.status != 0:
        return np.inf, None
    d = np.rint(res.x[M['d']]).reshape(mpc.N, mpc.delta_size)
    return float(res.fun), tuple(int(i) for i in d.argmax(axis=1))


def bar_e_milp(mpc, R, V_bar, eps_a, eps_r, big_m=50.):
    """
    The reference's bar_E problem (lib/oracle.py:89-97) as ONE mixed-integer LP in decision form:
    maximise t over the modes, the parameter theta = sum_i alpha_i v_i in the simplex and the
    trajectory, subject to  sum alpha_i V_i - V - eps_a >= t,  sum alpha_i V_i - (1+eps_r) V >= t.
    Returns (t_max, mode sequence); (-inf, None) if no mode sequence is feasible on the simplex.
    The reference's feasibility problem is feasible iff t_max >= 0.
    """
    R = np.asarray(R, dtype=float)
    na = R.shape[0]
    M = _model(mpc, big_m, extra_cols=na + 1)
    oa, ot = M['tail'], M['tail'] + na
    r = M['blank'](mpc.n_x)                       # x_0 = sum alpha_i v_i
    r[:, M['x0']] = np.eye(mpc.n_x)
    r[:, oa:oa + na] = -R.T
    M['add'](r, np.zeros(mpc.n_x), np.zeros(mpc.n_x))
    r = M['blank'](1)
    r[0, oa:oa + na] = 1.
    M['add'](r, [1.], [1.])
    M['lb'][oa:oa + na] = 0.
    for scale, shift in ((1., eps_a), (1. + eps_r, 0.)):
        r = M['blank'](1)                         # sum alpha V_i - scale * V - t >= shift
        r[0, oa:oa + na] = np.asarray(V_bar, dtype=float)
        r[0, :] -= scale * M['cost']
        r[0, ot] = -1.
        M['add'](r, [shift], [np.inf])
    c = np.zeros(M['nv'])
    c[ot] = -1.
    res = _solve(M, c)
    if res.status != 0:
        return -np.inf, None
    d = np.rint(res.x[M['d']]).reshape(mpc.N, mpc.delta_size)
    return float(-res.fun), tuple(int(i) for i in d.argmax(axis=1))


def _model_copies(mpc, big_m, copies, extra_cols=0):
    """
    ``copies`` trajectories (x, u, z, ex, eu) that SHARE the binary mode indicators d -- the shape
    of the reference's V_R and bar_D problems (lib/oracle.py:57-66, 101-102: per-vertex copies of
    x and u, one delta).  Column layout: copy c at [c * blk, (c + 1) * blk), then d, then
    ``extra_cols``.  Rows without any x_0 condition.
    """
    assert mpc.cost_type == 'inf'
    N, nx, nu, nm = mpc.N, mpc.n_x, mpc.n_u, mpc.delta_size
    blk = (N + 1) * nx + N * nu + N * nm * nx + 2 * N
    od = copies * blk
    nv = od + N * nm + extra_cols

    def X(c, k):
        return slice(c * blk + k * nx, c * blk + (k + 1) * nx)

    def U(c, k):
        o = c * blk + (N + 1) * nx
        return slice(o + k * nu, o + (k + 1) * nu)

    def Z(c, k, i):
        o = c * blk + (N + 1) * nx + N * nu
        return slice(o + (k * nm + i) * nx, o + (k * nm + i + 1) * nx)

    def EX(c, k):
        return c * blk + (N + 1) * nx + N * nu + N * nm * nx + k

    def EU(c, k):
        return EX(c, 0) + N + k

    def D(k, i):
        return od + k * nm + i
    rows, lo, hi = [], [], []

    def add(row, lb, ub):
        rows.append(row)
        lo.append(np.atleast_1d(np.asarray(lb, dtype=float)))
        hi.append(np.atleast_1d(np.asarray(ub, dtype=float)))

    def blank(n):
        return np.zeros((n, nv))
    for k in range(N):
        r = blank(1)
        for i in range(nm):
            r[0, D(k, i)] = 1.
        add(r, [1.], [1.])
    for c in range(copies):
        for k in range(N):
            r = blank(nx)
            r[:, X(c, k + 1)] = np.eye(nx)
            for i in range(nm):
                r[:, Z(c, k, i)] -= np.eye(nx)
            add(r, np.zeros(nx), np.zeros(nx))
            for i in range(nm):
                for sgn in (1., -1.):
                    r = blank(nx)
                    r[:, Z(c, k, i)] = sgn * np.eye(nx)
                    r[:, D(k, i)] = -big_m
                    add(r, np.full(nx, -np.inf), np.zeros(nx))
                    r = blank(nx)
                    r[:, Z(c, k, i)] = sgn * np.eye(nx)
                    r[:, X(c, k)] = -sgn * mpc.A[i]
                    r[:, U(c, k)] = -sgn * mpc.B[i]
                    r[:, D(k, i)] = big_m
                    add(r, np.full(nx, -np.inf), big_m + sgn * mpc.w[i])
                if mpc.regions[i] is not None:
                    Hx, hx = mpc.regions[i]
                    r = blank(Hx.shape[0])
                    r[:, X(c, k)] = Hx
                    r[:, D(k, i)] = big_m
                    add(r, np.full(Hx.shape[0], -np.inf), hx + big_m)
            r = blank(mpc.Gx.shape[0])
            r[:, X(c, k + 1)] = mpc.Gx
            add(r, np.full(mpc.Gx.shape[0], -np.inf), mpc.gx)
            r = blank(mpc.Gu.shape[0])
            r[:, U(c, k)] = mpc.Gu
            add(r, np.full(mpc.Gu.shape[0], -np.inf), mpc.gu)
            for sgn in (1., -1.):
                r = blank(mpc.Q.shape[0])
                r[:, X(c, k + 1)] = sgn * mpc.Q
                r[:, EX(c, k)] = -1.
                add(r, np.full(mpc.Q.shape[0], -np.inf), np.zeros(mpc.Q.shape[0]))
                r = blank(mpc.R.shape[0])
                r[:, U(c, k)] = sgn * mpc.R
                r[:, EU(c, k)] = -1.
                add(r, np.full(mpc.R.shape[0], -np.inf), np.zeros(mpc.R.shape[0]))
    integrality = np.zeros(nv)
    integrality[od:od + N * nm] = 1
    lb = np.full(nv, -np.inf)
    ub = np.full(nv, np.inf)
    lb[od:od + N * nm], ub[od:od + N * nm] = 0., 1.

    def cost_of(c):
        v = np.zeros(nv)
        v[EX(c, 0):EX(c, 0) + 2 * N] = 1.
        return v
    return dict(rows=rows, lo=lo, hi=hi, integrality=integrality, lb=lb, ub=ub, nv=nv,
                x0=lambda c: X(c, 0), d=slice(od, od + N * nm), tail=od + N * nm, add=add,
                blank=blank, cost_of=cost_of)


def _sequence(mpc, res, M):
    d = np.rint(res.x[M['d']]).reshape(mpc.N, mpc.delta_size)
    return tuple(int(i) for i in d.argmax(axis=1))


def v_r_milp(mpc, R, big_m=50.):
    """
    The reference's V_R problem (lib/oracle.py:57-66, 175-218) as ONE mixed-integer feasibility
    LP: a copy of the trajectory per vertex of R, x_0 of copy i = v_i, one shared set of mode
    indicators.  Returns a mode sequence feasible at EVERY vertex, or None.  (Which one is the
    solver's choice -- the reference minimises 0 --; the canonical rule of the oracles returns
    the first in enumeration order, so only feasibility and membership can be compared.)
    """
    R = np.asarray(R, dtype=float)
    M = _model_copies(mpc, big_m, R.shape[0])
    for c, v in enumerate(R):
        r = M['blank'](mpc.n_x)
        r[:, M['x0'](c)] = np.eye(mpc.n_x)
        M['add'](r, v, v)
    res = _solve(M, np.zeros(M['nv']))
    return _sequence(mpc, res, M) if res.status == 0 else None


def bar_d_milp(mpc, R, V_bar, eps_a, eps_r, big_m=50.):
    """
    The reference's bar_D problem (lib/oracle.py:101-102, 311-414): bar_E's system (copy 0: the
    parameter is a point of the simplex, the two suboptimality rows) AND V_R's system (copies
    1 .. p+1 at the vertices) with ONE shared mode sequence, in decision form: maximise t.
    Returns (t_max, mode sequence, theta*); (-inf, None, None) when no sequence is feasible at
    every vertex.  The reference's problem is feasible iff t_max >= 0; the canonical rule of the
    oracles returns the sequence of the LARGEST t -- this optimum.
    """
    R = np.asarray(R, dtype=float)
    na = R.shape[0]
    M = _model_copies(mpc, big_m, na + 1, extra_cols=na + 1)
    oa, ot = M['tail'], M['tail'] + na
    r = M['blank'](mpc.n_x)                       # copy 0: x_0 = sum alpha_i v_i
    r[:, M['x0'](0)] = np.eye(mpc.n_x)
    r[:, oa:oa + na] = -R.T
    M['add'](r, np.zeros(mpc.n_x), np.zeros(mpc.n_x))
    r = M['blank'](1)
    r[0, oa:oa + na] = 1.
    M['add'](r, [1.], [1.])
    M['lb'][oa:oa + na] = 0.
    for scale, shift in ((1., eps_a), (1. + eps_r, 0.)):
        r = M['blank'](1)
        r[0, oa:oa + na] = np.asarray(V_bar, dtype=float)
        r[0, :] -= scale * M['cost_of'](0)
        r[0, ot] = -1.
        M['add'](r, [shift], [np.inf])
    for c, v in enumerate(R):                     # copies 1..: feasible at every vertex
        r = M['blank'](mpc.n_x)
        r[:, M['x0'](c + 1)] = np.eye(mpc.n_x)
        M['add'](r, v, v)
    cvec = np.zeros(M['nv'])
    cvec[ot] = -1.
    res = _solve(M, cvec)
    if res.status != 0:
        return -np.inf, None, None
    return float(-res.fun), _sequence(mpc, res, M), res.x[oa:oa + na] @ R
