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


def p_theta_milp(mpc, theta, big_m=50.):
    """(optimal cost, mode sequence) of the mixed-integer LP; (inf, None) if infeasible."""
    assert mpc.cost_type == 'inf'
    N, nx, nu, nm = mpc.N, mpc.n_x, mpc.n_u, mpc.delta_size
    ox = 0                              # x_0 .. x_N
    ou = ox + (N + 1) * nx              # u_0 .. u_{N-1}
    oz = ou + N * nu                    # z_{k,i}
    od = oz + N * nm * nx               # d_{k,i}
    oex = od + N * nm                   # ex_1 .. ex_N
    oeu = oex + N                       # eu_0 .. eu_{N-1}
    nv = oeu + N
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
    r = blank(nx)
    r[:, X(0)] = np.eye(nx)
    add(r, np.asarray(theta, dtype=float), np.asarray(theta, dtype=float))
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
    c = np.zeros(nv)
    c[oex:] = 1.
    integrality = np.zeros(nv)
    integrality[od:oex] = 1
    lb = np.full(nv, -np.inf)
    ub = np.full(nv, np.inf)
    lb[od:oex], ub[od:oex] = 0., 1.
    res = milp(c, constraints=LinearConstraint(np.vstack(rows), np.concatenate(lo),
                                               np.concatenate(hi)),
               integrality=integrality, bounds=Bounds(lb, ub),
               options=dict(mip_rel_gap=1e-10))
    if res.status != 0:
        return np.inf, None
    d = np.rint(res.x[od:oex]).reshape(N, nm)
    return float(res.fun), tuple(int(i) for i in d.argmax(axis=1))
