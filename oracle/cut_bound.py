"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

numpy restatement of the tangent-plane bound of the suboptimality-test optimum
(``cut_bound`` in explicit_hybrid_mpc_amd/csrc/ehm_dev.h), with the
gradients of the optimal cost taken from HiGHS' dual solution instead of the kernels'
multipliers.  Not part of the reference: the reference answers every ``bar_E_delta_R`` call
(lib/oracle.py:285-309) with a solver run; the bound is a sufficient condition for the answer
"infeasible => close the leaf" that needs no solver, and this file exists to check that claim
(the bound is never below the true optimum) independently of the device code.
"""

import itertools
import numpy as np
from scipy.optimize import linprog

from .oracle_cpu import HIGHS_OPTIONS


def vertex_gradient(model, theta):
    """(V*, dV*/dtheta) of P_theta_delta at theta: the marginals of the x_0 = theta rows."""
    lp = model.lp_point(theta)
    r = linprog(lp['c'], A_ub=lp['A_ub'], b_ub=lp['b_ub'], A_eq=lp['A_eq'], b_eq=lp['b_eq'],
                bounds=(None, None), method='highs', options=HIGHS_OPTIONS)
    if r.status != 0:
        raise RuntimeError('vertex solve failed')
    return float(r.fun), np.array(r.eqlin.marginals[-len(theta):])


def rows_at_vertices(R, V, g, eps_a, eps_r):
    """f_r(v_j): f_{2i} = Vbar - L_i - eps_a, f_{2i+1} = Vbar - (1+eps_r) L_i, L_i the tangent
    plane of the optimal cost at vertex i."""
    na = R.shape[0]
    rows = []
    for i in range(na):
        Li = V[i] + (R - R[i]) @ g[i]
        rows.append(V - Li - eps_a)
        rows.append(V - (1. + eps_r) * Li)
    return np.array(rows)


def bound_single(rows):
    return float(rows.max(axis=1).min())


def bound_pairs(rows):
    """min over pairs of max over the simplex of min(f_a, f_b): vertices and edge crossings."""
    na = rows.shape[1]
    best = bound_single(rows)
    for a, b in itertools.combinations(range(rows.shape[0]), 2):
        f, h = rows[a], rows[b]
        m = np.max(np.minimum(f, h))
        for u, v in itertools.combinations(range(na), 2):
            du, dv = f[u] - h[u], f[v] - h[v]
            if du * dv < 0:
                s = du / (du - dv)
                m = max(m, f[u] + s * (f[v] - f[u]))
        best = min(best, m)
    return float(best)


def bound_all_cuts(R, V, g, eps_a, eps_r):
    """The exact max over the simplex of min over ALL the functions (a small LP)."""
    na, p = R.shape
    A, b = [], []
    for i in range(na):
        row = np.zeros(na + 2)
        row[:na] = R @ g[i]
        row[na] = -1.
        A.append(row)
        b.append(-(V[i] - g[i] @ R[i]))
    for kappa, e in ((1., eps_a), (1. + eps_r, 0.)):
        row = np.zeros(na + 2)
        row[:na] = -V
        row[na] = kappa
        row[na + 1] = 1.
        A.append(row)
        b.append(-e)
    Aeq = np.zeros((1, na + 2))
    Aeq[0, :na] = 1.
    c = np.zeros(na + 2)
    c[-1] = -1.
    r = linprog(c, A_ub=np.array(A), b_ub=np.array(b), A_eq=Aeq, b_eq=[1.],
                bounds=[(0, None)] * na + [(None, None)] * 2, method='highs')
    return float(-r.fun)


def vertex_gradient_quadratic(model, theta):
    """
    (V*, dV*/dtheta) of a quadratic-cost P_theta_delta (oracle/qp_numpy.py): in the
    uncondensed model the parameter enters only through the rows x_0 = theta (the last n_x
    equalities), so dV*/dtheta = -nu of those rows -- the same quantity the device forms from
    the condensed problem as -S^T lambda + F^T z* + C theta + c1 (envelope theorem).
    """
    from . import qp_numpy
    lp = model.lp_point(theta)
    r = qp_numpy.solve(lp['c'], lp['A_ub'], lp['b_ub'], lp['A_eq'], lp['b_eq'], P=lp['P'])
    if r.status != 0 and max(r.res_p, r.res_d, r.gap) > 1e-9:
        raise RuntimeError('vertex solve failed')
    return float(r.fun), -np.array(r.nu[-len(theta):])
