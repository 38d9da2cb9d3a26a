"""
Synthetic MPC instances for the partitioning hot path (BASELINE.json ``configs``,
SURVEY.md section 8d) and the oracle factory.

Role in the reference: ``lib/examples.py`` -- ``example()`` returns
``(full_set, partition_tree, oracle)`` (lib/examples.py:165-180) and ``create_oracle``
fixes eps_a as the largest optimal cost at the ``abs_frac``-scaled vertices of the set
to partition (lib/examples.py:18-47).  The reference's satellite / pendulum models need
MOSEK, cdd and data files that are not shipped, so the instances here are the seeded
random-polytope MPC problems BASELINE.json names.

Every builder is deterministic in ``seed`` (``numpy.random.default_rng``) and returns a
``PWAMPC`` plus the box ``Theta`` to partition.  The box half-widths are recorded
constants (``THETA_SCALE``): a fraction of the largest centred box whose vertices are all
feasible, computed once with ``tools/calibrate_configs.py``; tests re-check feasibility.
"""

import itertools
import numpy as np

from .mpc_library import PWAMPC

# Fraction of the maximal all-vertices-feasible box that is partitioned.  Staying strictly
# inside keeps every vertex LP strictly feasible (no zero-volume feasible sets).
THETA_SAFETY = 0.9

# name -> half-width scale of the unit box direction, from tools/calibrate_configs.py
THETA_SCALE = {}


def box_vertices(half_widths):
    """All 2^p vertices of the centred box, itertools.product order."""
    half_widths = np.asarray(half_widths, dtype=np.float64)
    signs = np.array(list(itertools.product([-1., 1.], repeat=half_widths.size)))
    return signs * half_widths[None, :]


def double_integrator(N=3, cost='inf'):
    """
    Config 1 (plumbing): 2-state double integrator, one input, single mode,
    A=[[1,T],[0,1]], B=[[T^2/2],[T]], T=1, |x|<=(5,5), |u|<=1, infinity-norm cost.
    """
    T = 1.
    A = np.array([[1., T], [0., 1.]])
    B = np.array([[T * T / 2.], [T]])
    Gx = np.vstack([np.eye(2), -np.eye(2)]) / 5.
    gx = np.ones(4)
    Gu = np.array([[1.], [-1.]])
    gu = np.ones(2)
    mpc = PWAMPC([A], [B], [np.zeros(2)], [None], Gx, gx, Gu, gu,
                 Q=np.eye(2), R=np.eye(1), N=N, name='double_integrator_N%d' % N, cost=cost)
    return mpc


def random_polytope(rng, n_x, n_random):
    """
    State set {x : g_i^T x <= 1}: 2 n_x box rows |x_j| <= 1 plus ``n_random`` random
    half-spaces scaled so that each contains the box 0.5*[-1,1]^n_x with a random margin.
    """
    rows = [np.eye(n_x), -np.eye(n_x)]
    a = rng.standard_normal((n_random, n_x))
    a /= np.linalg.norm(a, axis=1)[:, None]
    support_half_box = 0.5 * np.abs(a).sum(axis=1)
    margin = 1. + rng.uniform(0.1, 0.6, size=n_random)
    rows.append(a / (support_half_box * margin)[:, None])
    G = np.vstack(rows)
    return G, np.ones(G.shape[0])


def linear_mpc(seed=0, n_x=4, n_u=2, N=5, rho=1.05, n_random=8, r_weight=0.1, cost='inf'):
    """
    Config 2: n_x=4, n_u=2, N=5, p=4 linear MPC.  A = rho * M / spectral_radius(M),
    M ~ N(0,1); B ~ N(0,1); random-polytope state set; |u| <= 1;
    cost ||x||_inf + r_weight ||u||_inf, or with ``cost='quadratic'`` the quadratic cost of
    the same weights (sum x'x + r_weight u'u, the form of lib/mpc_library.py:515-517).
    """
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n_x, n_x))
    A = rho * M / np.max(np.abs(np.linalg.eigvals(M)))
    B = rng.standard_normal((n_x, n_u))
    Gx, gx = random_polytope(rng, n_x, n_random)
    Gu = np.vstack([np.eye(n_u), -np.eye(n_u)])
    gu = np.ones(2 * n_u)
    mpc = PWAMPC([A], [B], [np.zeros(n_x)], [None], Gx, gx, Gu, gu,
                 Q=np.eye(n_x), R=r_weight * np.eye(n_u), N=N,
                 name='linear_nx%d_nu%d_N%d_seed%d' % (n_x, n_u, N, seed), cost=cost)
    return mpc


def pwa_mpc(seed=0, n_x=4, n_u=2, N=5, rho=1.05, n_random=8, r_weight=0.1, kink=0.3,
            overlap=0.05, cost='inf'):
    """
    Config 3: two-mode PWA system, mode 0 admissible on x_1 >= -overlap and mode 1 on
    x_1 <= +overlap (pattern of lib/mpc_library.py:530-541 with a fixed commutation
    turned into exact mode dynamics).  The two dynamics matrices differ only in their
    first column.  Inside the guard band |x_1| <= overlap either mode may be commanded
    (the commutation is a decision, as for the reference's thruster-set choice,
    lib/mpc_library.py:207-215); the band has non-empty interior so that the
    feasible-commutation partition (lib/worker.py:241-291) terminates: with regions
    that only touch, a simplex straddling the surface need never be cut exactly on it.
    """
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n_x, n_x))
    A0 = rho * M / np.max(np.abs(np.linalg.eigvals(M)))
    B = rng.standard_normal((n_x, n_u))
    dcol = kink * rng.standard_normal(n_x)
    A1 = A0.copy()
    A1[:, 0] += dcol
    Gx, gx = random_polytope(rng, n_x, n_random)
    Gu = np.vstack([np.eye(n_u), -np.eye(n_u)])
    gu = np.ones(2 * n_u)
    e1 = np.zeros((1, n_x))
    e1[0, 0] = 1.
    regions = [(-e1, overlap * np.ones(1)), (e1, overlap * np.ones(1))]
    mpc = PWAMPC([A0, A1], [B, B], [np.zeros(n_x)] * 2, regions, Gx, gx, Gu, gu,
                 Q=np.eye(n_x), R=r_weight * np.eye(n_u), N=N,
                 name='pwa_nx%d_nu%d_N%d_seed%d' % (n_x, n_u, N, seed), cost=cost)
    return mpc


def pwa4_mpc(seed=0, n_x=8, n_u=3, N=4, rho=1.02, kink=0.25, overlap=0.05, r_weight=0.1):
    """
    The SHAPE of config 5 (n_x = 8, n_u = 3, 4 integer modes): the four modes are the sign
    patterns of (x_1, x_2), each admissible on its quadrant widened by ``overlap``; the dynamics
    matrices differ in their first two columns.  With N = 4 its 4^4 = 256 commutations can be
    enumerated (the engine's limit); config 5 proper (N = 8: 65 536 of them) needs the
    branch-and-bound over mode prefixes of DESIGN.md section 7c.
    """
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n_x, n_x))
    A0 = rho * M / np.max(np.abs(np.linalg.eigvals(M)))
    B = rng.standard_normal((n_x, n_u))
    d1 = kink * rng.standard_normal(n_x)
    d2 = kink * rng.standard_normal(n_x)
    A, regions = [], []
    for s1 in (1., -1.):
        for s2 in (1., -1.):
            Ai = A0.copy()
            if s1 < 0:
                Ai[:, 0] += d1
            if s2 < 0:
                Ai[:, 1] += d2
            A.append(Ai)
            H = np.zeros((2, n_x))
            H[0, 0], H[1, 1] = -s1, -s2          # s1 x_1 >= -overlap, s2 x_2 >= -overlap
            regions.append((H, overlap * np.ones(2)))
    Gx = np.vstack([np.eye(n_x), -np.eye(n_x)])
    gx = np.ones(2 * n_x)
    Gu = np.vstack([np.eye(n_u), -np.eye(n_u)])
    gu = np.ones(2 * n_u)
    mpc = PWAMPC(A, [B] * 4, [np.zeros(n_x)] * 4, regions, Gx, gx, Gu, gu,
                 Q=np.eye(n_x), R=r_weight * np.eye(n_u), N=N,
                 name='pwa4_nx%d_nu%d_N%d_seed%d' % (n_x, n_u, N, seed))
    THETA_SCALE.setdefault(mpc.name, 0.2)
    return mpc


def integrator_chain_mpc(n_axes=3, N=10, T=0.1, x_max=(1., 1.), u_max=1., r_weight=0.1):
    """
    Config 4: n_x = 2*n_axes "quadrotor" made of independent double integrators with box
    constraints (same shape as the reference's SatelliteXYZ, lib/mpc_library.py:386-405).
    """
    A1 = np.array([[1., T], [0., 1.]])
    B1 = np.array([[T * T / 2.], [T]])
    A = np.kron(np.eye(n_axes), A1)
    B = np.kron(np.eye(n_axes), B1)
    n_x, n_u = 2 * n_axes, n_axes
    scale = np.tile(np.asarray(x_max, dtype=np.float64), n_axes)
    Gx = np.vstack([np.eye(n_x), -np.eye(n_x)]) / np.concatenate([scale, scale])[:, None]
    gx = np.ones(2 * n_x)
    Gu = np.vstack([np.eye(n_u), -np.eye(n_u)]) / u_max
    gu = np.ones(2 * n_u)
    mpc = PWAMPC([A], [B], [np.zeros(n_x)], [None], Gx, gx, Gu, gu,
                 Q=np.eye(n_x), R=r_weight * np.eye(n_u), N=N,
                 name='chain_nx%d_N%d' % (n_x, N))
    return mpc


def theta_box(mpc, scale=None):
    """
    Half-widths of the centred box Theta that is partitioned: ``scale`` times the
    bounding half-widths of the state set along each axis (1 for the |x_j|<=1 rows).
    """
    if scale is None:
        scale = THETA_SCALE[mpc.name]
    # the state sets built above all contain the rows +-e_j / xmax_j
    half = np.empty(mpc.n_x)
    for j in range(mpc.n_x):
        e = np.zeros(mpc.n_x)
        e[j] = 1.
        hit = [mpc.gx[i] / mpc.Gx[i, j] for i in range(mpc.Gx.shape[0])
               if mpc.Gx[i, j] > 0 and np.count_nonzero(mpc.Gx[i]) == 1]
        half[j] = min(hit)
    return scale * half


THETA_SCALE.update({
    # output of tools/calibrate_configs.py (max feasible scale x THETA_SAFETY)
    'double_integrator_N3': 0.4199,   # max feasible 0.4666
    'linear_nx4_nu2_N5_seed0': 0.5063,   # max feasible 0.5626
    'linear_nx4_nu2_N5_seed1': 0.6718,   # max feasible 0.7465
    'linear_nx4_nu2_N5_seed2': 0.6008,   # max feasible 0.6677
    'linear_nx4_nu2_N5_seed3': 0.3595,   # max feasible 0.3995
    'linear_nx4_nu2_N5_seed4': 0.4459,   # max feasible 0.4955
    'pwa_nx4_nu2_N5_seed0': 0.5176,   # max feasible 0.5751
    'pwa_nx4_nu2_N5_seed1': 0.5574,   # max feasible 0.6194
    'pwa_nx4_nu2_N5_seed2': 0.4866,   # max feasible 0.5408
    'chain_nx6_N10': 0.6590,   # max feasible 0.7323
})


# ---------------------------------------------------------------------------------------
# oracle factory and example dispatcher (lib/examples.py:18-47, 165-180)
# ---------------------------------------------------------------------------------------
def create_oracle(mpc, set_vrep, abs_frac, abs_err, rel_err, device=0):
    """
    Same contract as lib/examples.py:18-47: if ``abs_err`` is None it becomes the largest
    optimal cost of P_theta over the ``abs_frac``-scaled vertices of the set (evaluated as
    one batched GPU call instead of 2^p MOSEK solves).
    """
    from .oracle import Oracle
    oracle = Oracle(mpc, eps_a=1., eps_r=1., device=device)
    if abs_err is None:
        J, _, didx = oracle.gpu.solve_pt(abs_frac * np.asarray(set_vrep, dtype=np.float64))
        if (didx < 0).any():
            raise RuntimeError('scaled vertex of the set is infeasible')
        abs_err = float(np.max(J))
    oracle.eps_a = abs_err
    oracle.eps_r = rel_err
    oracle.gpu.set_eps(abs_err, rel_err)
    return oracle


def satellite_z(N=4):
    """
    The reference's ``SatelliteZ`` law (lib/mpc_library.py:221-272): robust CWH z-axis
    control with an off/on-with-minimum-impulse input and quadratic cost; the partitioned
    set is the full state-error box (lib/examples.py:75-86, ``satellite_z_example``).
    """
    from .mpc_library import SatelliteZ
    mpc = SatelliteZ(N)
    THETA_SCALE.setdefault(mpc.name, 1.0)
    return mpc


EXAMPLES = {
    'cwh_z': lambda: satellite_z(4),          # make_jobs.sh:60-61 (EXAMPLE=cwh_z, MPC_N=4)
    'double_integrator': lambda: double_integrator(3),
    'linear': lambda: linear_mpc(0),
    'pwa': lambda: pwa_mpc(0),
}


def example(name, abs_frac=0.5, abs_err=None, rel_err=2.0, device=0):
    """
    (full_set, partition, oracle) like lib/examples.py:165-180: the vertices of the set to
    partition, its Delaunay pre-partition as a right-spine Tree, and the oracle.
    """
    from . import tools
    mpc = EXAMPLES[name]()
    full_set = box_vertices(theta_box(mpc))
    partition, _, _ = tools.delaunay(full_set)
    oracle = create_oracle(mpc, full_set, abs_frac, abs_err, rel_err, device=device)
    return full_set, partition, oracle
