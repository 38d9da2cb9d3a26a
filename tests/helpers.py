"""Shared helpers for the parity tests (test infrastructure)."""

import numpy as np

from explicit_hybrid_mpc_amd import examples
from oracle.oracle_cpu import OracleCPU
from oracle import geometry


CHAIN_SMALL_SCALE = 0.6      # x_max = (1, 1), T = 0.1, N = 8: every vertex feasible (tested)
PWA_SMALL_SCALE = 0.367     # 0.9 x max feasible scale (0.408, tools/calibrate_configs.py)


def make_instance(kind, seed=0):
    if kind == 'di':
        return examples.double_integrator(3)
    if kind == 'lin':
        return examples.linear_mpc(seed)
    if kind == 'pwa':
        return examples.pwa_mpc(seed)
    if kind == 'chain':
        # config 4 (n_x = 6, n_u = 3, N = 10, box constraints): LPs of 50..57 columns, 360..369
        # rows -- the wide kernels (ehm_k3.hip)
        return examples.integrator_chain_mpc()
    if kind == 'chain_small':
        # same family, n_x = 4, n_u = 2, N = 8: 32 + 4 + 1 = 37 columns, 3-tile normal matrix
        mpc = examples.integrator_chain_mpc(n_axes=2, N=8)
        examples.THETA_SCALE.setdefault(mpc.name, CHAIN_SMALL_SCALE)
        return mpc
    if kind == 'pwa_small':
        # 2 states, 1 input, N=3: 8 commutations -- a hybrid instance whose whole partition
        # the CPU oracle finishes in seconds
        mpc = examples.pwa_mpc(seed=seed, n_x=2, n_u=1, N=3, n_random=4, overlap=0.3)
        examples.THETA_SCALE.setdefault(mpc.name, PWA_SMALL_SCALE)
        return mpc
    raise ValueError(kind)


def eps_a_rule(mpc, abs_frac):
    """examples.create_oracle's eps_a rule (lib/examples.py:42-46) with the CPU oracle."""
    orc = OracleCPU(mpc, 1., 1.)
    V = examples.box_vertices(examples.theta_box(mpc))
    return max(orc.P_theta(abs_frac * v)[2] for v in V)


def roots_of(mpc):
    V = examples.box_vertices(examples.theta_box(mpc))
    return geometry.delaunay_simplices(V)


def random_simplices(mpc, rng, n, scale_lo=-2., scale_hi=0.):
    half = examples.theta_box(mpc)
    p = half.size
    out = []
    for _ in range(n):
        scale = 10 ** rng.uniform(scale_lo, scale_hi)
        ctr = rng.uniform(-1, 1, p) * half * (1 - scale)
        R = ctr + scale * rng.uniform(-1, 1, (p + 1, p)) * half
        out.append(np.clip(R, -half, half))
    return np.array(out)
