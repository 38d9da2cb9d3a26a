"""Shared helpers for the parity tests (test infrastructure)."""

import numpy as np

from explicit_hybrid_mpc_amd import examples
from oracle.oracle_cpu import OracleCPU
from oracle import geometry


def make_instance(kind, seed=0):
    if kind == 'di':
        return examples.double_integrator(3)
    if kind == 'lin':
        return examples.linear_mpc(seed)
    if kind == 'pwa':
        return examples.pwa_mpc(seed)
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
