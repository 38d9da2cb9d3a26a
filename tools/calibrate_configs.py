"""
Offline helper: find, for each synthetic MPC instance, the largest centred box (in units
of the state-set bounding half-widths) whose 2^p vertices all admit a feasible
commutation, by bisection with the CPU oracle, and print the table that
``explicit_hybrid_mpc_amd/examples.py:THETA_SCALE`` records (max scale x THETA_SAFETY).

Uses oracle/ (test infrastructure); it is a calibration tool, not part of the product.
"""

import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from explicit_hybrid_mpc_amd import examples  # noqa: E402
from oracle.oracle_cpu import OracleCPU  # noqa: E402


def max_feasible_scale(mpc, tol=1e-4):
    orc = OracleCPU(mpc, 1., 1.)

    def ok(scale):
        half = examples.theta_box(mpc, scale)
        return all(orc.P_theta(v, check_feasibility=True)
                   for v in examples.box_vertices(half))
    lo, hi = 0., 1.
    if ok(hi):
        return hi
    while hi - lo > tol:
        mid = 0.5 * (lo + hi)
        if ok(mid):
            lo = mid
        else:
            hi = mid
    return lo


def main():
    instances = [examples.double_integrator(3)]
    instances += [examples.linear_mpc(seed=s) for s in range(5)]
    instances += [examples.pwa_mpc(seed=s) for s in range(3)]
    instances += [examples.integrator_chain_mpc()]
    for mpc in instances:
        s = max_feasible_scale(mpc)
        print("    %r: %.4f,   # max feasible %.4f" %
              (mpc.name, np.floor(1e4 * s * examples.THETA_SAFETY) / 1e4, s))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
