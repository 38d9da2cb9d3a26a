"""Where the shared-block solver (csrc/ehm_ipm2.h) spends its cycles, phase by phase.

Needs a library built with the phase clocks:
    EHM_BUILD_FLAGS=-DEHM2_PROF=1 EHM_BUILD_TAG=prof python -m explicit_hybrid_mpc_amd.build
    EHM_LIB=explicit_hybrid_mpc_amd/lib/libehmpc_prof.so python tools/solver_phases.py [config2]
Runs the headline partition once (persistent frontier kernel) and prints, for the solves of the
launch: cycles per solve and per phase (lane 0's shader clock, summed over the wavefronts), i.e.
wall time of a wavefront inside the phase -- issue + every wait, with the other wavefronts of its
SIMD competing.  ehm_solver_phase_ticks (include/ehmpc.h).
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = ['residuals', 'cols_times(lam, D r_p)', 'eliminated block G, Delta', 'normal matrix (MFMA / blocks)',
         'zero fill, Schur(lds), T-transform', 'dense rows reduced', 'row load + rank-one terms',
         'lu_factor', 'predictor solve', 'rows_times 1', 'step lengths 1 + corrector rhs',
         'cols_times(corrector)', 'corrector solve', 'rows_times 2', 'step lengths 2 + update',
         'convergence test + reductions']


def main():
    import bench
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    workload = sys.argv[1] if len(sys.argv) > 1 else 'config2'
    mpc = bench.make_mpc(workload, 0)
    gp = engine.GpuProblem(mpc.compile(), 1., 1., device=0)
    half = examples.theta_box(mpc)
    V = examples.box_vertices(half)
    abs_frac, eps_r = {'config2q': (0.1, 0.1)}.get(workload, (0.02, 1e-2))
    J_abs, _, _ = gp.solve_pt(abs_frac * V)
    gp.set_eps(float(np.max(J_abs)), eps_r)
    roots, _ = ehm_tools.delaunay_roots(V)

    def ticks():
        out = (ctypes.c_int64 * 24)()
        engine.check(gp._lib.ehm_solver_phase_ticks(gp._handle, ctypes.addressof(out)))
        return np.array(list(out), dtype=np.float64)
    gp.partition(roots, action='ecc', max_nodes=1 << 22, export=False, with_volume=False)
    t0 = ticks()
    info = gp.partition(roots, action='ecc', max_nodes=1 << 22, export=False, with_volume=False)
    t = ticks() - t0
    n = t[23]
    if n == 0:
        print('no phase clocks in this library (build with -DEHM2_PROF=1)')
        return
    tot = t[:16].sum()
    its = info['ipm_iters'] if 'ipm_iters' in info else float('nan')
    print('%s: %d solves, %.0f iterations, %.0f cycles per solve, %.0f per iteration' %
          (workload, n, its, tot / n, tot / max(its, 1)))
    for k, name in enumerate(NAMES):
        print('  %-40s %6.1f %%  %8.0f cycles per iteration' % (name, 100. * t[k] / tot,
                                                                t[k] / max(its, 1)))


if __name__ == '__main__':
    main()
