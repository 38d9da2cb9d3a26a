#!/usr/bin/env python
"""
Timing of the wide kernels on BASELINE.json config 4 (n_x = 6, n_u = 3, N = 10, box
constraints; LPs of 50..57 columns x 360..369 rows): complete partition of the 652 Delaunay
roots of the 6-D box, device times per kernel.  Needs a GPU (run through gpurun).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from explicit_hybrid_mpc_amd import engine, examples          # noqa: E402
from explicit_hybrid_mpc_amd import tools as ehm_tools        # noqa: E402


def main():
    abs_frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
    eps_r = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
    mpc = examples.integrator_chain_mpc()
    can = mpc.compile()
    print('config 4: n=%d m=%d p=%d' % (can.n, can.m, can.p))
    gp = engine.GpuProblem(can, 1., 1., device=0)
    V = examples.box_vertices(examples.theta_box(mpc))
    roots, _ = ehm_tools.delaunay_roots(V)
    eps_a = float(np.max(gp.solve_pt(abs_frac * V)[0]))
    gp.set_eps(eps_a, eps_r)
    print('%d roots, eps_a %.4f (abs_frac %.3f), eps_r %.3f' % (len(roots), eps_a, abs_frac, eps_r))
    rng = np.random.default_rng(0)
    theta = rng.uniform(-1, 1, (8192, can.p)) * examples.theta_box(mpc)
    for rep in range(2):
        t0 = time.perf_counter()
        J, _, st, it = gp.solve_ptd(theta)
        dt = time.perf_counter() - t0
    print('P_theta_delta batch: %d LPs in %.4fs = %.3g LP/s, %.2f it/LP, stalled %d' %
          (len(theta), dt, len(theta) / dt, it.mean(), int((st != 0).sum())))
    for full in (0, 0, 1):
        gp.set_option('decide_full', full)
        t0 = time.perf_counter()
        i = gp.partition(roots, action='ecc', max_nodes=1 << 22, export=False, with_volume=False)
        dt = time.perf_counter() - t0
        print('decide_full=%d partition: %.3fs wall, device %.3fs (decide %.3f, expand %.3f), %d nodes, '
              '%d regions, %d LP, %.2f it/LP, %.3g LP/s, margin %.2e, stalled %d' %
              (full, dt, i['device_seconds'], i['decide_seconds'], i['expand_seconds'], i['n_nodes'],
               i['n_closed'], i['lp_solves'], i['ipm_iters'] / i['lp_solves'], i['lp_solves'] / dt,
               i['min_margin'], gp.stats()['stalled']))
        sys.stdout.flush()
    # phase timers of an experimental build (EHM_BUILD_FLAGS=-DEHM3_PROFILE)
    import ctypes
    from explicit_hybrid_mpc_amd import _capi
    lib = _capi.load()
    buf = (ctypes.c_ulonglong * 32)()
    if lib.ehm_k3_profile_2(buf, 0) == 1:
        names = ['residuals', 'cols_times x2', 'reduce+d', 'normal matrix', 'row load + LU',
                 'solve 1', 'barrier', 'rows_times 1', 'predictor', 'cols_times x1',
                 'solve 2', 'rows_times 2', 'update', 'nm: tiles', 'nm: reduce', 'nm: transform']
        tot = float(sum(buf[k] for k in range(16)))
        for k, nm in enumerate(names):
            print('  %-16s %6.2f %%' % (nm, 100. * buf[k] / tot))
        if buf[28]:
            print('  tile loop cycles per wavefront:', [int(buf[20 + w] * 4 / buf[28]) for w in range(4)],
                  ' SIMD masks:', [int(buf[24 + w]) for w in range(4)])
        if buf[17]:
            print('  shader clock / wall clock (100 MHz) = %.2f  ->  %.0f MHz' %
                  (buf[16] / buf[17], 100. * buf[16] / buf[17]))
    gp.close()


if __name__ == '__main__':
    main()
