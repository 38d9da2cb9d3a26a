"""Kernel time per LP of the batched oracles for one law, with and without the eliminated columns.

    python tools/lp_microbench.py [linear_mpc|pwa_mpc|pwa4_mpc] [n_simplices]

Times ehm_slack_batch (suboptimality-test LPs, full accuracy) and ehm_solve_ptd_batch (point LPs)
on random simplices / parameters of the law's box by the library's own HIP events
(ehm_counters.batch_seconds), once per setting of EHM_SPARSE (a child process each: the setting
is read when the problem handle is created).  With EHM_LIB pointing at a phase-clock build
(tools/solver_phases.py) the per-phase cycles of those solves are printed as well.
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(law, n):
    from explicit_hybrid_mpc_amd import engine, examples
    mpc = getattr(examples, law)()
    can = mpc.compile()
    gp = engine.GpuProblem(can, 0.05, 0.01, device=0)
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(0)
    p = can.p
    centre = (rng.random((n, 1, p)) - .5) * half
    R = centre + 0.05 * (rng.random((n, p + 1, p)) - .5) * half
    nd = can.G.shape[0]
    delta = np.asarray(can.deltas)[rng.integers(0, nd, size=n)] if nd > 1 else None
    th = R.reshape(-1, p)
    dpt = None if delta is None else np.repeat(delta, p + 1, axis=0)
    out = {}
    for rep in range(2):        # first pass warms the instance up
        s0 = gp.stats()
        J, _, st, _ = gp.solve_ptd(th, dpt)
        s1 = gp.stats()
        Vb = np.where(np.isfinite(J), J, 0.).reshape(n, p + 1)
        t, _, st2 = gp.slack(R, Vb, delta)
        s2 = gp.stats()
        out = {'point_us_per_lp': 1e6 * (s1['batch_seconds'][0] - s0['batch_seconds'][0]) / len(th),
               'slack_us_per_lp': 1e6 * (s2['batch_seconds'][1] - s1['batch_seconds'][1]) / n,
               'point_iters': (s1['ipm_iters'] - s0['ipm_iters']) / len(th),
               'slack_iters': (s2['ipm_iters'] - s1['ipm_iters']) / n,
               'feasible_points': float(np.isfinite(J).mean()), 'n': n, 'p': p, 'n_z': can.n,
               'm': can.m}
    ph = (ctypes.c_int64 * 24)()
    if gp._lib.ehm_solver_phase_ticks(gp._handle, ctypes.addressof(ph)) == 0 and ph[23]:
        out['phase_cycles'] = [int(v) for v in ph]
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        return child(sys.argv[2], int(sys.argv[3]))
    law = sys.argv[1] if len(sys.argv) > 1 else 'linear_mpc'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    for sparse in ('1', '0'):
        env = dict(os.environ, EHM_SPARSE=sparse)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', law, str(n)],
                           env=env, capture_output=True, text=True)
        line = r.stdout.strip().split('\n')[-1] if r.stdout.strip() else r.stderr[-400:]
        print('EHM_SPARSE=%s %s %s' % (sparse, law, line))


if __name__ == '__main__':
    main()
