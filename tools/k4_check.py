"""
A/B run of the two wide kernel families on the same batches (diagnostic, GPU box):
  python tools/k4_check.py            -- runs itself twice (EHM_K4=0: streaming kernels of
                                         ehm_k3.hip; default: the LDS-resident ehm_k4.hip),
                                         compares optima / verdicts / iteration counts and the
                                         batch times, prints one JSON object.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def batches(kind):
    from explicit_hybrid_mpc_amd import engine, examples
    from tests import helpers
    if kind == 'pwa8':
        mpc = examples.pwa_mpc(seed=0, n_x=4, n_u=2, N=8)
    else:
        mpc = helpers.make_instance(kind, 0)
    can = mpc.compile()
    gp = engine.GpuProblem(can, 0.05, 0.1)
    rng = np.random.default_rng(5)
    half = examples.theta_box(mpc) * (0.45 if kind == 'pwa8' else 1.0)
    p = half.size
    out = {}
    nrep = 2048
    theta = rng.uniform(-1, 1, (nrep, p)) * half
    delta = can.deltas[0] if can.n_delta == 1 else can.deltas[rng.integers(can.n_delta, size=nrep) % 2 * (can.n_delta - 1)]
    t0 = time.perf_counter()
    J, u0, st, it = gp.solve_ptd(theta, delta)
    out['t_point'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    J, u0, st, it = gp.solve_ptd(theta, delta)
    out['t_point2'] = time.perf_counter() - t0
    out['J'] = J; out['u0'] = u0; out['st'] = st; out['it'] = it
    feas, tau = gp.feasible_ptd(theta * 2.0, delta)
    out['feas'] = feas; out['tau'] = tau
    if can.n_delta == 1:
        R = helpers.random_simplices(mpc, rng, 1024)
        Vbar, stv = gp.solve_ptd(R.reshape(-1, p), can.deltas[0])[0::2]
        Vbar = Vbar.reshape(R.shape[0], p + 1)
        out['Vbar'] = Vbar
        t0 = time.perf_counter()
        t, alpha, sts = gp.slack(R, Vbar, can.deltas[0])
        out['t_slack'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        t, alpha, sts = gp.slack(R, Vbar, can.deltas[0])
        out['t_slack2'] = time.perf_counter() - t0
        out['tstar'] = t; out['alpha'] = alpha; out['sts'] = sts
        Jmin, st2 = gp.min_simplex(R, can.deltas[0])
        out['Jmin'] = Jmin; out['st2'] = st2
    out['stats'] = json.dumps({k: (int(v) if np.isscalar(v) else str(v)) for k, v in gp.stats().items()}
                              if hasattr(gp, 'stats') else {})
    gp.close()
    return out


def child(mode, kind, path):
    out = batches(kind)
    np.savez(path, **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(sys.argv[2], sys.argv[3], sys.argv[4])
        return
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    report = {}
    for kind in (sys.argv[1:] or ['chain_small', 'chain', 'pwa8']):
        res = {}
        for mode in ('k3', 'k4'):
            env = dict(os.environ)
            env['EHM_K4'] = '0' if mode == 'k3' else '1'
            path = os.path.join(ROOT, 'gpurun_out', 'k4_check_%s_%s.npz' % (kind, mode))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', mode, kind, path],
                               env=env, capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                res[mode] = None
                report[kind + '_' + mode + '_error'] = (r.stderr or r.stdout)[-1500:]
                continue
            res[mode] = dict(np.load(path))
        a, b = res.get('k3'), res.get('k4')
        if a is None or b is None:
            continue
        rep = {}
        for key in ('J', 'tau', 'tstar', 'Jmin', 'u0', 'alpha'):
            if key in a:
                ok = np.isfinite(a[key]) & np.isfinite(b[key])
                den = 1 + np.abs(a[key])
                rep['max_rel_' + key] = float(np.max(np.abs(a[key] - b[key])[ok] / den[ok])) if ok.any() else None
                rep['nonfinite_' + key] = [int((~np.isfinite(a[key])).sum()), int((~np.isfinite(b[key])).sum())]
        for key in ('st', 'sts', 'st2'):
            if key in a:
                rep['nonzero_' + key] = [int((a[key] != 0).sum()), int((b[key] != 0).sum())]
        if 'feas' in a:
            rep['feas_equal'] = bool(np.array_equal(a['feas'], b['feas']))
            rep['feas_count'] = [int(a['feas'].sum()), int(b['feas'].sum())]
        rep['iters_mean'] = [float(a['it'].mean()), float(b['it'].mean())]
        for key in ('t_point', 't_point2', 't_slack', 't_slack2'):
            if key in a:
                rep[key + '_ms'] = [round(float(a[key]) * 1e3, 2), round(float(b[key]) * 1e3, 2)]
        report[kind] = rep
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
