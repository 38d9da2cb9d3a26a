#!/usr/bin/env python
"""
On-device cross-check of the two kernel generations (needs a GPU; run through gpurun):
self test of the wave primitives, batched oracles of every LP kind (generation 2 against
generation 1 on the same inputs), a complete partition with both, and timings.
Prints everything; exit code 0 only if all comparisons pass.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from explicit_hybrid_mpc_amd import engine, examples          # noqa: E402
from explicit_hybrid_mpc_amd import tools as ehm_tools        # noqa: E402

ok = True


def report(name, good, detail=''):
    global ok
    ok = ok and bool(good)
    print('%-34s %s  %s' % (name, 'ok  ' if good else 'FAIL', detail))
    sys.stdout.flush()


def relerr(a, b):
    return float(np.max(np.abs(a - b) / (1. + np.abs(b)))) if len(a) else 0.


def main():
    st = engine.selftest()
    exp = np.array([1072., 99., 25., 1. / 3., -1.])
    bad = np.where(np.abs(st - exp).max(axis=1) > 1e-12)[0]
    report('selftest (%d instances)' % len(st), len(bad) == 0,
           '' if len(bad) == 0 else 'first bad row %d: %s' % (bad[0], st[bad[0]]))

    mpc = examples.linear_mpc(seed=0)
    can = mpc.compile()
    print('config 2: n=%d m=%d p=%d n_delta=%d' % (can.n, can.m, can.p, can.n_delta))
    gp = engine.GpuProblem(can, 0.0324, 1e-2, device=0)
    rng = np.random.default_rng(0)
    half = examples.theta_box(mpc)
    V = examples.box_vertices(half)
    roots, _ = ehm_tools.delaunay_roots(V)

    def both(fn):
        gp.set_solver(1)
        a = fn()
        gp.set_solver(2)
        b = fn()
        return a, b

    theta = rng.uniform(-1, 1, (3000, can.p)) * half
    (J1, u1, s1, i1), (J2, u2, s2, i2) = both(lambda: gp.solve_ptd(theta))
    report('P_theta_delta J', relerr(J2, J1) < 1e-8 and (s2 == 0).all(),
           'max rel %.2e, iters v1 %.2f v2 %.2f, stalled %d' %
           (relerr(J2, J1), i1.mean(), i2.mean(), int((s2 != 0).sum())))
    theta_f = rng.uniform(-1.6, 1.6, (2000, can.p)) * half
    (f1, t1), (f2, t2) = both(lambda: gp.feasible_ptd(theta_f))
    report('feasibility', (f1 == f2).all() and relerr(t2, t1) < 1e-7,
           'feasible %d / %d, max tau diff %.2e' % (f1.sum(), len(f1), relerr(t2, t1)))

    # simplices at several depths with their vertex costs
    Rs = []
    for k in range(600):
        R = roots[rng.integers(len(roots))].copy()
        for _ in range(rng.integers(0, 14)):
            S1, S2, _ = ehm_tools.split_along_longest_edge(R)
            R = S1 if rng.random() < 0.5 else S2
        Rs.append(R)
    Rs = np.array(Rs)
    gp.set_solver(1)
    Vb = gp.solve_ptd(Rs.reshape(-1, can.p))[0].reshape(len(Rs), can.p + 1)
    (ta, aa, sa), (tb, ab, sb) = both(lambda: gp.slack(Rs, Vb))
    report('slack t*', relerr(tb, ta) < 1e-7 and (sb == 0).all(),
           'max rel %.2e, decisions equal %s' % (relerr(tb, ta), bool(((ta >= 0) == (tb >= 0)).all())))
    th_a = np.einsum('kv,kvp->kp', aa, Rs)
    th_b = np.einsum('kv,kvp->kp', ab, Rs)
    report('slack alpha (sum, range)', np.abs(ab.sum(1) - 1).max() < 1e-9 and ab.min() > -1e-7,
           'theta* max diff %.2e (maximiser need not be unique)' % np.abs(th_a - th_b).max())
    (ma, msa), (mb, msb) = both(lambda: gp.min_simplex(Rs))
    report('min over simplex', relerr(mb, ma) < 1e-8 and (msb == 0).all(),
           'max rel %.2e' % relerr(mb, ma))

    # complete partitions
    for abs_frac, name in ((0.25, 'small'), (0.08, 'medium')):
        gp.set_solver(2)
        eps_a = float(np.max(gp.solve_pt(abs_frac * V)[0]))
        gp.set_eps(eps_a, 1e-2)
        t0 = time.perf_counter()
        gp.set_solver(1)
        f1 = gp.partition(roots, action='ecc')
        t1 = time.perf_counter()
        gp.set_solver(2)
        gp.set_option('decide_full', 1)
        f2 = gp.partition(roots, action='ecc')
        t2 = time.perf_counter()
        gp.set_option('decide_full', 0)
        f3 = gp.partition(roots, action='ecc')
        same3 = (f1.n_nodes == f3.n_nodes and np.array_equal(f1.vertices, f3.vertices) and
                 np.array_equal(f1.left, f3.left) and np.array_equal(f1.flags & 1, f3.flags & 1))
        report('partition %s sign-only decide' % name, same3,
               'nodes %d, margin bound %.2e, iters/LP %.2f (full %.2f)' %
               (f3.n_nodes, f3.info['min_margin'],
                f3.info['decide_iters'] / max(f3.info['decide_solves'], 1),
                f2.info['decide_iters'] / max(f2.info['decide_solves'], 1)))
        same = (f1.n_nodes == f2.n_nodes and np.array_equal(f1.vertices, f2.vertices) and
                np.array_equal(f1.left, f2.left) and np.array_equal(f1.flags & 1, f2.flags & 1))
        dc = relerr(f2.vertex_costs.ravel(), f1.vertex_costs.ravel()) if same else np.nan
        report('partition %s (%d nodes)' % (name, f1.n_nodes), same and dc < 1e-8,
               'v2 nodes %d, vertex cost rel %.2e, margin %.2e / %.2e, wall v1 %.3fs v2 %.3fs' %
               (f2.n_nodes, dc, f1.info['min_margin'], f2.info['min_margin'], t1 - t0, t2 - t1))
        for f, g in ((f1, 1), (f2, 2)):
            i = f.info
            print('   gen %d: device %.4fs decide %.4fs expand %.4fs  LP %d  iters/LP %.2f' %
                  (g, i['device_seconds'], i['decide_seconds'], i['expand_seconds'],
                   i['lp_solves'], i['ipm_iters'] / max(i['lp_solves'], 1)))

    # bench-size partition, timing only
    eps_a = float(np.max(gp.solve_pt(0.03 * V)[0]))
    gp.set_eps(eps_a, 1e-2)
    for g, full in ((2, 1), (1, 1), (2, 1), (2, 0), (2, 0)):
        gp.set_solver(g)
        gp.set_option('decide_full', full)
        print('decide_full=%d ' % full, end='')
        t0 = time.perf_counter()
        i = gp.partition(roots, action='ecc', max_nodes=1 << 22, export=False, with_volume=False)
        dt = time.perf_counter() - t0
        print('bench partition gen %d: %.3fs wall, device %.3fs (decide %.3f, expand %.3f), %d nodes, '
              '%d LP, %.2f it/LP, %.3g LP/s, margin %.2e' %
              (g, dt, i['device_seconds'], i['decide_seconds'], i['expand_seconds'], i['n_nodes'],
               i['lp_solves'], i['ipm_iters'] / i['lp_solves'], i['lp_solves'] / dt,
               i['min_margin']))
        sys.stdout.flush()
    gp.close()
    print('ALL OK' if ok else 'SOME CHECKS FAILED')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
