"""
Phase table of the LDS-resident wide solver (ehm_ipm4.h) from an experimental build:
    python tools/k4_phases.py build      -- lib/libehmpc_k4prof.so: ehm_k4.hip with -DEHM4_PROFILE,
                                            the other objects of the regular build
    EHM_LIB=.../libehmpc_k4prof.so python tools/k4_phases.py [chain|...]   (GPU box)
Cycles are thread 0's (wavefront 0's time line, barrier waits included).
"""
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, 'explicit_hybrid_mpc_amd', 'lib')
PROF = os.path.join(LIBDIR, 'libehmpc_k4prof%s.so' % os.environ.get('K4_TAG', ''))
NAMES = {0: 'residuals, vectors out', 1: 'A^T[u0 u1]', 2: 'merit, reductions, d out',
         3: 'G, Delta (eliminated block)', 4: 'MFMA tiles', 5: 'psi -> beta', 6: 'dense rows prep',
         7: 'mask + rank-one + diagonal', 8: 'panel 0', 9: 'trailing 0', 10: 'panel 1',
         11: 'trailing 1 + panel 2', 12: 'predictor solve', 13: 'A dx', 14: 'step lengths, corrector rhs',
         15: 'A^T corr', 16: 'corrector solve', 17: 'A dx', 18: 'step lengths, update'}
# inside the phases above (both calls of a function added up)
INNER = {20: 'A^T u: K-slices (to the barrier)', 21: 'A^T u: combine in wavefront 0', 22: 'solve: reduce the rhs',
         23: 'solve: forward', 24: 'solve: backward', 25: 'solve: eliminated block, dense rows',
         26: 'solve: psi-form'}


def build():
    from explicit_hybrid_mpc_amd import build as b
    b.build()
    obj = os.path.join(LIBDIR, 'obj_k4prof' + os.environ.get('K4_TAG', ''))
    os.makedirs(obj, exist_ok=True)
    o = os.path.join(obj, 'ehm_k4.o')
    flags = b.FLAGS + ['-DEHM4_PROFILE=1'] + os.environ.get('K4_FLAGS', '').split()
    subprocess.check_call([b._hipcc()] + flags + ['-c', os.path.join(b.SRC_DIR, 'ehm_k4.hip'), '-o', o])
    objs = [x for (x, _, _) in b._objects() if not x.endswith('ehm_k4.o')] + [o]
    objs += [x for x, _, _ in b.HOST_OBJECTS]
    subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', PROF])
    print(PROF)


def run(kind):
    from explicit_hybrid_mpc_amd import engine, examples, _capi
    from tests import helpers
    lib = _capi.load(build_if_missing=False)
    mpc = helpers.make_instance(kind, 0)
    can = mpc.compile()
    gp = engine.GpuProblem(can, 0.05, 0.1)
    rng = np.random.default_rng(5)
    half = examples.theta_box(mpc)
    p = half.size
    R = helpers.random_simplices(mpc, rng, 2048)
    Vbar = gp.solve_ptd(R.reshape(-1, p), can.deltas[0])[0].reshape(R.shape[0], p + 1)
    buf = (ctypes.c_ulonglong * 40)()
    out = {}
    for name, fn in (('slack', lambda: gp.slack(R, Vbar, can.deltas[0])),
                     ('point', lambda: gp.solve_ptd(R[:, 0, :].copy(), can.deltas[0]))):
        fn()
        lib.ehm_k4_profile(buf, 1)
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        lib.ehm_k4_profile(buf, 0)
        v = np.array(list(buf), dtype=np.float64)
        its = max(v[39], 1.0)
        tot = v[:19].sum()
        rows = {NAMES[k]+' [%d]' % k: [round(v[k] / its), round(100 * v[k] / tot, 1)] for k in range(19)}
        inner = {INNER[k] + ' [%d]' % k: round(v[k] / its) for k in sorted(INNER)}
        out[name] = {'batch_ms': round(dt * 1e3, 2), 'iterations_timed': int(its),
                     'cycles_per_iteration': round(tot / its), 'phases_cycles_and_percent': rows,
                     'inside_cycles': inner}
    gp.close()
    return out


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    else:
        print(json.dumps({k: run(k) for k in (sys.argv[1:] or ['chain'])}, indent=1))
