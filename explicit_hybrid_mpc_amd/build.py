"""
Builds ``explicit_hybrid_mpc_amd/lib/libehmpc.so`` (HIP kernels + C-ABI, include/ehmpc.h)
for gfx950 with hipcc.  hipcc cross-compiles without a GPU, so this also runs in the
CPU-only build container.
"""

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIB_DIR, 'libehmpc.so')
SOURCES = ['ehm_capi.hip']
HEADERS = ['ehm_ipm.h', 'ehm_kernels.h', os.path.join('..', '..', 'include', 'ehmpc.h')]


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: libehmpc.so cannot be built')


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SRC_DIR, s) for s in SOURCES + HEADERS] + [__file__]
    return any(os.path.getmtime(os.path.normpath(d)) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources."""
    if not force and not is_stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
           '-Wno-unused-result']
    cmd += [os.path.join(SRC_DIR, s) for s in SOURCES]
    cmd += ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
