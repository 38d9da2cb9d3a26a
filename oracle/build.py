"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

Builds the oracle's C restatement (``oracle/csrc/*.c`` -> ``oracle/_build/``) with gcc.
The reference itself is pure Python (no C/C++ sources), so there is
nothing to compile into ``oracle/_ref``; the live reference check for the geometry half
is done by importing ``/root/reference/lib/tools.py`` in
``tests/golden/make_geometry_golden.py`` (build container only).
"""

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'libgeom_ref.so')


def build(force=False):
    src = os.path.join(HERE, 'csrc', 'geom_ref.c')
    if (not force and os.path.exists(LIB) and
            os.path.getmtime(LIB) >= os.path.getmtime(src)):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', src, '-o', LIB, '-lm']
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True))
