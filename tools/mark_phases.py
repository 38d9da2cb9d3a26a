"""Copies csrc/ into a scratch directory with  asm volatile("; @@PHASE name")  markers at the phase
boundaries of ipm_solve (csrc/ehm_ipm2.h), for tools/isa_phases.py.

    python tools/mark_phases.py /tmp/isa/src
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEHM_NP=16 -DEHM_SLOTS=3 -DEHM_PERSIST_MIDFIRST=1 \
          -I include --cuda-device-only -S /tmp/isa/src/ehm_k2.hip -o /tmp/isa/k2.s
    python tools/isa_phases.py /tmp/isa/k2.s k2_lcss_decide
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'explicit_hybrid_mpc_amd', 'csrc')

MARKS = [
    ("        // ---- residuals ---", "residuals"),
    ("        double atl, atdr;", "cols_times_two"),
    ("        const double cjj = (lane < n) ? W.c[lane] : 0.0;", "convergence"),
    ("        // ---- normal matrix and its factorisation", "dvec"),
    ("        form_normal_matrix(S, W, W.vm0, W.ub, lane);", "form"),
    ("        dense_prep(S, W, W.ub, lane);", "dense_prep"),
    ("        double row[NP];", "rowload"),
    ("        lu_factor(row, W, lane);", "lu_factor"),
    ("        // ---- predictor ---", "predictor_solve"),
    ("        double adx[SLOTS];\n        rows_times(S, W, lane, W.t, adx);", "rows_times1"),
    ("        double ds_a[SLOTS], dl_a[SLOTS];", "steplen1"),
    ("        // ---- corrector ---", "corrector_cols"),
    ("        dxj = solve_full(row, S, W, rinv_l, rhs, lane);", "corrector_solve"),
    ("        rows_times(S, W, lane, W.t, adx);\n        double ds[SLOTS], dl[SLOTS];", "rows_times2"),
    ("        double ds[SLOTS], dl[SLOTS];", "steplen2"),
    ("    wsync();\n    if (gout) {", "endloop"),
]


def main():
    dst = sys.argv[1]
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    shutil.copytree(CSRC, dst)
    path = os.path.join(dst, 'ehm_ipm2.h')
    s = open(path).read()
    for anchor, name in MARKS:
        assert s.count(anchor) >= 1, anchor
        s = s.replace(anchor, '        asm volatile("; @@PHASE %s");\n%s' % (name, anchor), 1)
    open(path, 'w').write(s)


if __name__ == '__main__':
    main()
