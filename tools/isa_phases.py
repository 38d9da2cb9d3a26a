"""Static instruction census of a kernel's ISA per solver phase.

Usage: python tools/isa_phases.py file.s kernel_substring
Compile with -DEHM2_ISA_MARKS: csrc/ehm_ipm2.h then leaves a comment "@@PHASE k" where phase k of
ipm_solve ENDS (the phase numbers of tools/solver_phases.py), e.g.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEHM_NP=16 -DEHM_SLOTS=3 -DEHM_PERSIST_MIDFIRST=1 \
          -DEHM2_ISA_MARKS --cuda-device-only -S explicit_hybrid_mpc_amd/csrc/ehm_k2.hip -o /tmp/k2.s  For every phase: instruction counts by class, and the
loops found inside it (label, body length, classes) so that trip counts can be applied by hand.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith('v_mfma'):
        return 'MFMA'
    if op.startswith(('ds_', )):
        return 'LDS'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'VMEM'
    if op.startswith('v_'):
        if 'f64' in op and not op.startswith('v_cmp'):
            return 'VALU_F64'
        if 'readlane' in op or 'readfirstlane' in op:
            return 'VALU_RDLANE'
        if 'writelane' in op:
            return 'VALU_WRLANE'
        if op.startswith(('v_mov', 'v_accvgpr')):
            return 'VALU_MOV'
        return 'VALU_OTHER'
    if op.startswith('s_waitcnt'):
        return 'WAIT'
    if op.startswith('s_nop'):
        return 'NOP'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'BRANCH'
    if op.startswith('s_'):
        return 'SALU'
    return 'OTHER'


def main():
    path, kname = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_Z') and kname in l.split(':')[0] and ':' in l:
            start = i
            break
    if start is None:
        raise SystemExit('kernel not found')
    end = start
    while not lines[end].strip().startswith('.Lfunc_end'):
        end += 1
    body = lines[start:end]
    phases = OrderedDict()
    cur = 'prologue'
    seen = Counter()
    labels = {}
    insts = []      # (idx, phase, op, text)
    for l in body:
        t = l.strip()
        if '@@PHASE' in t:
            name = t.split('@@PHASE')[1].strip()
            seen[name] += 1
            cur = '%s#%d' % (name, seen[name])
            continue
        if re.match(r'^[.\w$]+:', t):
            labels[t.split(':')[0]] = len(insts)
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        op = t.split()[0]
        insts.append((len(insts), cur, op, t))
    for idx, ph, op, t in insts:
        phases.setdefault(ph, Counter())[classify(op)] += 1
    # loops: backward branches
    loops = []
    for idx, ph, op, t in insts:
        if op.startswith(('s_cbranch', 's_branch')):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= idx:
                b = insts[labels[tgt]:idx + 1]
                c = Counter(classify(x[2]) for x in b)
                loops.append((insts[labels[tgt]][1], ph, tgt, len(b), c))
    order = ['VALU_F64', 'VALU_OTHER', 'VALU_MOV', 'VALU_RDLANE', 'VALU_WRLANE', 'MFMA', 'LDS', 'VMEM',
             'SALU', 'WAIT', 'NOP', 'BRANCH', 'OTHER']
    print('%-26s' % 'phase' + ''.join('%8s' % o[-7:] for o in order) + '   total')
    for ph, c in phases.items():
        print('%-26s' % ph + ''.join('%8d' % c.get(o, 0) for o in order) + '%8d' % sum(c.values()))
    print('\nloops (phase of head -> phase of branch, label, body length):')
    for hp, bp, tgt, n, c in loops:
        valu = sum(v for k, v in c.items() if k.startswith('VALU'))
        print('  %-22s %-22s %-12s %5d  valu %4d f64 %4d lds %4d mfma %3d' %
              (hp, bp, tgt, n, valu, c.get('VALU_F64', 0), c.get('LDS', 0), c.get('MFMA', 0)))


if __name__ == '__main__':
    main()
