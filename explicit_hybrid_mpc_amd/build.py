"""
Builds ``explicit_hybrid_mpc_amd/lib/libehmpc.so`` (HIP kernels + C-ABI, include/ehmpc.h)
for gfx950 with hipcc.  hipcc cross-compiles without a GPU, so this also runs in the
CPU-only build container.

Objects: ``ehm_capi.hip`` (C-ABI, host orchestration, generation-1 kernels), one
instance of ``ehm_k2.hip`` per (column capacity NP, row slots) pair -- the solver keeps a
row of the normal matrix and the LP's row vectors in registers, so both are compile-time
sizes -- and the wide kernels ``ehm_k3.hip`` per row capacity.  The objects are compiled
in parallel and cached under ``lib/obj``.
"""

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
OBJ_DIR = os.path.join(LIB_DIR, 'obj')
LIB = os.path.join(LIB_DIR, 'libehmpc.so')
def _headers():
    """Every header under csrc/ (each is included by some object: derived from the directory, so a
    new header cannot be forgotten) + the public C-ABI header."""
    hs = sorted(f for f in os.listdir(SRC_DIR) if f.endswith('.h'))
    return hs + [os.path.join('..', '..', 'include', 'ehmpc.h')]


HEADERS = _headers()
# must match EHM_K2_ALL in ehm_capi.hip
K2_NPS = (8, 12, 16, 20, 24, 28, 32)
K2_SLOTS = (1, 2, 3, 4)
# instances with the quadratic block (-DEHM2_QUAD=1); must match EHM_K2Q_ALL in ehm_capi.hip
K2Q_NPS = (8, 16, 24, 32)
# persistent frontier kernel at two solver widths (ehm_kp.hip): (decide NP, expand NP, slots);
# must match EHM_KP_ALL in ehm_capi.hip
# (widths = FACTORISED columns: ehm_ipm2.h eliminates the epigraph columns of the z-block)
KP_INSTANCES = tuple((d, e, sl) for (d, e) in ((12, 8), (16, 8), (16, 12), (20, 12), (20, 16),
                                                (24, 16), (24, 20), (28, 20), (28, 24), (32, 24),
                                                (32, 28)) for sl in (2, 3)) + \
    ((16, 12, 4), (28, 20, 4), (32, 24, 4))
# the same with the midpoint solve first (the default flow; EHM_KPM_ALL in ehm_capi.hip)
KPM_INSTANCES = KP_INSTANCES
# wide kernels (ehm_k3.hip): row slots per thread, rows <= 256 * slots; must match the
# ehm_k3_api_* getters in ehm_capi.hip
K3_RS = (2, 4)
# the single-width persistent kernel of ehm_k2.hip (quadratic handles, unlisted width pairs) runs
# the midpoint-first flow, like the ehm_kpm objects
MIDFIRST = '-DEHM_PERSIST_MIDFIRST=1'
# -fvisibility=hidden: the library exports what include/*.h declare (a visibility pragma there)
# and nothing else -- not the dispatch tables the objects hand each other (ehm_k2_api_* ...)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result',
         '-fvisibility=hidden', '-fvisibility-inlines-hidden']
# experiments: extra -D flags and an alternative output name, e.g.
#   EHM_BUILD_FLAGS="-DEHM2_UNROLL=2" EHM_BUILD_TAG=u2 python -m explicit_hybrid_mpc_amd.build
FLAGS += os.environ.get('EHM_BUILD_FLAGS', '').split()
_TAG = os.environ.get('EHM_BUILD_TAG', '')
if _TAG:
    OBJ_DIR = os.path.join(LIB_DIR, 'obj_' + _TAG)
    LIB = os.path.join(LIB_DIR, 'libehmpc_%s.so' % _TAG)


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: libehmpc.so cannot be built')


def _cxx():
    for cand in (os.environ.get('CXX'), shutil.which('g++'), shutil.which('c++')):
        if cand and os.path.exists(cand):
            return cand
    return _hipcc()


_INC_CACHE = {}


def _includes(path):
    """The quoted #include files of a source, transitively (paths normalised)."""
    path = os.path.normpath(path)
    if path in _INC_CACHE:
        return _INC_CACHE[path]
    _INC_CACHE[path] = found = set()
    try:
        text = open(path).read()
    except OSError:
        return found
    import re
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        dep = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if os.path.exists(dep):
            found.add(dep)
            found |= _includes(dep)
    return found


def _dep_mtime(src=None):
    """Newest dependency of an object: the headers its source includes (transitively; derived
    from the #include lines, so none can be forgotten) and this file.  src=None: every header
    (the staleness test of the library)."""
    if src is None:
        deps = [os.path.normpath(os.path.join(SRC_DIR, h)) for h in HEADERS]
    else:
        deps = list(_includes(src))
    deps += [__file__]
    return max(os.path.getmtime(d) for d in deps)


SEARCH_SRC = os.path.join(SRC_DIR, 'ehm_search.cpp')      # host C++ only (include/ehm_search.h)
SEARCH_OBJ = os.path.join(OBJ_DIR, 'ehm_search.o')
# the native partition driver of configs[4] (include/ehm_frontier.h): host C++ on the public
# entry points of the library
FRONTIER_SRC = os.path.join(SRC_DIR, 'ehm_frontier.cpp')
FRONTIER_OBJ = os.path.join(OBJ_DIR, 'ehm_frontier.o')
SEARCH_DEPS = [SEARCH_SRC, os.path.normpath(os.path.join(HERE, '..', 'include', 'ehm_search.h')),
               os.path.normpath(os.path.join(HERE, '..', 'include', 'ehmpc.h'))]
FRONTIER_DEPS = [FRONTIER_SRC,
                 os.path.normpath(os.path.join(HERE, '..', 'include', 'ehm_frontier.h'))] + \
    SEARCH_DEPS[1:]
HOST_OBJECTS = ((SEARCH_OBJ, SEARCH_SRC, SEARCH_DEPS), (FRONTIER_OBJ, FRONTIER_SRC, FRONTIER_DEPS))


def _objects():
    """(object path, source path, extra flags)"""
    objs = [(os.path.join(OBJ_DIR, 'ehm_capi.o'), os.path.join(SRC_DIR, 'ehm_capi.hip'), []),
            (os.path.join(OBJ_DIR, 'ehm_explicit.o'), os.path.join(SRC_DIR, 'ehm_explicit.hip'),
             [])]
    for np_ in K2_NPS:
        for sl in K2_SLOTS:
            objs.append((os.path.join(OBJ_DIR, 'ehm_k2_%d_%d.o' % (np_, sl)),
                         os.path.join(SRC_DIR, 'ehm_k2.hip'),
                         ['-DEHM_NP=%d' % np_, '-DEHM_SLOTS=%d' % sl, MIDFIRST]))
    for np_ in K2Q_NPS:
        for sl in K2_SLOTS:
            objs.append((os.path.join(OBJ_DIR, 'ehm_k2q_%d_%d.o' % (np_, sl)),
                         os.path.join(SRC_DIR, 'ehm_k2.hip'),
                         ['-DEHM_NP=%d' % np_, '-DEHM_SLOTS=%d' % sl, '-DEHM2_QUAD=1', MIDFIRST]))
    for npd, npe, sl in KP_INSTANCES:
        objs.append((os.path.join(OBJ_DIR, 'ehm_kp_%d_%d_%d.o' % (npd, npe, sl)),
                     os.path.join(SRC_DIR, 'ehm_kp.hip'),
                     ['-DEHM_NPD=%d' % npd, '-DEHM_NPE=%d' % npe, '-DEHM_SLOTS=%d' % sl]))
    for npd, npe, sl in KPM_INSTANCES:
        objs.append((os.path.join(OBJ_DIR, 'ehm_kpm_%d_%d_%d.o' % (npd, npe, sl)),
                     os.path.join(SRC_DIR, 'ehm_kp.hip'),
                     ['-DEHM_NPD=%d' % npd, '-DEHM_NPE=%d' % npe, '-DEHM_SLOTS=%d' % sl,
                      '-DEHM_PERSIST_MIDFIRST=1']))
    for rs in K3_RS:
        objs.append((os.path.join(OBJ_DIR, 'ehm_k3_%d.o' % rs),
                     os.path.join(SRC_DIR, 'ehm_k3.hip'),
                     # the 64-step elimination is unrolled completely (rows live in registers)
                     ['-DEHM3_RS=%d' % rs, '-mllvm', '-pragma-unroll-threshold=200000']))
    # the LDS-resident wide family (ehm_ipm4.h): one object, both tile counts inside
    objs.append((os.path.join(OBJ_DIR, 'ehm_k4.o'), os.path.join(SRC_DIR, 'ehm_k4.hip'), []))
    return objs


def _regenerate_kp():
    """ehm_kp.hip is derived from k2_persist in ehm_k2.hip (tools/gen_kp.py): keep it in step."""
    k2 = os.path.join(SRC_DIR, 'ehm_k2.hip')
    kp = os.path.join(SRC_DIR, 'ehm_kp.hip')
    gen = os.path.normpath(os.path.join(HERE, '..', 'tools', 'gen_kp.py'))
    if os.path.exists(gen) and (not os.path.exists(kp) or
                                os.path.getmtime(kp) < max(os.path.getmtime(k2),
                                                           os.path.getmtime(gen))):
        subprocess.check_call([os.environ.get('PYTHON', 'python'), gen])


def _stale(obj, src, dep_t):
    return (not os.path.exists(obj) or
            os.path.getmtime(obj) < max(dep_t, os.path.getmtime(src)))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if not os.path.exists(os.path.join(SRC_DIR, 'ehm_kp.hip')):
        return True                 # generated (tools/gen_kp.py), not in the repository
    srcs = [os.path.join(SRC_DIR, f)
            for f in ('ehm_capi.hip', 'ehm_k2.hip', 'ehm_k3.hip', 'ehm_k4.hip', 'ehm_kp.hip',
                      'ehm_explicit.hip')]
    return max([_dep_mtime()] + [os.path.getmtime(s)
                                 for s in srcs + SEARCH_DEPS + FRONTIER_DEPS]) > t


def build(force=False, verbose=False, jobs=None):
    """Compile the library if it is missing or older than its sources."""
    if not force and not is_stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    _regenerate_kp()
    hipcc = _hipcc()
    todo = [(o, s, f) for (o, s, f) in _objects() if force or _stale(o, s, _dep_mtime(s))]

    def compile_one(item):
        obj, src, extra = item
        cmd = [hipcc] + FLAGS + extra + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)

    jobs = jobs or max(1, (os.cpu_count() or 2))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(compile_one, todo))
    # the searches' host bookkeeping and the native driver on top of it: plain C++, no device code
    for obj, src, deps in HOST_OBJECTS:
        if force or not os.path.exists(obj) or \
                os.path.getmtime(obj) < max(os.path.getmtime(d) for d in deps):
            cmd = [_cxx(), '-O2', '-std=c++17', '-fPIC', '-Wall', '-fvisibility=hidden', '-fvisibility-inlines-hidden',
                   '-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + [o for (o, _, _) in _objects()]
    cmd += [obj for obj, _, _ in HOST_OBJECTS]
    cmd += ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
