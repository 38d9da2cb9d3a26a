"""
Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol the
header declares, the canonical LP compiler, the tree types (against the reference's
semantics) and the reference-format pickles.
"""

import os
import pickle
import re
import subprocess
import sys

import numpy as np
import pytest

from explicit_hybrid_mpc_amd import _capi, examples, tree, tree_io
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'ehmpc.h')).read()
    declared = set(re.findall(r'\b(ehm_[a-z_]+)\s*\(', header))
    declared -= {'ehm_problem_desc', 'ehm_run_opts'}
    assert declared == set(_capi.EXPORTED), declared ^ set(_capi.EXPORTED)
    lib = _capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.ehm_version()
    out = subprocess.run(['nm', '-D', '--defined-only', _capi.library_path()],
                         capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(r'\bT %s\b' % name, out), name
    # the host bookkeeping of the prefix searches (include/ehm_search.h), same library
    header = open(os.path.join(ROOT, 'include', 'ehm_search.h')).read()
    declared = set(re.findall(r'\b(ehm_search_[a-z_]+)\s*\(', header))
    assert declared == set(_capi.EXPORTED_SEARCH), declared ^ set(_capi.EXPORTED_SEARCH)
    for name in declared:
        assert hasattr(lib, name), name
        assert re.search(r'\bT %s\b' % name, out), name
    # ... and NOTHING else: the library is built with -fvisibility=hidden, its exported functions
    # are what the three public headers declare plus the profile hooks of experimental builds
    every = set()
    for h in ('ehmpc.h', 'ehm_search.h', 'ehm_frontier.h'):
        text = open(os.path.join(ROOT, 'include', h)).read()
        every |= set(re.findall(r'\b(ehm_[a-z0-9_]+)\s*\(', text))
    every -= {'ehm_problem_desc', 'ehm_run_opts'}
    exported = set(re.findall(r'\bT (\S+)', out))
    hooks = {'ehm_k3_profile_2', 'ehm_k3_profile_4', 'ehm_k4_profile'}
    assert exported - hooks == every, sorted((exported - hooks) ^ every)


def test_no_cpu_fallback_without_a_gpu():
    """On a box without a GPU the product path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from explicit_hybrid_mpc_amd import engine
    mpc = helpers.make_instance('di')
    with pytest.raises(_capi.EhmError) as err:
        engine.GpuProblem(mpc.compile(), 0.1, 0.1)
    assert err.value.code == _capi.EHM_E_NO_DEVICE
    with pytest.raises(_capi.EhmError):
        engine.split_batch(np.zeros((1, 3, 2)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'explicit_hybrid_mpc_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), f


def test_canonical_sizes_match_survey():
    can = examples.linear_mpc(0).compile()
    assert (can.n, can.m, can.p, can.n_u, can.n_delta) == (20, 160, 4, 2, 1)
    can3 = examples.pwa_mpc(0).compile()
    assert can3.n_delta == 32 and can3.deltas.shape == (32, 10)
    assert (can3.deltas.reshape(32, 5, 2).sum(axis=2) == 1).all()
    assert can3.delta_index(can3.deltas[7]) == 7
    with pytest.raises(ValueError):
        can3.delta_index(np.ones(10))


def test_theta_box_vertices_are_feasible():
    from oracle.oracle_cpu import OracleCPU
    for mpc in (examples.double_integrator(3), examples.linear_mpc(2), examples.pwa_mpc(1)):
        orc = OracleCPU(mpc, 1., 1.)
        for v in examples.box_vertices(examples.theta_box(mpc)):
            assert orc.P_theta(v, check_feasibility=True)


def test_tree_semantics_match_reference(golden_geometry):
    """Same observable behaviour as lib/tree.py (flags recorded from the real classes)."""
    nd = tree.NodeData(np.zeros((3, 2)))
    t = tree.Tree(nd)
    flags = [hasattr(nd, 'commutation'), hasattr(nd, 'vertex_costs'),
             hasattr(nd, 'vertex_inputs'), nd.is_epsilon_suboptimal, t.top, t.is_leaf()]
    t.grow(tree.NodeData(np.ones((3, 2))), None)
    flags += [t.is_leaf(), t.left.top, t.right.data is None, t.left.is_leaf()]
    assert [int(f) for f in flags] == golden_geometry['tree_flags'].tolist()
    other = tree.Tree(None)
    t.copy(other)
    assert other.left is t.left and other.data is t.data


def test_reference_format_pickle_roundtrip(tmp_path):
    root = tree.Tree(tree.NodeData(np.zeros((3, 2))))
    cursor = root
    for k in range(3000):            # deeper than the default recursion limit allows
        cursor.grow(tree.NodeData(np.full((3, 2), float(k)), commutation=np.ones(3),
                                  vertex_costs=np.arange(3.), vertex_inputs=np.ones((3, 1))),
                    None)
        cursor = cursor.right
    path = str(tmp_path / 'tree.pkl')
    tree_io.dump_reference(root, path)
    raw = open(path, 'rb').read()
    assert b'explicit_hybrid_mpc_amd' not in raw and b'tree' in raw
    # a consumer that only knows a module called `tree` (like the reference) can load it
    code = ("import sys, types, pickle; sys.setrecursionlimit(100000);"
            "m = types.ModuleType('tree');"
            "exec('class Tree: pass\\nclass NodeData: pass', m.__dict__);"
            "sys.modules['tree'] = m;"
            "t = pickle.load(open(%r, 'rb'));"
            "n = 0\n"
            "while hasattr(t, 'left'): t = t.right; n += 1\n"
            "print(n, type(t).__module__)") % path
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert out.stdout.split() == ['3000', 'tree'], out.stderr
    back = tree_io.load_reference(path)
    assert isinstance(back, tree.Tree) and tree_io.tree_depth(back) == 3000
    assert sys.getrecursionlimit() < 100000
    assert tree.Tree.__module__ == 'explicit_hybrid_mpc_amd.tree'


REFERENCE_LIB = '/root/reference/lib'


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE_LIB, 'tree.py')),
                    reason='the reference tree is only present in the build container')
def test_reference_tree_module_loads_the_pickle(tmp_path):
    """f1 with the REAL consumer: the unmodified /root/reference/lib/tree.py (imports cleanly,
    SURVEY 8c) unpickles what ``tree_io.dump_reference`` wrote -- a deep Delaunay-like spine and a
    grown cell -- into ITS classes, and its own methods (``is_leaf``, attribute absence of
    lib/tree.py:34-39, 86-95) read the result.  A tree pickled by the reference's classes comes
    back through ``tree_io.load_reference``."""
    root = tree.Tree(tree.NodeData(np.zeros((3, 2))))
    cursor = root
    for k in range(1500):
        cursor.grow(tree.NodeData(np.full((3, 2), float(k)), commutation=np.ones(3),
                                  vertex_costs=np.arange(3.), vertex_inputs=np.ones((3, 1))),
                    None if k < 1499 else tree.NodeData(np.full((3, 2), -1.)))
        cursor = cursor.right
    root.left.grow(tree.NodeData(np.eye(3, 2)), tree.NodeData(np.eye(3, 2) * 2.))
    root.left.left.data.is_epsilon_suboptimal = True
    path = str(tmp_path / 'tree.pkl')
    back_path = str(tmp_path / 'from_reference.pkl')
    tree_io.dump_reference(root, path)
    code = ("import sys, pickle; sys.dont_write_bytecode = True; sys.setrecursionlimit(100000);"
            "sys.path.insert(0, %r); import tree;"
            "assert tree.__file__.startswith(%r), tree.__file__;"
            "t = pickle.load(open(%r, 'rb'));"
            "assert type(t) is tree.Tree and type(t.data) is tree.NodeData;"
            "assert not t.is_leaf() and t.left.left.is_leaf();"
            "assert t.left.left.data.is_epsilon_suboptimal and not t.left.right.data.is_epsilon_suboptimal;"
            "assert hasattr(t.left.data, 'commutation') and not hasattr(t.data, 'commutation');"
            "n = 0; c = t\n"
            "while not c.right.is_leaf(): c = c.right; n += 1\n"
            "assert c.data is None and c.right.data.vertices[0, 0] == -1.\n"
            "r = tree.Tree(tree.NodeData(vertices=t.left.data.vertices, commutation=t.left.data.commutation))\n"
            "r.grow(tree.NodeData(vertices=t.left.left.data.vertices), tree.NodeData(vertices=t.left.right.data.vertices))\n"
            "pickle.dump(r, open(%r, 'wb'))\n"
            "print(n)") % (REFERENCE_LIB, REFERENCE_LIB, path, back_path)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert out.stdout.split() == ['1499'], out.stderr
    back = tree_io.load_reference(back_path)
    assert isinstance(back, tree.Tree) and isinstance(back.data, tree.NodeData)
    assert np.array_equal(back.data.commutation, np.ones(3)) and not back.is_leaf()
    assert np.array_equal(back.left.data.vertices, np.eye(3, 2))
    assert not hasattr(back.left.data, 'commutation')


def test_status_files_have_the_reference_format(tmp_path):
    """status.txt / statistics.pkl of lib/scheduler.py:154-362, fed from progress counters."""
    from explicit_hybrid_mpc_amd import status

    class Clock:
        t = 100.

        def __call__(self):
            return self.t

    clock = Clock()
    # the rate estimator: exponentially weighted mean -- the numbers of the reference's
    # recursive estimator (lib/scheduler.py:113-152), restated here step by step
    mean = status.ForgettingMean(call_period=10., time_constant=180.)
    assert mean.value is None
    est, sigma = None, 1.
    for m in (0.02, 0.03, 0.01, 0.05):
        mean.update(m)
        if est is None:
            est = m
        else:
            sigma = 1. + np.exp(-10. / 180.) * sigma
            est = m / sigma + (1. - 1. / sigma) * est
        assert abs(mean.value - est) < 1e-15
    eta = status.EtaEstimate(10., 180.)
    assert eta.eta(0.3) is None
    eta.update(0.01)
    assert abs(eta.eta(0.3) - 70.) < 1e-12

    st, pk = str(tmp_path / 'status.txt'), str(tmp_path / 'statistics.pkl')
    pub = status.MainStatusPublisher(16., st, pk, clock=clock, eta_window_duration=1.)
    w = status.WorkerStatus('ecc', clock=clock)
    w.set_total_volume(16.)
    w.update(active=True)
    for k, (vol, nodes) in enumerate([(2., 100), (6., 400), (16., 900)]):
        clock.t += 2.
        # n_splits is what the reference's publisher counts (+1 per split, lib/worker.py:274,327)
        w.absorb(dict(volume_closed=vol, n_nodes=2 * nodes + 22, n_splits=nodes,
                      n_closed=nodes // 2, frontier=50 - 10 * k,
                      depth=3 + k, sweeps=k + 1, lp_solves=10 * nodes, ipm_iters=90 * nodes))
        overall = pub.update([w.data, None], num_tasks_in_queue=50 - 10 * k)
        assert abs(overall['volume_filled_frac'] - vol / 16.) < 1e-15
    clock.t += 1.
    w.update(active=False)
    pub.update([w.data, None], force=True)
    assert w.data['time_active_total'] == 7. and w.data['time_ecc'] == 7.
    text = open(st).read()
    for line in ('# overall', 'number of processes active: 0', 'volume filled (total [%]): 1.0000e+02',
                 'simplex_count: 900', 'processes: 0 x ecc, 0 x lcss', '# proc 0', 'status: idle',
                 'volume filled (current [%]): 1.0000e+02', 'simplex count (total [-]): 900'):
        assert line in text, line
    assert re.search(r'^ETA \[s\]: (None|\d+)$', text, re.M)
    stats = status.load_statistics(pk)          # the reader of lib/post_process.py:57-70
    assert stats['volume_filled_frac'] == [2. / 16., 6. / 16., 1., 1.]
    assert stats['simplex_count_total'] == [100, 400, 900, 900]
    assert stats['num_proc_active'] == [1, 1, 1, 0]
    assert stats['time_elapsed'] == sorted(stats['time_elapsed'])
    # volume filled 2/16 -> 6/16 in 2 s, then 6/16 -> 1 in 2 s; ETA after the second window
    assert stats['eta'][0] is None and stats['eta'][1] is not None
    with open(pk, 'rb') as f:
        rec = pickle.load(f)
    assert set(rec) == {'overall', 'process'} and rec['process'][1] is None
    assert set(rec['process'][0]) >= {'status', 'current_branch', 'current_location', 'algorithm',
                                      'volume_filled_total', 'volume_filled_current',
                                      'simplex_count_total', 'simplex_count_current',
                                      'time_active_total', 'time_active_current', 'time_idle',
                                      'time_ecc', 'time_lcss'}


def test_deep_pickles_do_not_overflow_the_c_stack(tmp_path):
    """
    tree_io (un)pickles nested trees in a thread whose stack is sized for the depth: the
    8-dimensional Delaunay spine is 34 573 levels deep, far beyond what a
    raised recursion limit alone survives; a tree deeper than the hint raises RecursionError
    instead of crashing the interpreter.
    """
    from explicit_hybrid_mpc_amd import tree_io
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    depth = 35000
    root = Tree(None)
    node = root
    for _ in range(depth):
        node.grow(NodeData(vertices=np.zeros((3, 2))), None)
        node = node.right
    path = str(tmp_path / 'spine.pkl')
    tree_io.dump_reference(root, path, depth_hint=depth + 1)
    back = tree_io.load_reference(path)
    n, node = 0, back
    while not node.is_leaf():
        node, n = node.right, n + 1
    assert n == depth
    with pytest.raises(RecursionError):      # a hint that is too small fails cleanly
        tree_io.dump_reference(root, str(tmp_path / 'x.pkl'), depth_hint=50)
    with pytest.raises(ValueError):
        tree_io.load_reference(path, depth_hint=10 ** 7)


def test_bench_helpers_quote_only_profiles_of_this_code(tmp_path, monkeypatch):
    """bench.py: the committed PMC summaries carry the hash of the kernel sources they were taken
    on and are quoted for those sources only; the flop formulas and the core count are sane."""
    import json
    import bench
    sha = bench.kernel_source_hash()
    assert len(sha) == 16 and sha == bench.kernel_source_hash()
    # the committed summaries of the four workloads name the kernel sources they belong to
    for summary, kernel in (('pmc_summary_bench.json', 'kp_persist'),
                            ('pmc_summary_wide.json', bench.wide_kernel('persist')),
                            ('pmc_summary_config3.json', 'k2_simplex_batch'),
                            ('pmc_summary_quad.json', 'k2_persist')):
        traffic, source = bench.pmc_traffic(kernel, summary)
        # quoted (taken on these sources) or refused as stale -- never silently quoted for
        # other code, never unreadable
        assert (traffic is not None and traffic > 0 and source.endswith(summary)) or \
            (traffic is None and source.startswith('stale')), (summary, source)
    # a summary taken on other sources is refused, with the reason
    root = tmp_path / 'profiles' / 'r2'
    root.mkdir(parents=True)
    doc = json.load(open(os.path.join(bench.ROOT, 'profiles', 'r2', 'pmc_summary_bench.json')))
    doc['kernel_source_sha'] = '0' * 16
    (root / 'pmc_summary_bench.json').write_text(json.dumps(doc))
    csrc = tmp_path / 'explicit_hybrid_mpc_amd' / 'csrc'
    csrc.mkdir(parents=True)
    (csrc / 'a.hip').write_text('// other code')
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    traffic, why = bench.pmc_traffic('kp_persist', 'pmc_summary_bench.json')
    assert traffic is None and why.startswith('stale')
    assert bench.pmc_traffic('no_such_kernel', 'pmc_summary_bench.json')[0] is None
    assert bench.pmc_traffic('kp_persist', 'no_such_summary.json')[0] is None
    # SURVEY 8(d) formula at config 2's suboptimality-test LP, and what the solver executes
    assert bench.flops_per_iteration(25, 167) == 2 * 167 * 625 + 25 ** 3 / 3. + 8 * 167 * 25 + 2500
    assert bench.np_capacity(25) == 28 and bench.np_capacity(20) == 20
    assert 0 < bench.flops_executed_per_iteration(25, 24, 167, 28) < bench.flops_per_iteration(25, 167)
    assert bench.node_bytes(4, 2, 5) == 301
    assert 1 <= bench.usable_cores() <= (os.cpu_count() or 1)


def test_prefix_blocks_built_incrementally_are_the_reference_blocks():
    """PWAMPC.condense_prefix (template + cached predictions, only the decided steps' rows
    computed) is bit-identical to the plain form -- also when the mode regions have different
    numbers of rows or are absent."""
    from explicit_hybrid_mpc_amd import mpc_library
    base = helpers.make_instance('pwa_small', 0)
    H1, h1 = base.regions[1]
    uneven = mpc_library.PWAMPC(base.A, base.B, base.w,
                                [base.regions[0], (np.vstack([H1, H1[:1]]), np.append(h1, h1[0] + 1.))],
                                base.Gx, base.gx, base.Gu, base.gu, base.Q, base.R, base.N,
                                name='uneven')
    free = mpc_library.PWAMPC(base.A, base.B, base.w, [base.regions[0], None],
                              base.Gx, base.gx, base.Gu, base.gu, base.Q, base.R, base.N,
                              name='one_free_mode')
    rng = np.random.default_rng(0)
    for mpc in (base, uneven, free, examples.pwa4_mpc(N=8)):
        prefixes = [()] + [tuple(int(i) for i in rng.integers(0, mpc.delta_size,
                                                              rng.integers(1, mpc.N + 1)))
                           for _ in range(60)]
        for q in prefixes + prefixes[:10]:              # the second pass hits the cache
            new, ref = mpc.condense_prefix(q), mpc._condense_prefix_reference(q)
            assert all(np.array_equal(a, b) for a, b in zip(new, ref)), (mpc.name, q)
        # a full prefix is the block of the compiled table
        full = tuple(int(i) for i in rng.integers(0, mpc.delta_size, mpc.N))
        G, w, S = mpc.condense_prefix(full)
        Gc, wc, Sc = mpc._condense(full, G.shape[0])
        assert np.array_equal(G, Gc) and np.array_equal(w, wc) and np.array_equal(S, Sc)
    # a restricted copy does not share the caches
    sub = base.restrict([(0, 0, 0), (1, 0, 1)])
    assert all(np.array_equal(a, b) for a, b in zip(sub.condense_prefix((1, 0)),
                                                    base._condense_prefix_reference((1, 0))))


def test_midpoint_table_protocol_under_threads(tmp_path):
    """csrc/ehm_midtable.h compiled for the host (device builtins mapped onto GCC atomics and
    fences), eight threads as wavefronts: every request gets the value of its key, every key is
    solved exactly once, nobody hangs (tests/native/midtable_stress.cpp)."""
    exe = str(tmp_path / 'midtable_stress')
    src = os.path.join(ROOT, 'tests', 'native', 'midtable_stress.cpp')
    subprocess.run(['g++', '-O2', '-std=c++17', '-pthread', '-include', 'algorithm', src, '-o', exe],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'OK' in out.stdout, out.stdout + out.stderr


def test_ctypes_mirrors_are_the_librarys_structs():
    """ehm_abi_sizes (include/ehmpc.h): the ctypes classes of _capi.py have the sizes of the C
    structs they mirror -- _capi.load() refuses a library it would let overrun its buffers."""
    import ctypes
    lib = _capi.load()
    mirrors = (_capi.ProblemDesc, _capi.RunOpts, _capi.NodeInit, _capi.Progress, _capi.TreeInfo,
               _capi.Counters)
    sizes = (ctypes.c_int64 * 8)()
    n = lib.ehm_abi_sizes(ctypes.addressof(sizes), 8)
    assert n == len(mirrors)
    assert [int(sizes[k]) for k in range(n)] == [ctypes.sizeof(m) for m in mirrors]
    assert lib.ehm_abi_sizes(None, 0) == n


def test_config5_helpers_of_the_benchmark():
    """bench.kuhn_cell: the Kuhn simplices of the rotated coordinate orders are cells of the box
    (volume |box| / p!, vertices at its corners, the main diagonal in all of them);
    sequences.short_horizon / make_table: the two-table form exactly where the full model needs
    the wide kernels and a shorter horizon fits the shared-block ones."""
    import math
    import bench
    from explicit_hybrid_mpc_amd import examples, sequences
    from oracle import geometry
    half = np.array([0.2, 0.3, 0.1, 0.25])
    p = half.size
    seen = set()
    for k in range(p):
        R = bench.kuhn_cell(half, k)
        assert R.shape == (p + 1, p)
        assert np.all(np.abs(np.abs(R) - half) == 0)                     # corners of the box
        assert np.array_equal(R[0], -half) and np.array_equal(R[-1], half)
        assert abs(geometry.simplex_volume(R) - np.prod(2 * half) / math.factorial(p)) < 1e-15
        seen.add(R.tobytes())
    assert len(seen) == p
    assert np.array_equal(bench.kuhn_cell(half, 0),
                          np.array([-half + 2 * half * (np.arange(p) < k) for k in range(p + 1)]))
    big = examples.pwa4_mpc(N=8)
    assert sequences.short_horizon(big) == 4                  # 29 x 195 fits, 34 columns do not
    assert sequences.short_horizon(examples.pwa4_mpc(N=4)) == 0          # fits as it is
    assert sequences.short_horizon(helpers.make_instance('pwa_small', 0)) == 0
    # the blocks of the two horizons are the same problem only where u = 0 is admissible (the
    # undecided steps of the full model then cost nothing): no split otherwise
    import copy
    shifted = copy.copy(big)
    shifted.gu = np.asarray(big.gu, dtype=np.float64).copy()
    shifted.gu[0] = -0.1
    assert sequences.short_horizon(shifted) == 0
    four = big.with_horizon(4)
    assert four.N == 4 and four.delta_size == big.delta_size
    G4, _, _ = four.condense_prefix((2, 1))
    G8, _, _ = big.condense_prefix((2, 1))
    assert G4.shape == (184, 20) and G8.shape == (368, 40)      # 29 x 195 against 49 x 379 slack LPs
    # (that the two blocks are the same PROBLEM: tests/test_host_bnb.py, the split-table test)


def test_a_failing_secondary_workload_does_not_cost_the_headline(monkeypatch):
    """bench.secondary_line: an exception (or SystemExit) of a secondary workload becomes an
    'error' entry of the list; the fields of a good entry are the documented subset."""
    import argparse
    import bench
    args = bench.parse_args([])

    def boom(a, ctx):
        raise RuntimeError('no device here')
    monkeypatch.setattr(bench, 'measure', boom)
    line = bench.secondary_line(args, {}, 'config3', 2, 1)
    assert line['workload'] == 'config3' and 'RuntimeError: no device here' in line['error']

    def fine(a, ctx):
        assert a.workload == 'config4' and (a.steps, a.warmup) == (2, 1)
        assert a.abs_frac is None and a.eps_r is None and a.cpu_seconds == args.secondary_cpu_seconds
        return dict(value=1., unit='LP solves/s', ms_per_step=2., regions_per_s=3.,
                    oracle_calls_answered_per_s=4., steps=2, warmup=1, roofline={'frac': 0.1},
                    cpu_baseline={'value': 5.}, config={'workload': 'w', 'regions_per_step': 7,
                                                        'not_copied': 1})
    monkeypatch.setattr(bench, 'measure', fine)
    line = bench.secondary_line(args, {}, 'config4', 2, 1)
    assert line['name'] == 'config4' and line['workload'] == 'w' and line['ms_per_step'] == 2.
    assert line['config'] == {'workload': 'w', 'regions_per_step': 7}
    assert line['roofline'] == {'frac': 0.1} and line['cpu_baseline'] == {'value': 5.}
    assert isinstance(args, argparse.Namespace) and args.workload == 'config2'   # untouched
    # the time-boxed configs[4] entry: the config5 driver with a region target, the Delaunay roots
    # in order, a soft time limit and the depth limit -- and skippable
    seen = {}

    def c5(a, ctx):
        seen.update(workload=a.workload, regions=a.regions, cells=a.cells, seconds=a.seconds,
                    max_depth=a.max_depth)
        return dict(value=1., unit='LP solves/s', ms_per_step=2., regions_per_s=3., steps=1,
                    warmup=0, roofline={}, cpu_baseline=None, config={'workload': 'c5'})
    monkeypatch.setattr(bench, 'measure_config5', c5)
    line = bench.secondary_line(args, {}, 'config5_scale', 1, 0)
    assert seen == dict(workload='config5', regions=10 ** 6, cells=400,
                        seconds=args.scale_seconds, max_depth=26)
    assert line['name'] == 'config5_scale' and line['limits']['soft_seconds'] == args.scale_seconds
    args.scale_seconds = 0.
    assert 'skipped' in bench.secondary_line(args, {}, 'config5_scale', 1, 0)


def test_cpu_baseline_leg_of_the_bench_runs():
    """bench.py's ``cpu_baseline`` leg (the oracle port on the host cores) injects nodes into
    PartitionCPU without ``run``: it must work on a partition whose path codes were never seeded
    (round 4: a recursion there would have failed the driver's bench run)."""
    import bench
    out = bench.cpu_baseline('config2', 0, 0.0216306, 0.01, 1.0)
    assert out['value'] > 0 and out['kind'] == 'port' and out['cores'] >= 1


def test_lexicographic_stage_problems_and_the_oracles_rule():
    """explicit_hybrid_mpc_amd/lexicographic.py builds stage 1+j as a canonical LP with 1+j more
    rows and parameters; the oracle's statement of the rule (HiGHS, uncondensed model) returns an
    OPTIMAL first input that is lexicographically no larger than the solver's own vertex."""
    from explicit_hybrid_mpc_amd import examples, lexicographic
    from oracle.oracle_cpu import OracleCPU
    mpc = examples.double_integrator(3)
    can = mpc.compile()
    for j in range(can.n_u):
        st = lexicographic.stage_problem(can, j)
        assert (st.m, st.p, st.n, st.n_delta) == (can.m + 1 + j, can.p + 1 + j, can.n, can.n_delta)
        assert np.array_equal(st.G[:, :can.m], can.G) and np.array_equal(st.G[0, can.m], can.c)
        assert st.c[j] == 1. and st.c.sum() == 1.
        assert np.array_equal(st.S[0, can.m:, can.p:], np.eye(1 + j))
    wide = examples.integrator_chain_mpc().compile()                # p = 6, n_u = 3
    with pytest.raises(ValueError, match='EHM_MAX_P'):
        lexicographic.LexicographicInputs(wide)
    with pytest.raises(ValueError, match='strictly convex'):
        lexicographic.LexicographicInputs(examples.double_integrator(3, cost='quadratic').compile())
    orc = OracleCPU(mpc, 0.1, 0.1)
    half = examples.theta_box(mpc)
    rng = np.random.default_rng(0)
    wider = 0
    for _ in range(40):
        theta = (rng.random(2) * 2 - 1) * half * 0.8
        ok, u, J = orc._point(theta, 0)
        if not ok:
            continue
        ul = orc.lexicographic_u0(theta, orc.deltas[0], tol=1e-7)
        assert ul[0] <= u[0] + 1e-7
        wider += ul[0] < u[0] - 1e-3
        # still optimal: fixing u_0 at the returned input costs no more than the tolerance
        m = orc.models[0]
        lp = m.lp_point(theta)
        row = np.zeros((1, m.nv))
        row[0, m.ou] = 1.
        from scipy.optimize import linprog
        res = linprog(lp['c'], A_ub=lp['A_ub'], b_ub=lp['b_ub'],
                      A_eq=np.vstack([lp['A_eq'], row]), b_eq=np.concatenate([lp['b_eq'], ul[:1]]),
                      bounds=(None, None), method='highs')
        assert res.status == 0 and res.fun <= J + 2e-7 * (1. + abs(J))
    assert wider >= 2           # the optimal face of the double integrator is wide at some
