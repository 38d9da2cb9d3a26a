"""
Whole-tree identity of the HEADLINE partition (bench.py default: configs[1], 22 Delaunay roots,
eps_r 1e-2, abs_frac 0.02 -- 1 610 186 nodes) or of bench.py --workload config4 (--workload
config4: configs[3], 652 roots, 325 080 nodes on the wide kernels) or of bench.py --workload config3
(--workload config3: configs[2], 32 commutations on the multi-commutation engine, nodes left open at
depth 18; see _check_block_hybrid for what differs) against the CPU oracle, node by node.

Every node of the exported device tree is decided AGAIN by the CPU restatement of
lib/worker.py:293-417 (oracle/partition_cpu.py on oracle/oracle_cpu.py: HiGHS on the uncondensed
model) from the record the device exported for it -- one visit per node, so a disagreement at one
node cannot hide the nodes below it:

    vertices of both children bit-equal; the verdict (closed / split) equal; the children's
    vertex costs within 1e-7; the root cells (action 'ecc') from their bare vertices.

A disagreement at a node whose |t*| is below EHM_ROUTE_TOL (1 + |V_0|) is a ROUTED one (two
correct solvers may part ways there, tests/test_gpu_bench_parity.py); any other is a failure.
The CPU side runs on every usable core (one process each, the nodes dealt in blocks).

    python tools/whole_tree_identity.py [--abs-frac 0.02] [--limit N] [--out FILE]

This is test infrastructure (it imports oracle/); the sampled form is the CI test.
"""

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

RTOL = 1e-7
ROUTE_TOL = 1e-6            # EHM_ROUTE_TOL (csrc/ehm_k2.h)
_G = {}


def usable_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


def _mpc_of(workload, seed):
    from explicit_hybrid_mpc_amd import examples
    if workload == 'config3':       # configs[2]: 32 commutations, the multi-commutation engine
        return examples.pwa_mpc(seed=seed)
    if workload == 'config4':       # configs[3]: the wide LPs (ehm_k4.hip / ehm_k3.hip)
        return examples.integrator_chain_mpc()
    return examples.linear_mpc(seed)


def _init(seed, eps_a, eps_r, arrays, workload='config2'):
    from oracle.oracle_cpu import OracleCPU
    mpc = _mpc_of(workload, seed)
    orc = OracleCPU(mpc, eps_a, eps_r)
    orc.memoize = True
    _G.update(orc=orc, **arrays)


def _check_block(block):
    """Nodes block[0] .. block[1]-1 re-decided; returns counters and the list of disagreements."""
    from oracle.partition_cpu import PartitionCPU
    orc = _G['orc']
    V, L, Rt, C, U, F, T, D, deltas, is_root = (_G[k] for k in (
        'vertices', 'left', 'right', 'vertex_costs', 'vertex_inputs', 'flags', 'tstar',
        'delta_idx', 'deltas', 'is_root'))
    out = dict(nodes=0, closed=0, splits=0, routed=0, unrouted=0, max_cost_diff=0.,
               max_input_diff=0., lp=0, bad=[])
    if len(orc._memo) > 400000:
        orc._memo.clear()
    n0 = orc.n_solves
    for k in range(block[0], block[1]):
        leaf = L[k] < 0
        closed = bool(F[k] & 1)
        cpu = PartitionCPU(orc, max_nodes=2 if is_root[k] else 1)
        if is_root[k]:
            # a root cell: 'ecc' from its bare vertices (lib/worker.py:241-291) gives the record,
            # the second visit decides it
            nodes = cpu.run([V[k].copy()], [''], 'ecc')
            ref = nodes['']
            if ref['commutation'] is None:
                out['bad'].append((int(k), 'root split by ecc on the CPU'))
                out['unrouted'] += 1
                continue
            d = float(np.max(np.abs(ref['vertex_costs'] - C[k])))
            out['max_cost_diff'] = max(out['max_cost_diff'], d)
            if not np.allclose(ref['vertex_costs'], C[k], rtol=RTOL, atol=RTOL):
                out['bad'].append((int(k), 'root vertex costs differ by %g' % d))
                out['unrouted'] += 1
                continue
        else:
            root = dict(vertices=V[k].copy(), commutation=deltas[D[k]].copy(),
                        vertex_costs=C[k].copy(), vertex_inputs=U[k].copy(),
                        is_epsilon_suboptimal=False, leaf=True)
            nodes = cpu.run([root], [''], 'lcss')
            ref = nodes['']
        out['nodes'] += 1
        same = (bool(leaf) == bool(ref['leaf'])) and (closed == bool(ref['is_epsilon_suboptimal']))
        if not same:
            tol = ROUTE_TOL * (1. + abs(C[k][0]))
            if abs(T[k]) < tol:
                out['routed'] += 1
            else:
                out['unrouted'] += 1
                out['bad'].append((int(k), 'verdict: device leaf=%s closed=%s, CPU leaf=%s '
                                   'closed=%s, t*=%g' % (bool(leaf), closed, ref['leaf'],
                                                         ref['is_epsilon_suboptimal'], T[k])))
            continue
        if leaf:
            out['closed'] += closed
            continue
        out['splits'] += 1
        for name, kid in (('0', int(L[k])), ('1', int(Rt[k]))):
            r = nodes[name]
            if not np.array_equal(V[kid], r['vertices']):
                out['unrouted'] += 1
                out['bad'].append((int(k), 'child %s: vertices differ' % name))
                continue
            d = float(np.max(np.abs(r['vertex_costs'] - C[kid])))
            du = float(np.max(np.abs(r['vertex_inputs'] - U[kid])))
            out['max_cost_diff'] = max(out['max_cost_diff'], d)
            out['max_input_diff'] = max(out['max_input_diff'], du)
            if not np.allclose(r['vertex_costs'], C[kid], rtol=RTOL, atol=RTOL):
                out['unrouted'] += 1
                out['bad'].append((int(k), 'child %s: vertex costs differ by %g' % (name, d)))
    out['lp'] = orc.n_solves - n0
    return out


def _check_block_hybrid(block):
    """
    The same for a multi-commutation tree (configs[2], csrc/ehm_hybrid.h).  A node's exported
    record is its FINAL one (lib/worker.py:396-401 swaps the commutation in place and visits the
    node again), so the CPU continues from that record and has to make the device's decision in
    ONE visit; a node 'ecc' split (it never held a commutation: flags bit 1 clear) is given to the
    CPU's 'ecc' from its bare vertices.  A child's costs are compared where the child still
    holds the commutation the split handed it (it may have swapped since; that swap is checked
    when the child's own turn comes).  Nodes the device left open at the depth limit are counted,
    not decided.
    """
    from oracle.partition_cpu import PartitionCPU
    orc = _G['orc']
    V, L, Rt, C, U, F, T, D, deltas, depth, max_depth = (_G[k] for k in (
        'vertices', 'left', 'right', 'vertex_costs', 'vertex_inputs', 'flags', 'tstar',
        'delta_idx', 'deltas', 'depth', 'max_depth'))
    out = dict(nodes=0, closed=0, splits=0, routed=0, unrouted=0, max_cost_diff=0.,
               max_input_diff=0., lp=0, bad=[], ecc_splits=0, open_at_depth_limit=0,
               children_compared=0, children_swapped_since=0)
    if len(orc._memo) > 400000:
        orc._memo.clear()
    n0 = orc.n_solves

    def bad(k, what):
        out['unrouted'] += 1
        out['bad'].append((int(k), what))

    for k in range(block[0], block[1]):
        leaf = L[k] < 0
        closed = bool(F[k] & 1)
        has_data = bool(F[k] & 2)
        if leaf and not closed and depth[k] >= max_depth:
            out['open_at_depth_limit'] += 1
            continue
        cpu = PartitionCPU(orc, max_nodes=1)
        if not has_data:
            # split by 'ecc' (lib/worker.py:269-277): no commutation is feasible at every vertex
            nodes = cpu.run([V[k].copy()], [''], 'ecc')
            out['nodes'] += 1
            if leaf or nodes['']['commutation'] is not None or nodes['']['leaf']:
                bad(k, "'ecc': device leaf=%s, CPU found commutation %s" % (
                    bool(leaf), nodes['']['commutation']))
                continue
            out['ecc_splits'] += 1
            for name, kid in (('0', int(L[k])), ('1', int(Rt[k]))):
                if not np.array_equal(V[kid], nodes[name]['vertices']):
                    bad(k, 'child %s: vertices differ' % name)
            continue
        root = dict(vertices=V[k].copy(), commutation=deltas[D[k]].copy(),
                    vertex_costs=C[k].copy(), vertex_inputs=U[k].copy(),
                    is_epsilon_suboptimal=False, leaf=True)
        nodes = cpu.run([root], [''], 'lcss')
        ref = nodes['']
        out['nodes'] += 1
        revisit = ref['leaf'] and not ref['is_epsilon_suboptimal']      # swapped in place again
        same = (not revisit) and (bool(leaf) == bool(ref['leaf'])) and \
            (closed == bool(ref['is_epsilon_suboptimal']))
        if not same:
            tol = ROUTE_TOL * (1. + abs(C[k][0]))
            if abs(T[k]) < tol:
                out['routed'] += 1
            else:
                bad(k, 'verdict: device leaf=%s closed=%s, CPU leaf=%s closed=%s revisit=%s, '
                       't*=%g' % (bool(leaf), closed, ref['leaf'], ref['is_epsilon_suboptimal'],
                                  revisit, T[k]))
            continue
        if leaf:
            out['closed'] += closed
            continue
        out['splits'] += 1
        for name, kid in (('0', int(L[k])), ('1', int(Rt[k]))):
            r = nodes[name]
            if not np.array_equal(V[kid], r['vertices']):
                bad(k, 'child %s: vertices differ' % name)
                continue
            if not (F[kid] & 2):
                bad(k, 'child %s carries no data on the device' % name)
                continue
            if not np.array_equal(deltas[D[kid]].astype(int), r['commutation'].astype(int)):
                out['children_swapped_since'] += 1
                continue
            out['children_compared'] += 1
            d = float(np.max(np.abs(r['vertex_costs'] - C[kid])))
            du = float(np.max(np.abs(r['vertex_inputs'] - U[kid])))
            out['max_cost_diff'] = max(out['max_cost_diff'], d)
            out['max_input_diff'] = max(out['max_input_diff'], du)
            if not np.allclose(r['vertex_costs'], C[kid], rtol=RTOL, atol=RTOL):
                bad(k, 'child %s: vertex costs differ by %g' % (name, d))
    out['lp'] = orc.n_solves - n0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--workload', choices=['config2', 'config3', 'config4'], default='config2',
                    help='config2 = the headline; config4 = bench.py --workload config4 (n_x=6, '
                         'N=10: 325 080 nodes on the wide kernels, abs_frac 0.4, eps_r 0.25); '
                         'config3 = bench.py --workload config3 (32 commutations, abs_frac 0.1, '
                         'eps_r 1e-2, nodes left open at depth 18)')
    ap.add_argument('--max-depth', type=int, default=None)
    ap.add_argument('--abs-frac', type=float, default=None)
    ap.add_argument('--eps-r', type=float, default=None)
    ap.add_argument('--limit', type=int, default=0, help='check only the first N nodes (0 = all)')
    ap.add_argument('--stride', type=int, default=1, help='check every s-th block of nodes')
    ap.add_argument('--cores', type=int, default=0)
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    from explicit_hybrid_mpc_amd import engine, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    hybrid = args.workload == 'config3'
    if args.abs_frac is None:
        args.abs_frac = {'config4': 0.4, 'config3': 0.1}.get(args.workload, 0.02)
    if args.eps_r is None:
        args.eps_r = 0.25 if args.workload == 'config4' else 1e-2
    if args.max_depth is None:
        args.max_depth = 18 if hybrid else 0          # bench.CONFIG3
    mpc = _mpc_of(args.workload, args.seed)
    gp = engine.GpuProblem(mpc.compile(), 1., 1.)
    Vb = examples.box_vertices(examples.theta_box(mpc))
    eps_a = float(np.max(gp.solve_pt(args.abs_frac * Vb)[0]))
    gp.set_eps(eps_a, args.eps_r)
    roots, _ = ehm_tools.delaunay_roots(Vb)
    t0 = time.perf_counter()
    flat = gp.partition(roots, action='ecc', max_nodes=1 << 22, max_depth=args.max_depth)
    t_dev = time.perf_counter() - t0
    gp.close()
    n = flat.n_nodes
    is_root = np.zeros(n, dtype=bool)
    is_root[:len(roots)] = True
    # the breadth-first export lists the roots first, in the order given
    assert all(np.array_equal(flat.vertices[k], roots[k]) for k in range(len(roots)))
    arrays = dict(vertices=flat.vertices, left=flat.left, right=flat.right,
                  vertex_costs=flat.vertex_costs, vertex_inputs=flat.vertex_inputs,
                  flags=flat.flags, tstar=flat.tstar, delta_idx=flat.delta_idx,
                  deltas=flat.deltas, is_root=is_root)
    if hybrid:
        depth = np.zeros(n, dtype=np.int32)
        for k in range(n):                  # parents precede children in the export
            if flat.left[k] >= 0:
                depth[flat.left[k]] = depth[flat.right[k]] = depth[k] + 1
        arrays.update(depth=depth, max_depth=args.max_depth)
    limit = min(n, args.limit) if args.limit else n
    size = 250 if hybrid else 2000      # a multi-commutation visit is ~80 LPs on the CPU
    blocks = [(a, min(a + size, limit)) for a in range(0, limit, size)][::max(1, args.stride)]
    cores = args.cores or usable_cores()
    print('device tree: %d nodes, %d regions (%.2f s incl. export); CPU oracle on %d cores, %d '
          'blocks of <= %d nodes' % (n, int(flat.info['n_closed']), t_dev, cores, len(blocks),
                                     size), flush=True)
    tot = dict(nodes=0, closed=0, splits=0, routed=0, unrouted=0, max_cost_diff=0.,
               max_input_diff=0., lp=0, bad=[])
    if hybrid:
        tot.update(ecc_splits=0, open_at_depth_limit=0, children_compared=0,
                   children_swapped_since=0)
    t0 = time.perf_counter()
    with mp.get_context('fork').Pool(cores, initializer=_init,
                                     initargs=(args.seed, eps_a, args.eps_r, arrays,
                                               args.workload)) as pool:
        for i, r in enumerate(pool.imap_unordered(_check_block_hybrid if hybrid else _check_block,
                                                   blocks)):
            for k, v in r.items():
                if k.startswith('max_'):
                    tot[k] = max(tot[k], v)
                else:
                    tot[k] += v
            if (i + 1) % 50 == 0 or i + 1 == len(blocks):
                print('  %d / %d blocks, %d nodes, %d routed, %d UN-ROUTED, %.0f s' % (
                    i + 1, len(blocks), tot['nodes'], tot['routed'], tot['unrouted'],
                    time.perf_counter() - t0), flush=True)
    wall = time.perf_counter() - t0
    near = np.abs(flat.tstar) < ROUTE_TOL * (1. + np.abs(flat.vertex_costs[:, 0]))
    if hybrid:          # nodes 'ecc' split and nodes left open never ran a slack LP (t* = 0)
        near &= ((flat.flags & 2) != 0) & ((flat.left >= 0) | ((flat.flags & 1) != 0))
    near = int(np.sum(near))
    rec = dict(workload='bench.py %s: %s seed %d, abs_frac %g, eps_r %g, %d Delaunay '
                        'roots' % ('headline' if args.workload == 'config2' else
                                   '--workload ' + args.workload, mpc.name, args.seed, args.abs_frac,
                                   args.eps_r, len(roots)),
               eps_a=eps_a, device_nodes=int(n), device_regions=int(flat.info['n_closed']),
               nodes_checked=tot['nodes'], closed_leaves_equal=tot['closed'],
               splits_equal=tot['splits'], routed_disagreements=tot['routed'],
               unrouted_disagreements=tot['unrouted'], near_threshold_nodes_in_the_tree=near,
               max_vertex_cost_difference=tot['max_cost_diff'],
               max_vertex_input_difference=tot['max_input_diff'],
               cpu_lp_solves=tot['lp'], cpu_cores=cores, cpu_seconds=wall,
               first_disagreements=tot['bad'][:20])
    if hybrid:
        rec.update(max_depth=args.max_depth, ecc_splits_equal=tot['ecc_splits'],
                   open_at_the_depth_limit_not_decided=tot['open_at_depth_limit'],
                   children_compared=tot['children_compared'],
                   children_that_swapped_their_commutation_since=tot['children_swapped_since'])
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as f:
            json.dump(rec, f, indent=1)
    return 1 if tot['unrouted'] else 0


if __name__ == '__main__':
    sys.exit(main())
