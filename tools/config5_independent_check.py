"""
configs[4] (n_x = 8, n_u = 3, 4 modes, N = 8: 65 536 mode sequences): nodes of trees grown by the
PRODUCT search driver, re-decided by a checker that shares NO code with it.

    sample  (GPU)   grow --cells root cells of the box (tools.delaunay_roots: the reference's
                    roots) with the product driver, draw --per-cell nodes of every tree and write
                    their exported records to --out (npz)
    check   (CPU)   every sampled node against ONE mixed-integer LP per oracle call, HiGHS'
                    branch-and-bound on the reference's own big-M statement of the law
                    (oracle/milp_check.py <- lib/oracle.py:42-102, lib/mpc_library.py:521-552) and
                    the uncondensed fixed-sequence LP (oracle/lp_models.py):

      a cell split WITHOUT a commutation (lib/worker.py:268-291)
            V_R as one MILP over p+1 trajectory copies sharing the mode indicators: infeasible
      a cell that holds a commutation
            its vertex costs = the optimum of that sequence at every vertex (9 LPs, 1e-7);
            where the commutation was adopted AT this cell (root, or parent without one): it is
            the FIRST sequence in enumeration order feasible at every vertex -- V_R's MILP with
            the objective sum_k 4^(N-1-k) mode_k (lexicographic minimum) returns it
      a closed leaf   bar_E as one MILP in decision form: max t < 0
      an lcss split   bar_E's MILP: max t >= 0; children = longest-edge bisection
                      (oracle/geometry.py), their commutation the node's or bar_D's MILP optimum

    Nothing of bnb.py / bnb_frontier.py / sequences.py / csrc/ehm_search.cpp is imported by
    ``check``: prefix relaxations, memos, inherited bounds, tie-break walks -- none of the
    product's search logic takes part in the verdict it is checked against.

    python tools/config5_independent_check.py sample --cells 5 --per-cell 44 --out gpurun_out/c5_samples.npz
    python tools/config5_independent_check.py check gpurun_out/c5_samples.npz --procs 8 --out profiles/r5/config5_independent_check.json

Test infrastructure (imports oracle/).
"""

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ABS_FRAC, EPS_R, N_STEPS, SEED = 0.2, 1e-3, 8, 0
RTOL = 1e-7
ROUTE_TOL = 1e-6


# ---------------------------------------------------------------------------------------------
def cmd_sample(args):
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier, examples, frontier
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from explicit_hybrid_mpc_amd.tree import NodeData, Tree
    mpc = examples.pwa4_mpc(N=N_STEPS, seed=SEED)
    half = examples.theta_box(mpc)
    V = examples.box_vertices(half)
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=8192, device=0)
    eps_a = float(np.max([j for _, _, j in bnb_frontier.p_theta_many(orc, ABS_FRAC * V)]))
    orc.eps_a, orc.eps_r = eps_a, EPS_R
    orc.table.set_eps(eps_a, EPS_R)
    roots, _ = ehm_tools.delaunay_roots(V)
    native = frontier.NativeFrontier(mpc, eps_a, EPS_R, slots=8192) if args.driver == 'native' \
        else None
    rng = np.random.default_rng(args.seed)
    cells = [int(c) for c in args.cell_list.split(',')] if args.cell_list else list(range(args.cells))
    rec = dict(cell=[], loc=[], kind=[], R=[], seq=[], V=[], adopted_here=[], kids_R=[],
               kids_seq=[])
    nv, p = roots.shape[1], roots.shape[2]
    for c in cells:
        orc.table.forget()
        tree = Tree(NodeData(vertices=roots[c].copy()))
        t0 = time.perf_counter()
        if native is not None:          # the product's default driver (bench.py --driver native)
            st = frontier.grow_cells(native, tree, slow_oracle=lambda: orc, slow_opts=dict(
                order='lcss-first', table_backoff=True, round_cap=4096),
                max_depth=args.max_depth, deadline=time.perf_counter() + args.cell_seconds,
                slice_visits=20000)
        else:
            st = bnb_frontier.grow_frontier(orc, tree, 'ecc', order='lcss-first',
                                            table_backoff=True, round_cap=4096)
        nodes = list(tree.walk())
        has = {loc: hasattr(nd.data, 'commutation') for nd, loc in nodes}
        # kinds: 0 = split without a commutation, 1 = closed leaf, 2 = lcss split
        kinds = {}
        for nd, loc in nodes:
            if nd.is_leaf():
                # (open leaves: the depth limit, or a cell cut short by --cell-seconds -- undecided,
                # nothing to check)
                kinds[loc] = 1 if nd.data.is_epsilon_suboptimal else -1
            else:
                kinds[loc] = 2 if has[loc] else 0
        by_kind = {k: [i for i, (nd, loc) in enumerate(nodes) if kinds[loc] == k] for k in (0, 1, 2)}
        print('cell %d: %d nodes (%d ecc splits, %d closed leaves, %d lcss splits), %.1f s, %d '
              'rounds' % (c, len(nodes), len(by_kind[0]), len(by_kind[1]), len(by_kind[2]),
                          time.perf_counter() - t0, st['rounds']), flush=True)
        # a third of the sample each, what a kind cannot fill goes to the closed leaves; half of
        # the closed leaves among those that adopted their commutation themselves
        want = {0: args.per_cell // 3, 2: min(len(by_kind[2]), args.per_cell // 3)}
        want[1] = args.per_cell - want[0] - want[2]
        picks = []
        for k in (0, 2):
            picks += list(rng.choice(by_kind[k], size=min(want[k], len(by_kind[k])), replace=False))
        own = [i for i in by_kind[1] if nodes[i][1] == '' or not has[nodes[i][1][:-1]]]
        n_own = min(len(own), want[1] // 2)
        picks += list(rng.choice(own, size=n_own, replace=False))
        rest = [i for i in by_kind[1] if i not in set(picks)]
        picks += list(rng.choice(rest, size=min(len(rest), want[1] - n_own), replace=False))
        for i in picks:
            nd, loc = nodes[i]
            d = nd.data
            rec['cell'].append(c)
            rec['loc'].append(loc)
            rec['kind'].append(kinds[loc])
            rec['R'].append(np.asarray(d.vertices, dtype=np.float64))
            rec['seq'].append(np.array(orc.sequence_of(d.commutation), dtype=np.int32)
                              if has[loc] else np.full(mpc.N, -1, dtype=np.int32))
            rec['V'].append(np.asarray(d.vertex_costs, dtype=np.float64) if has[loc]
                            else np.full(nv, np.nan))
            rec['adopted_here'].append(bool(has[loc] and (loc == '' or not has[loc[:-1]])))
            if nd.is_leaf():
                rec['kids_R'].append(np.full((2, nv, p), np.nan))
                rec['kids_seq'].append(np.full((2, mpc.N), -1, dtype=np.int32))
            else:
                rec['kids_R'].append(np.array([np.asarray(k.data.vertices, dtype=np.float64)
                                               for k in (nd.left, nd.right)]))
                rec['kids_seq'].append(np.array([
                    np.array(orc.sequence_of(k.data.commutation), dtype=np.int32)
                    if hasattr(k.data, 'commutation') else np.full(mpc.N, -1, dtype=np.int32)
                    for k in (nd.left, nd.right)]))
    orc.close()
    if native is not None:
        native.close()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, eps_a=eps_a, eps_r=EPS_R, cell=np.array(rec['cell']),
                        loc=np.array(rec['loc']), kind=np.array(rec['kind']),
                        R=np.array(rec['R']), seq=np.array(rec['seq']), V=np.array(rec['V']),
                        adopted_here=np.array(rec['adopted_here']),
                        kids_R=np.array(rec['kids_R']), kids_seq=np.array(rec['kids_seq']))
    print('%d nodes of %d cells written to %s (eps_a %.9g)' % (len(rec['cell']), len(cells),
                                                                args.out, eps_a))
    return 0


# ---------------------------------------------------------------------------------------------
def v_r_first_milp(mpc, R, big_m=50.):
    """V_R (lib/oracle.py:57-66, 175-218) as ONE MILP whose objective makes the optimum the FIRST
    sequence in enumeration order among those feasible at every vertex (the canonical rule):
    minimise sum_k n_modes^(N-1-k) * mode_k.  None if no sequence is feasible at every vertex."""
    from oracle import milp_check
    R = np.asarray(R, dtype=float)
    M = milp_check._model_copies(mpc, big_m, R.shape[0])
    for c, v in enumerate(R):
        r = M['blank'](mpc.n_x)
        r[:, M['x0'](c)] = np.eye(mpc.n_x)
        M['add'](r, v, v)
    cost = np.zeros(M['nv'])
    d0 = M['d'].start
    for k in range(mpc.N):
        for i in range(mpc.delta_size):
            cost[d0 + k * mpc.delta_size + i] = i * float(mpc.delta_size) ** (mpc.N - 1 - k)
    res = milp_check._solve(M, cost)
    return milp_check._sequence(mpc, res, M) if res.status == 0 else None


def _check_one(job):
    from explicit_hybrid_mpc_amd import examples          # the law's data only (A, B, regions)
    from oracle import geometry, milp_check
    from oracle.lp_models import FixedCommutationModel
    from scipy.optimize import linprog
    (idx, kind, R, seq, V, adopted_here, kids_R, kids_seq, eps_a, eps_r) = job
    mpc = _check_one.mpc = getattr(_check_one, 'mpc', None) or examples.pwa4_mpc(N=N_STEPS,
                                                                                  seed=SEED)
    out = dict(idx=int(idx), kind=int(kind), ok=True, routed=False, notes=[], milps=0, lps=0)
    t0 = time.perf_counter()

    def fail(msg):
        out['ok'] = False
        out['notes'].append(msg)
    seq = tuple(int(i) for i in seq)
    if kind == 0:
        out['milps'] += 1
        s = milp_check.v_r_milp(mpc, R)
        if s is not None:
            fail('V_R MILP finds %s feasible at every vertex of a cell the product split' % (s,))
    else:
        model = FixedCommutationModel(mpc, seq)
        worst = 0.
        for i, v in enumerate(R):
            lp = model.lp_point(v)
            res = linprog(lp['c'], A_ub=lp['A_ub'], b_ub=lp['b_ub'], A_eq=lp['A_eq'],
                          b_eq=lp['b_eq'], bounds=lp.get('bounds', (None, None)), method='highs',
                          options=dict(primal_feasibility_tolerance=1e-10,
                                       dual_feasibility_tolerance=1e-10))
            out['lps'] += 1
            if res.status != 0:
                fail('the cell\'s sequence is infeasible at vertex %d' % i)
                continue
            worst = max(worst, abs(res.fun - V[i]) / (1. + abs(V[i])))
        out['max_cost_diff'] = worst
        if worst > RTOL:
            fail('vertex costs differ by %g (relative)' % worst)
        if adopted_here:
            out['milps'] += 1
            first = v_r_first_milp(mpc, R)
            if first != seq:
                # lib/worker.py:396-401: a cell may adopt bar_D's commutation IN PLACE after V_R's
                # -- then the one it holds is bar_D's optimum for V_R's vertex costs
                swapped = False
                if first is not None:
                    m1 = FixedCommutationModel(mpc, first)
                    V1 = []
                    for v in R:
                        lp = m1.lp_point(v)
                        r1 = linprog(lp['c'], A_ub=lp['A_ub'], b_ub=lp['b_ub'], A_eq=lp['A_eq'],
                                     b_eq=lp['b_eq'], bounds=(None, None), method='highs',
                                     options=dict(
                                         primal_feasibility_tolerance=1e-10,
                                         dual_feasibility_tolerance=1e-10))
                        V1.append(r1.fun if r1.status == 0 else np.nan)
                    out['lps'] += len(V1)
                    out['milps'] += 1
                    td, s_d, _ = milp_check.bar_d_milp(mpc, R, np.array(V1), eps_a, eps_r)
                    swapped = s_d == seq and td >= 0.
                    out['notes'].append('adopted in place after V_R\'s %s' % (first,))
                if not swapped:
                    fail('first feasible sequence by the MILP %s, the product holds %s' %
                         (first, seq))
        # bar_E in decision form: max t.  HiGHS' branch-and-bound on a big-M model is itself a
        # numerical method (one of the first 160 nodes came back "optimal" at -0.0057 with M = 50
        # and at +0.0016 with M = 10 and M = 200): every run returns a FEASIBLE point of the
        # reference's problem, so the largest t over several M is a certified lower bound of the
        # maximum -- t >= 0 from any run proves the cell open; a closed leaf must come out
        # negative for every M tried
        out['milps'] += 1
        t, s_e = milp_check.bar_e_milp(mpc, R, V, eps_a, eps_r)
        tried = {50.: float(t)}
        more = (10.,) if kind == 1 else ((10., 200.) if (kind == 2 and not t >= 0.) else ())
        for big_m in more:
            out['milps'] += 1
            t2, s2 = milp_check.bar_e_milp(mpc, R, V, eps_a, eps_r, big_m=big_m)
            tried[big_m] = float(t2)
            if t2 > t:
                t, s_e = t2, s2
        out['t_max'] = float(t)
        out['t_max_by_big_m'] = tried
        near = abs(t) < ROUTE_TOL * (1. + float(np.max(np.abs(V))))
        if kind == 1 and not t < 0.:
            if near:
                out['routed'] = True
            else:
                fail('closed leaf, but bar_E\'s MILP is feasible: max t = %g (%s)' % (t, s_e))
        if kind == 2:
            if not t >= 0.:
                if near:
                    out['routed'] = True
                else:
                    fail('lcss split, but bar_E\'s MILP is infeasible: max t = %g' % t)
            S1, S2, _ = geometry.split_along_longest_edge(R)
            if not (np.array_equal(S1, kids_R[0]) and np.array_equal(S2, kids_R[1])):
                fail('children are not the longest-edge bisection')
            out['milps'] += 1
            td, s_d, _ = milp_check.bar_d_milp(mpc, R, V, eps_a, eps_r)
            allowed = {seq} | ({s_d} if s_d is not None and td >= 0. else set())
            for k in range(2):
                ks = tuple(int(i) for i in kids_seq[k])
                if ks in allowed:
                    continue
                # bar_D's optimum is often attained by several sequences (modes that are
                # interchangeable in the overlap band, or at the last step): the canonical rule
                # takes the first in enumeration order, the MILP returns any.  The child's
                # sequence is accepted where it is one of them -- feasible at every vertex and its
                # own slack (a fixed-sequence LP, oracle/prefix_bb.py) within the canonical tie
                # tolerance of the MILP's maximum
                from oracle import prefix_bb
                t_child = prefix_bb.prefix_slack(mpc, ks, R, V, eps_a, eps_r)
                out['lps'] += 1
                mk = FixedCommutationModel(mpc, ks)
                feas = True
                for v in R:
                    lp = mk.lp_point(v)
                    rk = linprog(lp['c'], A_ub=lp['A_ub'], b_ub=lp['b_ub'], A_eq=lp['A_eq'],
                                 b_eq=lp['b_eq'], bounds=(None, None), method='highs')
                    out['lps'] += 1
                    feas = feas and rk.status == 0
                # (the MILP's objective carries its feasibility tolerance -- 4.6e-6 at one of the
                # first 240 nodes --: its SEQUENCE is re-priced by the same fixed-sequence LP)
                td_lp = prefix_bb.prefix_slack(mpc, s_d, R, V, eps_a, eps_r) if s_d else td
                out['lps'] += 1
                td = min(td, td_lp)
                tie = feas and t_child >= td - 1e-6 * (1. + abs(td)) and t_child >= 0.
                if tie:
                    out['notes'].append('child %d: %s ties with the MILP optimum %s (t %.9g / %.9g)'
                                        % (k, ks, s_d, t_child, td))
                    break               # (both children hold the same commutation)
                fail('child %d holds %s (t = %g, feasible at every vertex: %s); the node %s, '
                     'bar_D\'s MILP optimum %s (t = %g)' % (k, ks, t_child, feas, seq, s_d, td))
    if kind == 0:
        S1, S2, _ = geometry.split_along_longest_edge(R)
        if not (np.array_equal(S1, kids_R[0]) and np.array_equal(S2, kids_R[1])):
            fail('children are not the longest-edge bisection')
    out['seconds'] = time.perf_counter() - t0
    return out


def cmd_check(args):
    z = np.load(args.samples)
    eps_a, eps_r = float(z['eps_a']), float(z['eps_r'])
    n = len(z['kind'])
    jobs = [(i, int(z['kind'][i]), z['R'][i], z['seq'][i], z['V'][i], bool(z['adopted_here'][i]),
             z['kids_R'][i], z['kids_seq'][i], eps_a, eps_r) for i in range(n)]
    if args.limit:
        jobs = jobs[:args.limit]
    os.environ.update(OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    t0 = time.perf_counter()
    res = []
    # every finished node is appended to <out>.jsonl at once: a run that is stopped keeps what it
    # has, and the next one goes on with the nodes that are missing
    log = (args.out or args.samples) + '.jsonl'
    if os.path.exists(log):
        res = [json.loads(line) for line in open(log) if line.strip()]
        done = {r['idx'] for r in res}
        jobs = [j for j in jobs if j[0] not in done]
        print('%d nodes already checked (%s), %d to go' % (len(res), log, len(jobs)), flush=True)
    # the cheap kinds first (one MILP per split without a commutation), so that all cells are
    # covered early
    jobs.sort(key=lambda j: (j[1] != 0, j[0]))
    if args.summary_only:
        jobs = []
    with mp.get_context('spawn').Pool(args.procs) as pool, open(log, 'a') as lf:
        for r in pool.imap_unordered(_check_one, jobs):
            res.append(r)
            lf.write(json.dumps(r) + '\n')
            lf.flush()
            if len(res) % 20 == 0:
                print('  %d / %d nodes, %d failed, %.0f s' % (
                    len(res), len(jobs), sum(not x['ok'] for x in res), time.perf_counter() - t0),
                    flush=True)
    res.sort(key=lambda r: r['idx'])
    kinds = {0: 'split without a commutation (V_R infeasible)', 1: 'closed leaf',
             2: 'lcss split'}
    summary = dict(
        what='configs[4]: nodes of trees grown by the product search driver (bnb_frontier on the '
             'device tables), re-decided by one MILP per oracle call (oracle/milp_check.py, HiGHS '
             'branch-and-bound on the big-M statement of the law) and uncondensed fixed-sequence '
             'LPs; no code of the product\'s search takes part',
        eps_a=eps_a, eps_r=eps_r, nodes=len(res), nodes_sampled=int(n),
        cells=sorted(set(int(z['cell'][r['idx']]) for r in res)),
        by_kind={kinds[k]: dict(nodes=sum(r['kind'] == k for r in res),
                                agree=sum(r['kind'] == k and r['ok'] and not r['routed']
                                          for r in res),
                                routed=sum(r['kind'] == k and r['routed'] for r in res),
                                failed=sum(r['kind'] == k and not r['ok'] for r in res))
                 for k in kinds},
        adopted_here_checked_against_the_lexicographic_milp=int(
            sum(bool(z['adopted_here'][r['idx']]) for r in res)),
        max_vertex_cost_difference=max([r.get('max_cost_diff', 0.) for r in res] + [0.]),
        milps=sum(r['milps'] for r in res), lps=sum(r['lps'] for r in res),
        failed=[dict(idx=r['idx'], cell=int(z['cell'][r['idx']]), loc=str(z['loc'][r['idx']]),
                     notes=r['notes']) for r in res if not r['ok']],
        smallest_margins=sorted(abs(r['t_max']) for r in res if 't_max' in r)[:5],
        wall_seconds=time.perf_counter() - t0, procs=args.procs)
    print(json.dumps(summary, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as f:
            json.dump(summary, f, indent=1)
    return 1 if summary['failed'] else 0


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest='cmd', required=True)
    a = sub.add_parser('sample')
    a.add_argument('--cells', type=int, default=5)
    a.add_argument('--cell-list', default='', help='comma-separated root indices instead')
    a.add_argument('--per-cell', type=int, default=44)
    a.add_argument('--seed', type=int, default=0)
    a.add_argument('--driver', choices=['native', 'python'], default='native')
    a.add_argument('--max-depth', type=int, default=28)
    a.add_argument('--cell-seconds', type=float, default=30.)
    a.add_argument('--out', default='gpurun_out/c5_samples.npz')
    b = sub.add_parser('check')
    b.add_argument('samples')
    b.add_argument('--procs', type=int, default=8)
    b.add_argument('--limit', type=int, default=0)
    b.add_argument('--summary-only', action='store_true',
                   help='no new checks: the summary of what <out>.jsonl holds')
    b.add_argument('--out', default='')
    args = ap.parse_args()
    return cmd_sample(args) if args.cmd == 'sample' else cmd_check(args)


if __name__ == '__main__':
    sys.exit(main())
