"""
Where the host time of the config-5 search driver goes, measured WITHOUT a GPU.

`bnb_frontier.grow_frontier` on the whole-box cell of `examples.pwa4_mpc(N=8)` (DESIGN.md section
3.3e) with the device problem replaced by a pool of HiGHS processes that solve the CONDENSED blocks
the table writes into its slots (the blocks live in shared memory; a launch of n problems is dealt
over the processes).  The time inside the stand-in is clocked separately, so

    host time = wall - time inside the solver stand-in

is what the search's own bookkeeping costs (Python lists, the prefix -> slot map, the feasibility
memo, the numpy condensation of new blocks) at a given number of node visits -- the part section
7c item 1 wants off the interpreter.  The tree is summarised by a digest over its node records so
that two builds of the bookkeeping can be compared on the same stand-in.

    python tools/seq_host_profile.py [visits=2000] [procs=8] [profile]
"""
import hashlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_SH = {}


def _shared(shape):
    n = int(np.prod(shape))
    raw = mp.RawArray('d', n)
    return np.frombuffer(raw, dtype=np.float64).reshape(shape)


def _solve_points(job):
    from scipy.optimize import linprog
    lo, hi, feas = job
    G, w, S, c = _SH['G'], _SH['w'], _SH['S'], _SH['c']
    theta, slot = _SH['theta'], _SH['slot']
    J = np.zeros(hi - lo)
    u0 = np.zeros((hi - lo, _SH['n_u']))
    for k in range(lo, hi):
        s = int(slot[k])
        g, rhs = G[s], w[s] + S[s] @ theta[k]
        if feas:
            A = np.hstack([g, -np.ones((g.shape[0], 1))])
            cc = np.zeros(A.shape[1])
            cc[-1] = 1.
            res = linprog(cc, A_ub=A, b_ub=rhs,
                          bounds=[(None, None)] * g.shape[1] + [(-1., None)], method='highs')
            J[k - lo] = res.fun
        else:
            res = linprog(c, A_ub=g, b_ub=rhs, bounds=(None, None), method='highs')
            J[k - lo] = res.fun if res.status == 0 else np.nan
            if res.status == 0:
                u0[k - lo] = res.x[:_SH['n_u']]
    return J, u0


def _solve_simplices(job):
    from scipy.optimize import linprog
    lo, hi, mode, eps_a, eps_r = job
    G, w, S, c = _SH['G'], _SH['w'], _SH['S'], _SH['c']
    R, slot, Vbar = _SH['R'], _SH['slot'], _SH['Vbar']
    n = G.shape[2]
    na = R.shape[1]
    obj = np.zeros(hi - lo)
    alpha = np.zeros((hi - lo, na))
    for k in range(lo, hi):
        s = int(slot[k])
        g, ww, ss = G[s], w[s], S[s]
        m = g.shape[0]
        extra = 0 if mode == 0 else 1
        A = np.zeros((m + (2 if mode == 1 else 0), n + na + extra))
        A[:m, :n], A[:m, n:n + na] = g, -ss @ R[k].T
        b = np.concatenate([ww, np.zeros(2 if mode == 1 else 0)])
        cc = np.zeros(n + na + extra)
        bounds = [(None, None)] * n + [(0., None)] * na
        if mode == 0:
            cc[:n] = c
        elif mode == 2:
            A[:m, -1] = -1.
            cc[-1] = 1.
            bounds.append((-1., None))
        else:
            for r, scale, shift in ((m, 1., eps_a), (m + 1, 1. + eps_r, 0.)):
                A[r, :n], A[r, n:n + na], A[r, -1] = scale * c, -Vbar[k], 1.
                b[r] = -shift
            cc[-1] = -1.
            bounds.append((None, None))
        Aeq = np.zeros((1, A.shape[1]))
        Aeq[0, n:n + na] = 1.
        res = linprog(cc, A_ub=A, b_ub=b, A_eq=Aeq, b_eq=[1.], bounds=bounds, method='highs')
        # (an optimisation problem that comes back infeasible was sent without its phase one:
        # the device solver would not notice -- counted, must stay 0)
        obj[k - lo] = (-res.fun if mode == 1 else res.fun) if res.status == 0 else np.nan
        if res.status == 0:
            alpha[k - lo] = res.x[n:n + na]
    return obj, alpha


class PooledStubProblem:
    """Stands in for engine.GpuProblem under sequences.PrefixTable (HiGHS on the condensed blocks,
    the launches dealt over a process pool); `busy` accumulates the wall time spent inside."""

    CAP = 1 << 17            # problems per launch the shared argument arrays hold
    procs = 8
    busy = 0.
    launches = 0
    problems = 0
    by_kind = {}

    def __init__(self, can, eps_a, eps_r, device=0):
        self.can = can
        self.eps_a, self.eps_r = eps_a, eps_r
        p = can.S.shape[2]
        _SH['G'], _SH['w'], _SH['S'] = _shared(can.G.shape), _shared(can.w.shape), _shared(can.S.shape)
        _SH['G'][:], _SH['w'][:], _SH['S'][:] = can.G, can.w, can.S
        _SH['c'] = np.array(can.c)
        _SH['n_u'] = can.n_u
        _SH['theta'] = _shared((self.CAP, p))
        _SH['slot'] = _shared((self.CAP,))
        _SH['R'] = _shared((self.CAP, p + 1, p))
        _SH['Vbar'] = _shared((self.CAP, p + 1))
        self.pool = mp.get_context('fork').Pool(self.procs)

    def set_eps(self, eps_a, eps_r):
        self.eps_a, self.eps_r = eps_a, eps_r

    def close(self):
        self.pool.terminate()

    def update_blocks(self, first, G, w, S):
        t = time.perf_counter()
        n = G.shape[0]
        _SH['G'][first:first + n], _SH['w'][first:first + n], _SH['S'][first:first + n] = G, w, S
        PooledStubProblem.busy += time.perf_counter() - t

    def _deal(self, n):
        per = max(1, -(-n // (4 * self.procs)))
        return [(lo, min(n, lo + per)) for lo in range(0, n, per)]

    def point_idx(self, theta, slot, feas=False):
        t = time.perf_counter()
        theta = np.atleast_2d(theta)
        n = theta.shape[0]
        assert n <= self.CAP
        _SH['theta'][:n], _SH['slot'][:n] = theta, slot
        parts = self.pool.map(_solve_points, [(lo, hi, feas) for lo, hi in self._deal(n)])
        J = np.concatenate([q[0] for q in parts]) if parts else np.zeros(0)
        u0 = np.concatenate([q[1] for q in parts]) if parts else np.zeros((0, self.can.n_u))
        PooledStubProblem.busy += time.perf_counter() - t
        PooledStubProblem.launches += 1
        PooledStubProblem.problems += n
        kind = 'point phase one' if feas else 'point optimum'
        bad = int(np.isnan(J).sum())
        if bad:
            PooledStubProblem.by_kind['UNSOLVABLE ' + kind] = \
                PooledStubProblem.by_kind.get('UNSOLVABLE ' + kind, 0) + bad
        PooledStubProblem.by_kind[kind] = PooledStubProblem.by_kind.get(kind, 0) + n
        return J, u0, np.zeros(n, dtype=np.int32)

    def simplex_idx(self, R, slot, mode, Vbar=None):
        t = time.perf_counter()
        R = np.asarray(R, dtype=np.float64)
        n = R.shape[0]
        assert n <= self.CAP
        _SH['R'][:n], _SH['slot'][:n] = R, slot
        if Vbar is not None:
            _SH['Vbar'][:n] = Vbar
        parts = self.pool.map(_solve_simplices, [(lo, hi, mode, self.eps_a, self.eps_r)
                                                 for lo, hi in self._deal(n)])
        obj = np.concatenate([q[0] for q in parts]) if parts else np.zeros(0)
        alpha = np.concatenate([q[1] for q in parts]) if parts else np.zeros((0, R.shape[1]))
        PooledStubProblem.busy += time.perf_counter() - t
        PooledStubProblem.launches += 1
        PooledStubProblem.problems += n
        kind = ('simplex min', 'simplex slack', 'simplex phase one')[mode]
        bad = int(np.isnan(obj).sum())
        if bad:
            PooledStubProblem.by_kind['UNSOLVABLE ' + kind] = \
                PooledStubProblem.by_kind.get('UNSOLVABLE ' + kind, 0) + bad
            obj = np.where(np.isnan(obj), np.inf, obj)
        PooledStubProblem.by_kind[kind] = PooledStubProblem.by_kind.get(kind, 0) + n
        return obj, alpha, np.zeros(n, dtype=np.int32)


def host_split_batch(R):
    """tools.split_along_longest_edge for a batch (lib/tools.py:191-257): first longest edge in
    (i, j) order, child 1 replaces vertex i by the midpoint, child 2 vertex j."""
    R = np.asarray(R, dtype=np.float64)
    n, nv, _ = R.shape
    iu, ju = np.triu_indices(nv, 1)
    d = np.linalg.norm(R[:, iu] - R[:, ju], axis=2)
    k = np.argmax(d, axis=1)
    i, j = iu[k], ju[k]
    rows = np.arange(n)
    mid = 0.5 * (R[rows, i] + R[rows, j])
    S1, S2 = R.copy(), R.copy()
    S1[rows, i] = mid
    S2[rows, j] = mid
    return S1, S2, np.stack([i, j], axis=1).astype(np.int32)


def tree_digest(branch):
    h = hashlib.sha256()
    n = 0
    for nd, loc in branch.walk():
        n += 1
        h.update(loc.encode())
        h.update(np.ascontiguousarray(nd.data.vertices).tobytes())
        h.update(b'1' if nd.data.is_epsilon_suboptimal else b'0')
        com = getattr(nd.data, 'commutation', None)
        if com is not None:
            h.update(np.asarray(com).astype(np.int8).tobytes())
    return n, h.hexdigest()[:16]


def main():
    visits = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    PooledStubProblem.procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    profile = len(sys.argv) > 3 and sys.argv[3] == 'profile'
    from explicit_hybrid_mpc_amd import examples, bnb, bnb_frontier, sequences
    from explicit_hybrid_mpc_amd.tree import Tree, NodeData
    sequences.engine.GpuProblem = PooledStubProblem
    mpc = examples.pwa4_mpc(N=8)
    half = examples.theta_box(mpc)
    p = 8
    R = np.array([-half + 2 * half * (np.arange(p) < k) for k in range(p + 1)])
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=2048)
    J = [orc.P_theta(v)[2] for v in R]
    # config-5 tolerances of DESIGN.md 3.3e by default; EHM_EPS_A_FRAC / EHM_EPS_R tighten them
    # (then the suboptimality-test searches do real work)
    orc.eps_a = float(os.environ.get('EHM_EPS_A_FRAC', '0.5')) * max(J)
    orc.eps_r = float(os.environ.get('EHM_EPS_R', '1.0'))
    orc.table.set_eps(orc.eps_a, orc.eps_r)
    PooledStubProblem.busy = 0.
    PooledStubProblem.by_kind = {}
    t0 = time.perf_counter()
    branch = Tree(NodeData(vertices=R.copy()))
    tables = bool(os.environ.get('EHM_TABLES'))
    if tables:
        # region tables as the hand-off computes them for every open lcss node, WITHOUT the device
        # engine behind them (nothing is handed off): what the tables themselves cost;
        # EHM_TABLES=cold computes them without what the parents found
        cold = os.environ['EHM_TABLES'] == 'cold'

        def tables_only(oracle, nodes, table_max, engine_opts, stats, above=None, costs=None,
                        excess=None):
            Rs = [np.asarray(nd.data.vertices, dtype=np.float64) for nd in nodes]
            out = bnb_frontier.region_tables_many(
                oracle, Rs, [nd.data.commutation for nd in nodes],
                [float(np.max(nd.data.vertex_costs)) for nd in nodes], table_max,
                None if cold else above, costs, excess)
            stats['table_sizes'] += [len(t) for t in out if t is not None]
            stats['tables_too_large'] += sum(t is None for t in out)
            return list(range(len(nodes)))
        bnb_frontier._hand_off = tables_only
    run = lambda: bnb_frontier.grow_frontier(
        orc, branch, 'ecc', max_visits=visits, round_cap=16384, handoff=tables,
        table_backoff=bool(os.environ.get('EHM_TABLE_BACKOFF')),
        split_batch=host_split_batch,
        log=lambda m: print('  ', m, '%.1fs (solver stand-in %.1fs)' % (
            time.perf_counter() - t0, PooledStubProblem.busy), flush=True))
    if profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        stats = pr.runcall(run)
        pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
        pstats.Stats(pr).sort_stats('tottime').print_stats(30)
    else:
        stats = run()
    wall = time.perf_counter() - t0
    n, dig = tree_digest(branch)
    print('visits %d rounds %d nodes %d digest %s | LPs %d launches %d blocks %d | wall %.1fs '
          'solver stand-in %.1fs host %.2fs = %.1f us per visit' % (
              stats['host_visits'], stats['rounds'], n, dig, orc.table.lp_solves,
              PooledStubProblem.launches, orc.table.blocks_loaded, wall, PooledStubProblem.busy,
              wall - PooledStubProblem.busy,
              1e6 * (wall - PooledStubProblem.busy) / max(stats['host_visits'], 1)), flush=True)
    print('problems by kind', PooledStubProblem.by_kind, 'oracle calls', dict(orc.calls),
          'driver', {k: (v if k != 'table_sizes' else sorted(v)[-5:]) for k, v in stats.items()},
          flush=True)
    orc.close()


if __name__ == '__main__':
    main()
