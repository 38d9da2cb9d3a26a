"""
Study (CPU, test infrastructure -- it grows the tree with the oracle): how many of the
suboptimality-test problems of a HYBRID partition does the parent already answer?

The multi-commutation engine solves, on every node it visits, one suboptimality-test problem per
candidate commutation: "closed" needs every optimum t*_d negative, bar_D needs the argmax
(lib/oracle.py:285-414).  On the reference's cwh_z jobs these ranking solves are 93 % of all LPs
(DESIGN.md section 7c).  A child simplex lies inside its parent and, as long as it inherits the
commutation (or adopts one whose vertex costs are nowhere larger), the interpolant of its vertex
costs lies below the parent's -- so for EVERY commutation d

        t*_d(child)  <=  t*_d(parent)   (+ the largest increase of a vertex cost, if any).

This script grows a cwh_z job with the CPU oracle, records t*_d for every commutation on every
visited node, checks the inequality on every parent -> child pair it applies to, and counts what a
warm-started visit would have had to solve:

    closed test:  only the d with t*_d(parent) >= 0;
    ranking:      the parent's best commutation first (an incumbent t0), then only the d with
                  t*_d(parent) >= max(0, t0) - tolerance.

    PYTHONPATH=. python tools/study_ranking_bounds.py [job=2]      (job k of make_jobs.sh:60-66; 1..3 on a CPU)
"""
import sys
import time

import numpy as np

JOBS = {1: (0.5, 0.048658577500541), 2: (0.25, 0.012183769272642), 3: (0.1, 0.001957893965646)}
TIE = 1e-7


def main():
    job = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    from explicit_hybrid_mpc_amd import examples
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    from oracle.satellite_cpu import SatelliteZCPU
    from oracle import geometry
    abs_frac, eps_a = JOBS[job]
    mpc = examples.satellite_z(4)
    full_set = examples.box_vertices(examples.theta_box(mpc))
    roots, locs = geometry.delaunay_simplices(full_set)

    class Recording(OracleCPU):
        """Keeps t*_d of the current node for every commutation the oracles evaluate."""
        current = None

        def slack(self, R, V, d):
            t, a = OracleCPU.slack(self, R, V, d)
            if self.current is not None:
                self.current[d] = t
            return t, a

    orc = Recording(SatelliteZCPU(4), eps_a, 2.0)
    orc.memoize = True
    part = PartitionCPU(orc)
    record = {}             # location -> list of visits: (vertex costs, commutation index, {d: t})
    inner = part._lcss_visit

    def visit(loc, work):
        node = part.nodes[loc]
        orc.current = {}
        V, d = np.array(node['vertex_costs']), orc.delta_index(node['commutation'])
        inner(loc, work)
        record.setdefault(loc, []).append((V, d, orc.current))
        orc.current = None
    part._lcss_visit = visit
    t0 = time.time()
    part.run(roots, locs, 'ecc')
    n_delta = len(orc.models)
    visits = sum(len(v) for v in record.values())
    print('cwh_z job %d (abs_frac %g, eps_a %.6g): %d nodes, %d leaves, %d lcss visits, %d '
          'commutations, %d LP solves, %.0f s' % (
              job, abs_frac, eps_a, len(part.nodes), len(part.leaves()), visits, n_delta,
              orc.n_solves, time.time() - t0))
    pairs = applies = violated = shifted = 0
    cold = warm_closed = warm_rank = 0
    worst = 0.
    for loc, vis in record.items():
        if len(loc) <= len(locs[0]) or loc[:-1] not in record:
            continue
        Vp, dp, tp = record[loc[:-1]][-1]          # the parent's last visit (the one that split)
        Vc, dc, tc = vis[0]                        # the child's first visit
        pairs += 1
        Rp, Rc = part.nodes[loc[:-1]]['vertices'], part.nodes[loc]['vertices']
        same = np.all(Rp == Rc, axis=1)            # the vertices the child shares with its parent
        # a commutation with larger vertex costs was adopted: the interpolant rises by at most
        # shift = the largest increase, and so does every t*_d
        shift = max(0., float(np.max(Vc[same] - Vp[same])))
        shifted += shift > 0.
        tp = {d: t + shift for d, t in tp.items()}
        applies += 1
        for d, t in tc.items():
            if d in tp and np.isfinite(t) and np.isfinite(tp[d]):
                worst = max(worst, t - tp[d])
                violated += t > tp[d] + 1e-7 * (1. + abs(tp[d]))
        cold += len(tc)
        # closed test: what the parent does not refute
        need = [d for d in tc if not (d in tp and tp[d] < -TIE)]
        warm_closed += len(need)
        # ranking: incumbent = the parent's best commutation, evaluated on the child first
        best_p = max(tp, key=lambda d: tp[d]) if tp else None
        t_inc = tc.get(best_p, -np.inf)
        floor = max(0., t_inc) if np.isfinite(t_inc) else 0.
        need = [d for d in tc if d == best_p or not (d in tp and tp[d] < floor - TIE * (1 + floor))]
        warm_rank += len(need)
    print('parent -> child pairs %d (%d of them adopt a commutation with larger vertex costs: bound '
          'shifted by the largest increase); inequality violated %d times (largest t_child - bound '
          '%.2e)' % (pairs, shifted, violated, worst))
    print('on those children: %d problems solved cold; a closed test that skips what the parent refutes '
          'needs %d (%.1f %%); a ranking that also starts from the parent\'s best needs %d (%.1f %%)' % (
              cold, warm_closed, 100. * warm_closed / max(cold, 1), warm_rank,
              100. * warm_rank / max(cold, 1)))


if __name__ == '__main__':
    main()
