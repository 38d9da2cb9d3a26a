"""
Study (CPU, test infrastructure -- it grows trees with the oracle): which reading of the
reference's mixed-integer FEASIBILITY problems reproduces its published cwh_z leaf counts?

lib/post_process.py:489,526 holds leaves / depth of the reference's own cwh_z runs: 101 / 13,
978 / 17, 13 500 / 20, ...  The reference's V_R and bar_D minimise 0 (lib/oracle.py:201,347): which
feasible commutation comes back is MOSEK's choice, and bar_E / bar_D are "feasible" within the
solver's tolerances.  This build uses a canonical rule (V_R: first feasible in enumeration order;
bar_D: the largest slack) and exact verdicts (t* >= 0): 78 / 12 and 761 / 14 on the first two jobs.
Here the CPU oracle grows the same jobs under other readings:

    rule 'best'   canonical (what the device runs)
    rule 'first'  bar_D returns the first commutation with t* >= 0
    rule 'random' V_R and bar_D return a uniformly drawn admissible commutation (seeded)
    tau           verdicts taken at t* >= -tau (1 + |V_0|) (a solver's feasibility tolerance)

    python tools/cwh_leafcount_study.py [jobs=1,2] [procs=8]
"""
import multiprocessing as mp
import sys
import time

import numpy as np

sys.path.insert(0, '.')
JOBS = {1: (0.5, 2.0, 0.048658577500541, 101, 13), 2: (0.25, 1.0, 0.012183769272642, 978, 17),
        3: (0.1, 0.1, 0.001957893965646, 13500, 20)}


def run(case):
    job, rule, seed, tau = case
    from explicit_hybrid_mpc_amd import examples
    from oracle import geometry
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    from oracle.satellite_cpu import SatelliteZCPU
    abs_frac, rel_err, eps_a, _, _ = JOBS[job]
    mpc = examples.satellite_z(4)
    roots, locs = geometry.delaunay_simplices(examples.box_vertices(examples.theta_box(mpc)))
    orc = OracleCPU(SatelliteZCPU(4), eps_a, rel_err)
    orc.memoize = True
    orc.bar_d_rule = rule
    orc.rng = np.random.default_rng(seed)
    orc.verdict_tol = tau
    orc.qp_accept = 1e-6            # like the reference's OPTIMAL_INACCURATE
    part = PartitionCPU(orc)
    t0 = time.time()
    try:
        part.run(roots, locs, 'ecc')
    except Exception as e:          # a QP the CPU oracle's solver gives up on ends this case only
        return case, -1, -1, orc.n_solves, time.time() - t0, '%s: %s' % (type(e).__name__, e)
    leaves = part.leaves()
    depth = max(len(loc) for loc in leaves) - min(len(loc) for loc in locs)
    return case, len(leaves), depth, orc.n_solves, time.time() - t0, ''


def main():
    jobs = [int(j) for j in (sys.argv[1] if len(sys.argv) > 1 else '1,2').split(',')]
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cases = []
    for job in jobs:
        if job == 1:
            cases += [(job, 'best', 0, tau) for tau in (0., 1e-8, 1e-6, 1e-4)]
            cases += [(job, 'first', 0, 0.)]
            cases += [(job, 'random', seed, 0.) for seed in range(8)]
        elif job == 2:              # minutes per case: the canonical rule, one tolerance, four draws
            cases += [(job, 'best', 0, 0.), (job, 'best', 0, 1e-6)]
            cases += [(job, 'random', seed, 0.) for seed in range(4)]
        else:                       # hours per case: the canonical rule and two draws
            cases += [(job, 'best', 0, 0.)]
            cases += [(job, 'random', seed, 0.) for seed in range(2)]
    with mp.get_context('spawn').Pool(procs) as pool:
        for (job, rule, seed, tau), leaves, depth, solves, secs, err in pool.imap_unordered(run, cases):
            print('job %d (reference: %d leaves, depth %d)  rule %-6s seed %d tau %-7g -> %6d leaves, '
                  'depth %2d, %8d solves, %5.0f s' % (job, JOBS[job][3], JOBS[job][4], rule, seed,
                                                     tau, leaves, depth, solves, secs) +
                  ('   FAILED ' + err if err else ''), flush=True)


if __name__ == '__main__':
    main()
