"""Full-size (bench workload) tree identity across the three numerical paths (gpurun)."""
import sys, time
import numpy as np
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools

abs_frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
mpc = examples.linear_mpc(seed=0)
can = mpc.compile()
gp = engine.GpuProblem(can, 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
J, _, _ = gp.solve_pt(abs_frac * V)
eps_a = float(J.max())
gp.set_eps(eps_a, 1e-2)
roots, _ = ehm_tools.delaunay_roots(V)
trees = {}
for name, gen, full in (('g2_sign', 2, 0), ('g2_full', 2, 1), ('g1', 1, 1)):
    gp.set_solver(gen)
    gp.set_option('decide_full', full)
    t0 = time.time()
    flat = gp.partition(roots, action='ecc', max_nodes=1 << 22)
    trees[name] = flat
    print(name, flat.n_nodes, flat.info['n_closed'], 'min_margin', flat.info['min_margin'],
          '%.2fs' % (time.time() - t0), flush=True)
ref = trees['g2_full']
for name, t in trees.items():
    same = (t.n_nodes == ref.n_nodes and np.array_equal(t.left, ref.left) and
            np.array_equal(t.flags & 1, ref.flags & 1) and np.array_equal(t.vertices, ref.vertices))
    print(name, 'identical to g2_full:', same)
    if same:
        dv = np.max(np.abs(t.vertex_costs - ref.vertex_costs) / (1 + np.abs(ref.vertex_costs)))
        print('   max rel vertex-cost difference', dv)
    else:
        n = min(t.n_nodes, ref.n_nodes)
        diff = np.nonzero((t.flags[:n] & 1) != (ref.flags[:n] & 1))[0]
        print('   first differing closed flags at', diff[:10], 'of', len(diff))
        if len(diff):
            k = diff[0]
            print('   tstar', name, t.tstar[k] if hasattr(t, 'tstar') else None)
gp.close()
