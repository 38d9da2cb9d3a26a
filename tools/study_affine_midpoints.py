"""How many split edges carry an AFFINE optimal cost?  (offline, from an exported tree)

A split solves P_theta_delta at the midpoint of the node's longest edge [v_i, v_j]
(lib/worker.py:406-407).  The optimal cost is convex piecewise affine; where it is affine along
the edge, V*(mid) = (V_i + V_j)/2 and -- with a unique optimiser -- the optimal input at the
midpoint is the mean of the two vertex inputs: the LP tells nothing new.  This counts such edges
(all splits, and the distinct midpoints, which are what the persistent kernel solves once), as the
upper bound of what a certificate "V_j = V_i + g_i.(v_j - v_i)" could save.
"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools

af = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
er = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
mpc = examples.linear_mpc(0)
gp = engine.GpuProblem(mpc.compile(), 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
eps_a = float(np.max(gp.solve_pt(af * V)[0]))
gp.set_eps(eps_a, er)
roots, _ = ehm_tools.delaunay_roots(V)
flat = gp.partition(roots, max_nodes=1 << 22)
gp.close()
internal = np.nonzero(flat.left >= 0)[0]
L, Rr = flat.left[internal], flat.right[internal]
n = len(internal)
ar = np.arange(n)
i_new = np.argmax(np.any(flat.vertices[L] != flat.vertices[internal], axis=2), axis=1)
j_new = np.argmax(np.any(flat.vertices[Rr] != flat.vertices[internal], axis=2), axis=1)
Vp = flat.vertex_costs[internal]
Up = flat.vertex_inputs[internal]
Vbar_mid = 0.5 * (Vp[ar, i_new] + Vp[ar, j_new])
Ubar_mid = 0.5 * (Up[ar, i_new] + Up[ar, j_new])
Vmid = flat.vertex_costs[L][ar, i_new]
Umid = flat.vertex_inputs[L][ar, i_new]
mid = flat.vertices[L][ar, i_new]
gap = (Vbar_mid - Vmid) / (1. + np.abs(Vmid))
du = np.max(np.abs(Ubar_mid - Umid), axis=1)
_, first = np.unique(mid.view([('', mid.dtype)] * mid.shape[1]), return_index=True)
print('%d splits, %d distinct midpoints' % (n, len(first)))
for tol in (1e-12, 1e-10, 1e-9, 1e-8, 1e-6):
    a = gap <= tol
    print('  relative cost gap <= %.0e: %5.1f %% of the splits, %5.1f %% of the distinct midpoints; '
          'among those max |u_mid - mean u| = %.2e (median %.2e)' %
          (tol, 100 * a.mean(), 100 * a[first].mean(), du[a].max() if a.any() else 0.,
           np.median(du[a]) if a.any() else 0.))
print('  negative gaps (convexity violated by solver tolerance): min %.2e' % gap.min())
