"""How often would the midpoint solve alone prove a node open?  (offline, from an exported tree)

A node stays open iff some point of it has Vbar - V* >= max(eps_a, eps_r V*).  The split's
midpoint is such a candidate point, and its optimal cost is computed anyway when the node is
split: Vbar(mid) = (V_i + V_j)/2, V*(mid) = the children's new vertex cost.
"""
import sys
import numpy as np
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools

af = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
er = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
quad = len(sys.argv) > 3 and sys.argv[3] == 'qp'
mpc = examples.linear_mpc(0, cost='quadratic' if quad else 'inf')
gp = engine.GpuProblem(mpc.compile(), 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
eps_a = float(np.max(gp.solve_pt(af * V)[0]))
gp.set_eps(eps_a, er)
roots, _ = ehm_tools.delaunay_roots(V)
flat = gp.partition(roots, max_nodes=1 << 22)
gp.close()
internal = np.nonzero(flat.left >= 0)[0]
L = flat.left[internal]
# the new vertex of the left child = the slot where its vertices differ from the parent's
diff = np.any(flat.vertices[L] != flat.vertices[internal], axis=2)          # (n, p+1)
i_new = np.argmax(diff, axis=1)
Rr = flat.right[internal]
diff_r = np.any(flat.vertices[Rr] != flat.vertices[internal], axis=2)
j_new = np.argmax(diff_r, axis=1)
n = len(internal)
Vp = flat.vertex_costs[internal]
Vbar_mid = 0.5 * (Vp[np.arange(n), i_new] + Vp[np.arange(n), j_new])
Vmid = flat.vertex_costs[L][np.arange(n), i_new]
t_mid = Vbar_mid - Vmid - np.maximum(eps_a, er * Vmid)
ok = t_mid >= 0
print('internal nodes', n, 'closed leaves', int(np.sum(flat.flags & 1 > 0)),
      'midpoint certificate succeeds for %.1f %% of the open nodes' % (100 * ok.mean()))
dep = np.zeros(flat.n_nodes, dtype=int)
for k in internal:
    dep[flat.left[k]] = dep[flat.right[k]] = dep[k] + 1
for d in range(0, dep.max() + 1, 3):
    m = dep[internal] == d
    if m.any():
        print('  depth %2d: %7d open nodes, certificate %.1f %%' % (d, m.sum(), 100 * ok[m].mean()))
