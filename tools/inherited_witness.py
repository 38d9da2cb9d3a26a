"""How many suboptimality-test LPs would an INHERITED witness save?  (offline, bench tree)

A node the suboptimality-test LP finds open comes with the LP's maximiser theta_w and the optimal
cost there, c_w = V*(theta_w).  theta_w lies in one of the two children; there the interpolated
vertex cost Vbar_child(theta_w) is known from the child's record, and
    min(Vbar_child - c_w - eps_a, Vbar_child - (1 + eps_r) c_w) > 0
proves the child open without its own LP (it still needs its midpoint solve to be split).  The
witness travels on down while it keeps proving nodes open.  This script replays the bench
partition's decisions with that rule (converged solves: an upper estimate of what the sign-only
iterates of the engine would give) and counts the LPs.
"""
import sys
import numpy as np
sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import engine, examples
from explicit_hybrid_mpc_amd import tools as ehm_tools

af = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
er = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
tol = 1e-6
mpc = examples.linear_mpc(0)
can = mpc.compile()
gp = engine.GpuProblem(can, 1., 1.)
V = examples.box_vertices(examples.theta_box(mpc))
eps_a = float(np.max(gp.solve_pt(af * V)[0]))
gp.set_eps(eps_a, er)
roots, _ = ehm_tools.delaunay_roots(V)
flat = gp.partition(roots, max_nodes=1 << 22)
N, p = flat.n_nodes, can.p
left, right = flat.left, flat.right
internal = np.nonzero(left >= 0)[0]
n = len(internal)
L, Rr = left[internal], right[internal]
ar = np.arange(n)
i_new = np.argmax(np.any(flat.vertices[L] != flat.vertices[internal], axis=2), axis=1)
j_new = np.argmax(np.any(flat.vertices[Rr] != flat.vertices[internal], axis=2), axis=1)
Vp = flat.vertex_costs[internal]
Vmid = flat.vertex_costs[L][ar, i_new]
vb = 0.5 * (Vp[ar, i_new] + Vp[ar, j_new])
t_mid = np.minimum(vb - Vmid - eps_a, vb - (1 + er) * Vmid)
mid_ok = np.zeros(N, dtype=bool)
mid_ok[internal] = t_mid > tol * (1 + np.abs(vb))
closed = (flat.flags & 1) > 0
# the LP's maximiser and the optimal cost there, for every open node the midpoint does not prove
need = internal[~mid_ok[internal]]
t, alpha, st = gp.slack(flat.vertices[need], flat.vertex_costs[need])
theta_w = np.einsum('ni,nij->nj', alpha, flat.vertices[need])
c_w = gp.solve_ptd(theta_w)[0]
gp.close()
own = np.full(N, -1)
own[need] = np.arange(len(need))
print('nodes %d, internal %d, closed leaves %d; open nodes proved by their midpoint %d; '
      'open nodes that need the LP today %d (t* min %.3g)' %
      (N, n, int(closed.sum()), int(mid_ok.sum()), len(need), t.min()))
# replay, parents before children (export order)
wit_theta = np.zeros((N, p))
wit_c = np.zeros(N)
has = np.zeros(N, dtype=bool)
saved = 0
projected_given = 0
PROJECT = len(sys.argv) > 3 and sys.argv[3] == 'project'
split_ij = {}
for q, kk in enumerate(internal):
    split_ij[kk] = (int(i_new[q]), int(j_new[q]))
chain = np.zeros(N, dtype=int)
lp_open = 0
for k in range(N):
    if left[k] < 0:
        continue
    proved = False
    if has[k]:
        # barycentric coordinates of the witness in this node
        Rk = flat.vertices[k]
        A = np.vstack([Rk.T, np.ones(p + 1)])
        a = np.linalg.solve(A, np.append(wit_theta[k], 1.))
        if a.min() >= -1e-9:
            vbw = a @ flat.vertex_costs[k]
            tw = min(vbw - wit_c[k] - eps_a, vbw - (1 + er) * wit_c[k])
            proved = tw > tol * (1 + abs(vbw))
    if proved:
        if not mid_ok[k]:
            saved += 1
        th, c = wit_theta[k], wit_c[k]
        ch = chain[k] + 1
    elif mid_ok[k]:
        continue
    else:
        lp_open += 1
        q = own[k]
        th, c = theta_w[q], c_w[q]
        ch = 0
    # hand the witness to the child that contains it
    holder = None
    for child in (left[k], right[k]):
        Rc = flat.vertices[child]
        A = np.vstack([Rc.T, np.ones(p + 1)])
        try:
            a = np.linalg.solve(A, np.append(th, 1.))
        except np.linalg.LinAlgError:
            continue
        if a.min() >= -1e-9:
            has[child], wit_theta[child], wit_c[child], chain[child] = True, th, c, ch
            holder = child
            break
    if PROJECT and holder is not None:
        # the other child: slide the witness towards the parent's vertex on its side until the
        # shared face; (1 - mu) c_w + mu V_v bounds the optimal cost there (convex combination
        # of feasible decision vectors)
        other = right[k] if holder == left[k] else left[k]
        Rk = flat.vertices[k]
        a = np.linalg.solve(np.vstack([Rk.T, np.ones(p + 1)]), np.append(th, 1.))
        kk = int(np.nonzero(internal == k)[0][0]) if False else None
        i_, j_ = split_ij[k]
        # holder keeps v_j (child 0, alpha_j >= alpha_i) or v_i (child 1)
        far, near = (i_, j_) if a[j_] >= a[i_] else (j_, i_)
        mu = (a[near] - a[far]) / (1. + a[near] - a[far])
        th2 = (1 - mu) * th + mu * Rk[far]
        c2 = (1 - mu) * c + mu * flat.vertex_costs[k][far]
        has[other], wit_theta[other], wit_c[other], chain[other] = True, th2, c2, ch
        projected_given += 1
print('suboptimality-test LPs on open nodes: today %d, with inherited witnesses%s %d (%.1f %% fewer); '
      'longest chain %d' % (len(need), ' + face projection' if PROJECT else '', lp_open,
                            100. * saved / max(1, len(need)), chain.max()))
