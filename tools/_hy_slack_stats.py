"""How many suboptimality-test problems would a bound-guided evaluation save?  For the nodes of a
cwh_z job: all slack values of a node and of its children (own-commutation splits), then count
what 'evaluate the top-K of the parent first, skip everything whose parent value is below the best
child value found' would solve."""
import sys, json
import numpy as np
sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import examples
from oracle import geometry
job = int(sys.argv[1]) if len(sys.argv) > 1 else 1
known = json.load(open('tests/golden/known_answers.json'))['runs']
fracs = [0.5, 0.25, 0.1, 0.03, 0.01]
r = known[job]
full_set, part, oracle = examples.example('cwh_z', abs_frac=fracs[job], rel_err=float(r['rel_err']))
roots, locs = geometry.delaunay_simplices(full_set)
g = oracle.gpu
can = g.can
flat = g.partition(np.array(roots), action='ecc', max_nodes=1 << 22)
nd = can.n_delta
# nodes that were split with their own commutation: children carry the same delta
ks = [k for k in range(flat.n_nodes) if flat.left[k] >= 0 and (flat.flags[k] & 2)
      and flat.delta_idx[flat.left[k]] == flat.delta_idx[k]]
rng = np.random.default_rng(0)
ks = rng.choice(ks, size=min(400, len(ks)), replace=False)
def slacks(k):
    R = np.repeat(flat.vertices[k][None], nd, axis=0)
    V = np.repeat(flat.vertex_costs[k][None], nd, axis=0)
    t, al, st = g.slack(R, V, can.deltas)
    t = np.where(st == 0, t, -np.inf)
    return t
tot_all = tot_pos = tot_guided = 0
nopen = 0
for k in ks:
    tp = slacks(k)
    for c in (flat.left[k], flat.right[k]):
        tc = slacks(c)
        feas = np.isfinite(tc)
        cand = feas & ~(tp < -1e-7)             # what the engine evaluates today (negatives inherited)
        tot_all += int(feas.sum()); tot_pos += int(cand.sum())
        # guided: order by parent value, evaluate top 4, then everything whose parent value >= best child value found - tie
        order = np.argsort(-np.where(cand, tp, -np.inf))
        order = [d for d in order if cand[d]]
        ev = set(order[:4])
        L = max([tc[d] for d in ev], default=-np.inf)
        if L >= 0:
            nopen += 1
            for d in order[4:]:
                if tp[d] >= L - 1e-6 * (1 + abs(L)):
                    ev.add(d)
        else:
            ev = set(order)                      # closed candidates: everything must be seen negative
        tot_guided += len(ev)
print(json.dumps(dict(job=job, nodes=len(ks) * 2, open=nopen, all_feasible=tot_all, evaluated_today=tot_pos, guided=tot_guided)))
oracle.close()
