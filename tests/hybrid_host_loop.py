"""
TEST INFRASTRUCTURE: the round-1 hybrid driver -- ``Worker.ecc`` / ``Worker.lcss`` as a Python
loop over the BATCHED device oracles (ehm_lcss_batch, ehm_vr_batch, ...).  The product path is
the device engine behind ``ehm_partition_run`` (csrc/ehm_hybrid.h); this loop stays as an
independent route through the Level-2 oracle entry points that must grow the same tree.
"""

import numpy as np

from explicit_hybrid_mpc_amd import engine
from explicit_hybrid_mpc_amd.engine import FlatTree


class _Pool:
    """Growable struct-of-arrays node pool (host side of the hybrid driver)."""

    def __init__(self, p, n_u):
        self.p, self.n_u = p, n_u
        self.vertices, self.left, self.right = [], [], []
        self.didx, self.vcost, self.vinput, self.flags, self.tstar = [], [], [], [], []

    def add(self, R, didx=-1, vcost=None, vinput=None):
        p, n_u = self.p, self.n_u
        self.vertices.append(np.array(R, dtype=np.float64))
        self.left.append(-1)
        self.right.append(-1)
        self.didx.append(int(didx))
        self.vcost.append(np.zeros(p + 1) if vcost is None else np.array(vcost))
        self.vinput.append(np.zeros((p + 1, n_u)) if vinput is None else np.array(vinput))
        self.flags.append(2 if didx >= 0 else 0)
        self.tstar.append(0.)
        return len(self.left) - 1

    def flat(self, info, deltas):
        return FlatTree(np.array(self.vertices), np.array(self.left, dtype=np.int32),
                        np.array(self.right, dtype=np.int32),
                        np.array(self.didx, dtype=np.int32), np.array(self.vcost),
                        np.array(self.vinput), np.array(self.flags, dtype=np.uint8),
                        np.array(self.tstar), info, deltas)


def grow_hybrid_host(gp, roots, action='ecc', init=None, max_nodes=1 << 20):
    """
    Level-synchronous ``ecc`` / ``lcss`` for any number of commutations, every oracle
    evaluated as a batched GPU call.  Returns a FlatTree.
    """
    can = gp.can
    p, n_u = can.p, can.n_u
    pool = _Pool(p, n_u)
    roots = np.asarray(roots, dtype=np.float64).reshape(-1, p + 1, p)
    ecc_front, lcss_front = [], []
    for k in range(roots.shape[0]):
        if action == 'ecc':
            ecc_front.append(pool.add(roots[k]))
        else:
            d = can.delta_index(init['delta'][k])
            lcss_front.append(pool.add(roots[k], d, init['vertex_costs'][k],
                                       init['vertex_inputs'][k]))
    stats0 = gp.stats()
    sweeps = 0
    min_margin = np.inf
    ref_solves = 0
    # what is known about the live lcss nodes (dropped when a node is closed or split):
    # vfeas[i] (p+1, n_delta) feasibility of every commutation at every vertex; cand[i]
    # (n_delta,) commutations feasible somewhere on the PARENT simplex
    vfeas, cand = {}, {}
    all_true = np.ones(can.n_delta, dtype=bool)

    def split(ids):
        R = np.array([pool.vertices[i] for i in ids])
        S1, S2, ij = engine.split_batch(R, device=gp.device)
        return S1, S2, ij

    while ecc_front or lcss_front:
        sweeps += 1
        if len(pool.left) > max_nodes:
            raise RuntimeError('node pool exhausted (max_nodes=%d)' % max_nodes)
        next_ecc, next_lcss = [], []
        if ecc_front:                                     # lib/worker.py:262-291
            R = np.array([pool.vertices[i] for i in ecc_front])
            _, _, d_c = gp.solve_pt(R.mean(axis=1))
            if (d_c < 0).any():
                raise RuntimeError('STOP, Theta contains infeasible regions')
            didx, vJ, vu = gp.v_r(R)
            ref_solves += 2 * len(ecc_front)
            to_split = [k for k in range(len(ecc_front)) if didx[k] < 0]
            for k, i in enumerate(ecc_front):
                if didx[k] >= 0:
                    pool.didx[i] = int(didx[k])
                    pool.vcost[i] = vJ[k].copy()
                    pool.vinput[i] = vu[k].copy()
                    pool.flags[i] |= 2
                    next_lcss.append(i)
                    ref_solves += p + 1
            if to_split:
                ids = [ecc_front[k] for k in to_split]
                S1, S2, _ = split(ids)
                for q, i in enumerate(ids):
                    a, b = pool.add(S1[q]), pool.add(S2[q])
                    pool.left[i], pool.right[i] = a, b
                    next_ecc += [a, b]
        if lcss_front:                                    # lib/worker.py:367-417
            ids = lcss_front
            R = np.array([pool.vertices[i] for i in ids])
            V = np.array([pool.vcost[i] for i in ids])
            # per-vertex feasibility of every commutation: inherited with the vertices; a node
            # that arrives from ecc (or as an lcss root) gets it here, once
            fresh = [i for i in ids if i not in vfeas]
            if fresh:
                Rf = np.array([pool.vertices[i] for i in fresh]).reshape(-1, p)
                mf = gp.feas_all(Rf).reshape(len(fresh), p + 1, can.n_delta)
                for q, i in enumerate(fresh):
                    vfeas[i] = mf[q]
            vf = np.array([vfeas[i] for i in ids])
            cd = np.array([cand.get(i, all_true) for i in ids])
            dref = can.deltas[[pool.didx[i] for i in ids]]
            closed, tbest, cand_out, dstar, ths, vJ, vu, vsmall = gp.lcss(R, V, dref, vf, cd)
            ref_solves += len(ids)
            min_margin = min(min_margin, float(np.min(np.abs(tbest))))
            split_ids, split_d, split_c, split_u = [], [], [], []
            for k, i in enumerate(ids):
                pool.tstar[i] = float(tbest[k])
                cand[i] = cand_out[k]
                if closed[k]:
                    pool.flags[i] |= 1
                    vfeas.pop(i, None)
                    cand.pop(i, None)
                    continue
                ref_solves += 1
                if dstar[k] >= 0:
                    ref_solves += p + 3
                    if vsmall[k]:                     # swap in place, revisit the node
                        pool.didx[i] = int(dstar[k])
                        pool.vcost[i] = vJ[k].copy()
                        pool.vinput[i] = vu[k].copy()
                        next_lcss.append(i)
                        continue
                    # split with the better commutation; the parent keeps its own data
                    split_d.append(int(dstar[k]))
                    split_c.append(vJ[k].copy())
                    split_u.append(vu[k].copy())
                else:                                 # lib/worker.py:381-386
                    split_d.append(pool.didx[i])
                    split_c.append(pool.vcost[i].copy())
                    split_u.append(pool.vinput[i].copy())
                split_ids.append(i)
            if split_ids:
                S1, S2, ij = split(split_ids)
                mids = np.array([S1[q][ij[q, 0]] for q in range(len(split_ids))])
                Jm, um, st, _ = gp.solve_ptd(mids, can.deltas[split_d])
                ref_solves += len(split_ids)
                if (st != 0).any():
                    raise RuntimeError('midpoint solve did not converge')
                mid_feas = gp.feas_all(mids)          # the one new vertex of the two children
                for q, i in enumerate(split_ids):
                    c1, c2 = split_c[q].copy(), split_c[q].copy()
                    u1, u2 = split_u[q].copy(), split_u[q].copy()
                    c1[ij[q, 0]] = Jm[q]
                    c2[ij[q, 1]] = Jm[q]
                    u1[ij[q, 0]] = um[q]
                    u2[ij[q, 1]] = um[q]
                    a = pool.add(S1[q], split_d[q], c1, u1)
                    b = pool.add(S2[q], split_d[q], c2, u2)
                    pool.left[i], pool.right[i] = a, b
                    f1, f2 = vfeas[i].copy(), vfeas[i].copy()
                    f1[ij[q, 0]] = mid_feas[q]
                    f2[ij[q, 1]] = mid_feas[q]
                    vfeas[a], vfeas[b] = f1, f2
                    cand[a] = cand[b] = cand[i]       # feasible on the parent simplex
                    vfeas.pop(i, None)
                    cand.pop(i, None)
                    next_lcss += [a, b]
        ecc_front, lcss_front = next_ecc, next_lcss
    stats1 = gp.stats()
    flags = np.array(pool.flags)
    left = np.array(pool.left)
    n_closed = int(np.sum(flags & 1 > 0))
    vol = 0.
    if n_closed:
        vol = float(np.sum(engine.volume_batch(
            np.array(pool.vertices)[(flags & 1) > 0], device=gp.device)))
    info = dict(n_nodes=len(pool.left), n_leaves=int(np.sum(left < 0)),
                n_roots=roots.shape[0], n_closed=n_closed,
                lp_solves=stats1['lp_solves'] - stats0['lp_solves'], ref_solves=ref_solves,
                ipm_iters=stats1['ipm_iters'] - stats0['ipm_iters'], sweeps=sweeps,
                max_depth=sweeps, truncated=0, volume_closed=vol, min_margin=min_margin,
                device_seconds=0.)
    return pool.flat(info, can.deltas)


