"""
configs[4]: catches the suboptimality-test problems of FULL sequences that stall on the device
(sequences.PrefixTable._settle_stalled raises for them), writes them to an npz (prefix, simplex,
vertex costs, what the device returned) and prints what the three problem kinds say about them.

    python tools/debug_stalled_c5.py --cells 3,4,5 --out gpurun_out/r5/stalled.npz
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cells', default='3')
    ap.add_argument('--out', default='gpurun_out/r5/stalled.npz')
    args = ap.parse_args()
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier, examples
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from explicit_hybrid_mpc_amd.tree import NodeData, Tree
    mpc = examples.pwa4_mpc(N=8, seed=0)
    V = examples.box_vertices(examples.theta_box(mpc))
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=8192, device=0)
    eps_a = float(np.max([j for _, _, j in bnb_frontier.p_theta_many(orc, 0.2 * V)]))
    orc.eps_a, orc.eps_r = eps_a, 1e-3
    orc.table.set_eps(eps_a, 1e-3)
    roots, _ = ehm_tools.delaunay_roots(V)
    caught = []
    for name in ('short', 'long'):
        tb = getattr(orc.table, name)
        real = tb.gp.simplex_idx

        def spy(R, slot, mode, Vbar=None, _real=real, _tb=tb, _name=name):
            obj, alpha, st = _real(R, slot, mode, Vbar)
            bad = np.flatnonzero(st != 0)
            if bad.size and mode == 1:
                inv = {s: q for q, s in _tb._slot_of.items()}
                for b in bad:
                    caught.append(dict(table=_name, prefix=inv[int(np.asarray(slot)[b])],
                                       R=np.asarray(R)[b].copy(), V=np.asarray(Vbar)[b].copy(),
                                       t=float(obj[b]), status=int(st[b])))
            return obj, alpha, st
        tb.gp.simplex_idx = spy
    for c in [int(x) for x in args.cells.split(',')]:
        orc.table.forget()
        tree = Tree(NodeData(vertices=roots[c].copy()))
        try:
            st = bnb_frontier.grow_frontier(orc, tree, 'ecc', order='lcss-first',
                                            table_backoff=True, round_cap=4096)
            print('cell %d: finished, %d regions' % (c, st['regions']))
        except Exception as e:
            print('cell %d: %s: %s' % (c, type(e).__name__, e))
    print('%d stalled slack problems caught' % len(caught))
    full = [x for x in caught if len(x['prefix']) == mpc.N]
    print('%d of them full sequences' % len(full))
    for x in caught[:40]:
        tb = getattr(orc.table, x['table'])
        slot = tb._ensure([x['prefix']])
        res = {}
        for mode, nm in ((2, 'phase_one'), (0, 'min'), (1, 'slack')):
            o, _, s = type(tb.gp).simplex_idx(tb.gp, x['R'][None], slot, mode, x['V'][None])
            res[nm] = (float(o[0]), int(s[0]))
        edge = max(np.linalg.norm(x['R'][i] - x['R'][j]) for i in range(9) for j in range(i))
        print(len(x['prefix']), x['prefix'], 't=%.6g st=%d' % (x['t'], x['status']), res,
              'longest edge %.3g' % edge, 'V', np.array2string(x['V'], precision=4))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez(args.out, eps_a=eps_a, n=len(caught),
             prefix=np.array([list(x['prefix']) + [-1] * (mpc.N - len(x['prefix'])) for x in caught]
                             or np.zeros((0, mpc.N))),
             R=np.array([x['R'] for x in caught] or np.zeros((0, 9, 8))),
             V=np.array([x['V'] for x in caught] or np.zeros((0, 9))),
             t=np.array([x['t'] for x in caught]), status=np.array([x['status'] for x in caught]))
    orc.close()


if __name__ == '__main__':
    main()
