"""Exploration: regional partitions at config-5 shape for a few regions / tolerances."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from explicit_hybrid_mpc_amd import examples, engine, sequences
mpc = examples.pwa4_mpc(N=8)
half = examples.theta_box(mpc)
E = np.vstack([np.zeros(8), np.eye(8)]) - 1. / 9
V = examples.box_vertices(half)
table = sequences.PrefixTable(mpc, slots=1024)
for frac, size in ((0.9, 0.04), (0.9, 0.02)):
    R = frac * V[37] + size * half * E
    seqs, info = sequences.relevant_sequences(mpc, R[None], table=table)
    can = mpc.restrict(seqs).compile()
    gp = engine.GpuProblem(can, 1., 1.)
    J = gp.solve_pt(R)[0]
    print(frac, size, len(seqs), 'J', J.min(), J.max(), flush=True)
    for eps in ((1e-3, 2e-3), (2e-4, 5e-4), (5e-5, 1e-4)):
        gp.set_eps(eps[0] * float(J.max()), eps[1])
        t = time.time()
        try:
            flat = gp.partition(R[None], action='ecc', max_nodes=1 << 19, max_depth=40)
            print(' ', eps, 'nodes', flat.n_nodes, 'leaves', flat.info['n_leaves'], 'depth', flat.info['max_depth'],
              'trunc', flat.info['truncated'], 'lp', flat.info['lp_solves'], 'used', len(set(flat.delta_idx[flat.delta_idx>=0].tolist())),
              'margin %.2e' % flat.info['min_margin'], 'dev %.3fs' % flat.info['device_seconds'], '%.2fs' % (time.time() - t), flush=True)
        except Exception as e:
            print(' ', eps, 'failed', e, flush=True)
    gp.close()
table.close()
