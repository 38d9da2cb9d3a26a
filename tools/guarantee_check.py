"""Epsilon-suboptimality guarantee on a finished device partition (see tests/test_gpu_explicit.py)."""
import sys, time, json
import numpy as np
from explicit_hybrid_mpc_amd import examples, explicit, partition
from explicit_hybrid_mpc_amd import tools as ehm_tools

af, er, n_s = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
full_set, _, orc = examples.example('cwh_z', abs_frac=af, rel_err=er)
roots, _ = ehm_tools.delaunay_roots(full_set)
t0 = time.time()
flat = partition.run_engine(orc, roots, action='ecc', max_nodes=1 << 22)
t1 = time.time()
law = explicit.ExplicitMPC(flat, orc)
rng = np.random.default_rng(1)
half = np.abs(full_set).max(axis=0)
X = rng.uniform(-1, 1, (n_s, 2)) * half * (1 - 1e-9)
u, leaf, visits, _ = law.evaluate(X, return_info=True)
R = flat.vertices[leaf]
E = np.transpose(R[:, 1:] - R[:, :1], (0, 2, 1))
beta = np.linalg.solve(E, (X - R[:, 0])[:, :, None])[:, :, 0]
alpha = np.concatenate([1. - beta.sum(axis=1, keepdims=True), beta], axis=1)
Vbar = np.sum(alpha * flat.vertex_costs[leaf], axis=1)
Vstar, _, didx = orc.gpu.solve_pt(X)
gap = Vbar - Vstar
bound = np.maximum(orc.eps_a, orc.eps_r * Vstar)
print(json.dumps(dict(abs_frac=af, rel_err=er, nodes=flat.n_nodes, partition_seconds=t1 - t0,
                      samples=n_s, all_closed=bool((flat.flags[leaf] & 1).all()),
                      alpha_min=float(alpha.min()), gap_min=float(gap.min()),
                      worst_ratio=float((gap / bound).max()), violations=int((gap > bound * (1 + 1e-6) + 1e-12).sum()),
                      negative=int((gap < -1e-9).sum()), slivers=int(orc.gpu.stats()['slivers']))))
