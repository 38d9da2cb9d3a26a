"""GPU sanity run of the quadratic-cost oracles on the reference's cwh_z law (gpurun)."""
import sys, time
import numpy as np
from explicit_hybrid_mpc_amd import examples, engine, partition
from explicit_hybrid_mpc_amd.oracle import Oracle
from oracle.satellite_cpu import SatelliteZCPU, KNOWN_EPS_A
from oracle.oracle_cpu import OracleCPU
from oracle.partition_cpu import PartitionCPU
from oracle import geometry

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mpc = examples.satellite_z(N)
V = examples.box_vertices(examples.theta_box(mpc))
for af in (0.5, 0.25):
    orc = examples.create_oracle(mpc, V, af, None, 2.0)
    print('eps_a', af, orc.eps_a, KNOWN_EPS_A.get((N, af)), flush=True)
    orc.close()
cpu_mpc = SatelliteZCPU(N)
eps_a, eps_r = 0.012, 1.0
gpu = Oracle(mpc, eps_a, eps_r)
cpu = OracleCPU(cpu_mpc, eps_a, eps_r)
cpu.memoize = True
rng = np.random.default_rng(3)
half = examples.theta_box(mpc)
worst = dict(pt=0., slack=0., mins=0.)
nd = len(cpu.deltas)
for k in range(12):
    th = rng.uniform(-0.8, 0.8, 2) * half
    u, d, J, _ = gpu.P_theta(th)
    ur, dr, Jr, _ = cpu.P_theta(th)
    worst['pt'] = max(worst['pt'], abs(J - Jr) / (1 + abs(Jr)))
    assert np.array_equal(d.astype(int), dr.astype(int)), (d, dr)
    assert abs(u[0] - ur[0]) < 1e-7, (u, ur)
print('P_theta', worst['pt'], flush=True)
cnt = 0
for k in range(40):
    ctr = rng.uniform(-0.7, 0.7, 2) * half
    R = ctr + rng.uniform(-1, 1, (3, 2)) * half * 10 ** rng.uniform(-2, -0.5)
    d, vx = cpu.V_R(R)
    dg, vxg = gpu.V_R(R)
    assert (d is None) == (dg is None)
    if d is None:
        continue
    assert np.array_equal(d.astype(int), dg.astype(int))
    Vb = np.array([v[1] for v in vx])
    Vg = np.array([v[1] for v in vxg])
    assert np.allclose(Vb, Vg, rtol=1e-7, atol=1e-7), (Vb, Vg)
    cnt += 1
    for dd in (cpu.delta_index(d), int(rng.integers(nd))):
        t_c, _ = cpu.slack(R, Vb, dd)
        t_g, al, st = gpu.gpu.slack(R[None], Vb[None], cpu.deltas[dd])
        if np.isfinite(t_c):
            worst['slack'] = max(worst['slack'], abs(t_g[0] - t_c) / (1 + abs(t_c)))
            assert st[0] == 0
    assert gpu.bar_E_delta_R(R, Vb) == cpu.bar_E_delta_R(R, Vb)
    a = cpu.bar_D_delta_R(R, Vb, d)
    b = gpu.bar_D_delta_R(R, Vb, d)
    assert (a[0] is None) == (b[0] is None), (a[0], b[0])
    if a[0] is not None:
        assert np.array_equal(a[0].astype(int), b[0].astype(int))
        assert a[3] == b[3]
print('simplices', cnt, worst, flush=True)
gpu.close()

# whole partition of the reference's example (hybrid driver) vs the CPU restatement
af, er = (0.5, 2.0)
orc = examples.create_oracle(mpc, V, af, None, er)
t0 = time.time()
roots, locs = geometry.delaunay_simplices(V)
flat = partition.run_engine(orc, np.array(roots), action='ecc')
print('gpu partition', flat.n_nodes, 'nodes', int(sum(flat.is_leaf(k) for k in range(flat.n_nodes))),
      'leaves', time.time() - t0, 's', {k: flat.info[k] for k in ('lp_solves', 'min_margin', 'sweeps') if k in flat.info}, flush=True)
cpu = OracleCPU(cpu_mpc, orc.eps_a, er)
cpu.memoize = True
pc = PartitionCPU(cpu)
t0 = time.time()
pc.run(roots, locs, 'ecc')
print('cpu partition', len(pc.nodes), len(pc.leaves()), time.time() - t0, 's', flush=True)
loc = flat.locations(locs)
assert set(loc) == set(pc.nodes.keys()), (len(loc), len(pc.nodes))
for k, name in enumerate(loc):
    ref = pc.nodes[name]
    assert np.array_equal(flat.vertices[k], ref['vertices'])
    assert flat.is_leaf(k) == ref['leaf']
    assert np.allclose(flat.vertex_costs[k], ref['vertex_costs'], rtol=1e-7, atol=1e-7)
print('TREE IDENTICAL')
