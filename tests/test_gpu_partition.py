"""
GPU partition engine vs the CPU restatement of lib/worker.py:241-417: identical region
tree (same node set, bit-identical vertices, same closed leaves), vertex costs within
1e-7 relative, volume closure, and no decision closer than 1e-6 to its threshold.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

RTOL = 1e-7


def compare_trees(flat, nodes_cpu, locs, inputs_tol=None):
    """``inputs_tol``: also compare the vertex INPUTS -- what the explicit law interpolates
    (lib/mpc_library.py:786-789).  The first input of an LP optimum is a function of the
    parameter only where the optimal face is a point in u_0: so it is on the random dense
    instances ('lin': checked on the CPU over the bench instance, face width <= 1e-6 at every
    sampled parameter, HiGHS' vertex and the interior-point limit agree to 1e-10).  On the
    double integrator the infinity-norm cost leaves u_0 free on a face, so the RAW inputs of two
    solvers differ there; its inputs are compared after the lexicographic second stage
    (tests/test_gpu_lexicographic.py: every vertex of every node, device against oracle)."""
    loc = flat.locations(locs)
    assert len(loc) == len(nodes_cpu)
    assert set(loc) == set(nodes_cpu.keys())
    for k, name in enumerate(loc):
        ref = nodes_cpu[name]
        assert np.array_equal(flat.vertices[k], ref['vertices']), name
        assert flat.is_leaf(k) == ref['leaf'], name
        assert bool(flat.flags[k] & 1) == ref['is_epsilon_suboptimal'], name
        assert np.allclose(flat.vertex_costs[k], ref['vertex_costs'],
                           rtol=RTOL, atol=RTOL), name
        if inputs_tol is not None and ref['vertex_inputs'] is not None:
            assert np.allclose(flat.vertex_inputs[k], ref['vertex_inputs'], rtol=0.,
                               atol=inputs_tol), name


@pytest.mark.parametrize('decide_full', [0, 1])
@pytest.mark.parametrize('kind,seed,abs_frac,eps_r', [('di', 0, 0.25, 0.1),
                                                      ('lin', 0, 0.5, 1.0),
                                                      ('lin', 1, 0.5, 0.5)])
def test_tree_identical_to_cpu_partition(kind, seed, abs_frac, eps_r, decide_full):
    from explicit_hybrid_mpc_amd import engine, examples
    from oracle.oracle_cpu import OracleCPU
    from oracle.partition_cpu import PartitionCPU
    mpc = helpers.make_instance(kind, seed)
    eps_a = helpers.eps_a_rule(mpc, abs_frac)
    roots, locs = helpers.roots_of(mpc)
    orc = OracleCPU(mpc, eps_a, eps_r)
    cpu = PartitionCPU(orc)
    cpu.run(roots, locs, 'ecc')
    gp = engine.GpuProblem(mpc.compile(), eps_a, eps_r)
    gp.set_option('decide_full', decide_full)
    total = np.prod(2 * examples.theta_box(mpc))
    flats = []
    for eng in (1, 0):      # persistent frontier kernel / level-synchronous sweeps
        flat = gp.partition(np.array(roots), action='ecc', engine=eng)
        compare_trees(flat, cpu.nodes, locs, inputs_tol=1e-6 if kind == 'lin' else None)
        assert abs(flat.info['volume_closed'] - total) <= 1e-9 * total
        assert flat.info['min_margin'] > 1e-6
        assert min(cpu.min_margin, flat.info['min_margin']) > 1e-6
        flats.append(flat)
    gp.close()
    # the export of the persistent engine is relabelled to the sweeps' breadth-first numbering
    assert np.array_equal(flats[0].left, flats[1].left)
    assert np.array_equal(flats[0].vertices, flats[1].vertices)
    assert flats[0].info['sweeps'] == flats[1].info['sweeps']
    assert flats[0].info['decide_launches'] == 1 < flats[1].info['decide_launches']


def test_status_publisher_follows_the_run(tmp_path):
    """
    SURVEY section 8 f4: status.txt / statistics.pkl in the reference's formats, fed from the
    device's progress counters between sweeps; the tree is the one a plain run grows.
    """
    from explicit_hybrid_mpc_amd import engine, examples, status
    from oracle import geometry
    mpc = helpers.make_instance('lin', 0)
    eps_a = helpers.eps_a_rule(mpc, 0.25)
    roots, locs = helpers.roots_of(mpc)
    roots = np.array(roots)
    total = sum(geometry.simplex_volume(R) for R in roots)
    gp = engine.GpuProblem(mpc.compile(), eps_a, 0.5)
    plain = gp.partition(roots, action='ecc')
    st, pk = str(tmp_path / 'status.txt'), str(tmp_path / 'statistics.pkl')
    pub = status.MainStatusPublisher(total, st, pk)
    flat = gp.partition(roots, action='ecc', status=pub, status_sweeps=2)
    gp.close()
    assert np.array_equal(flat.vertices, plain.vertices) and np.array_equal(flat.left, plain.left)
    assert np.array_equal(flat.flags & 1, plain.flags & 1)
    stats = status.load_statistics(pk)
    frac = stats['volume_filled_frac']
    assert len(frac) >= 3 and all(b >= a for a, b in zip(frac, frac[1:]))
    assert abs(frac[-1] - 1.) < 1e-9
    n_splits = (flat.n_nodes - len(roots)) // 2     # the reference counts +1 per split
    assert stats['simplex_count_total'][-1] == n_splits
    assert stats['num_proc_active'][-1] == 0 and max(stats['num_proc_active']) == 1
    text = open(st).read()
    assert 'volume filled (total [%]): 1.0000e+02' in text
    assert 'simplex_count: %d' % n_splits in text
