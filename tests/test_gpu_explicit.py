"""
Batched explicit-MPC evaluation (ehm_explicit_*) against the CPU restatement of the reference's
ExplicitMPC (oracle/explicit_cpu.py, lib/mpc_library.py:685-792) on a partition grown by the
engine: same containing leaf, same interpolated input.  Both tree layouts are exercised -- the
reference's nested right-spine tree and the engine's flat forest.
"""

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def test_explicit_evaluation_matches_reference_walk():
    from explicit_hybrid_mpc_amd import examples, explicit, partition
    from oracle.explicit_cpu import ExplicitCPU
    orc = examples.create_oracle(helpers.make_instance('lin', 0),
                                 examples.box_vertices(examples.theta_box(helpers.make_instance('lin', 0))),
                                 abs_frac=0.3, abs_err=None, rel_err=0.5)
    V = examples.box_vertices(examples.theta_box(orc.mpc))
    root, flat = partition.partition_set(orc, V)
    assert flat.n_nodes > 500
    cpu = ExplicitCPU(root)
    rng = np.random.default_rng(5)
    half = examples.theta_box(orc.mpc)
    X = rng.uniform(-1, 1, (4000, half.size)) * half
    # states hugging the corners / faces of the set (1e-7 inside: exactly ON a face the
    # reference's own eps-test walk is ill-defined and can end in a leaf that does not contain
    # the state)
    X[:16] = V[:16] * (1 - 1e-7)
    X[16:32] = 0.5 * (V[:16] + V[rng.integers(16, size=16)]) * (1 - 1e-7)
    nested = explicit.ExplicitMPC(root, orc)
    forest = explicit.ExplicitMPC(flat, orc)
    u_n, leaf_n, vis_n, _ = nested.evaluate(X, return_info=True)
    u_f, leaf_f, vis_f, _ = forest.evaluate(X, return_info=True)
    u_ref = np.array([cpu(x) for x in X[:1500]])
    cells = [cpu.get_containing_cell(x) for x in X[:1500]]
    same = 0
    for k in range(1500):
        got = nested.nodes[leaf_n[k]].data
        if got is cells[k]:
            same += 1
            assert np.allclose(u_n[k], u_ref[k], rtol=1e-10, atol=1e-12)
        else:
            # a state within rounding of a shared face may be given to the neighbour, which
            # must then contain it too
            E = np.column_stack([v - got.vertices[0] for v in got.vertices[1:]])
            a = np.linalg.solve(E, X[k] - got.vertices[0])
            assert min(a.min(), 1 - a.sum()) > -1e-9
    assert same >= 1490
    # the two layouts walk the same tree
    assert np.allclose(u_n, u_f, rtol=1e-12, atol=1e-13)
    for k in range(0, 4000, 97):
        assert np.array_equal(nested.nodes[leaf_n[k]].data.vertices, flat.vertices[leaf_f[k]])
    # single-state call signature of the reference
    u1, t1 = nested(X[40])
    assert np.allclose(u1, u_n[40]) and t1 >= 0.
    assert nested.get_containing_cell(X[40]) is nested.nodes[leaf_n[40]].data
    nested.close()
    forest.close()
