"""
The native partition driver (include/ehm_frontier.h) on the device, configs[4] (n_x = 8, n_u = 3,
4 modes, N = 8: 65 536 sequences): one Delaunay root cell of the box grown by
``frontier.grow_cells`` (C++ round loop, native condensation, two device tables) and by
``bnb_frontier.grow_frontier`` (rounds 3-4) -- the same tree, cell for cell; the blocks the native
driver writes are the blocks ``PWAMPC.condense_prefix`` states.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nodes(t):
    return {loc: nd for nd, loc in t.walk()}


def test_native_driver_grows_the_tree_of_the_python_driver_on_the_device():
    from explicit_hybrid_mpc_amd import bnb, bnb_frontier, examples, frontier
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from explicit_hybrid_mpc_amd.tree import NodeData, Tree
    mpc = examples.pwa4_mpc(N=8, seed=0)
    V = examples.box_vertices(examples.theta_box(mpc))
    orc = bnb.PrefixOracle(mpc, 1., 1., slots=4096)
    eps_a = float(np.max([j for _, _, j in bnb_frontier.p_theta_many(orc, 0.2 * V)]))
    eps_r = 1e-3
    orc.eps_a, orc.eps_r = eps_a, eps_r
    orc.table.set_eps(eps_a, eps_r)
    roots, _ = ehm_tools.delaunay_roots(V)
    assert len(roots) > 30000                       # Qhull's triangulation of the 8-cube
    # the eps_a rule (lib/examples.py:42-46: 2^p P_theta searches) on the native driver: the
    # optima and canonical sequences of bnb_frontier.p_theta_many
    ref_pt = bnb_frontier.p_theta_many(orc, 0.2 * V)
    nat0 = frontier.NativeFrontier(mpc, 1., 1., slots=4096)
    got_pt = nat0.p_theta(0.2 * V)
    nat0.close()
    for (u_r, d_r, J_r), (u_g, d_g, J_g) in zip(ref_pt, got_pt):
        assert d_g is not None and np.array_equal(d_g, d_r)
        assert abs(J_g - J_r) <= 1e-9 * (1. + abs(J_r)) and np.allclose(u_g, u_r, atol=1e-7)
    assert abs(max(j for _, _, j in got_pt) - eps_a) <= 1e-9
    R = roots[2]                                    # a small cell: ~1.5 k regions
    ref = Tree(NodeData(vertices=R.copy()))
    s_ref = bnb_frontier.grow_frontier(orc, ref, 'ecc', order='lcss-first', table_backoff=True)
    nat = frontier.NativeFrontier(mpc, eps_a, eps_r, slots=4096)
    got = Tree(NodeData(vertices=R.copy()))
    orc.table.forget()
    st = frontier.grow_cells(nat, got, slow_oracle=lambda: orc,
                             slow_opts=dict(order='lcss-first', table_backoff=True))
    a, b = _nodes(ref), _nodes(got)
    assert set(a) == set(b) and len(a) > 1000
    assert st['regions'] == s_ref['regions'] == sum(nd.is_leaf() for nd in a.values())
    for loc, x in a.items():
        y = b[loc]
        assert np.array_equal(x.data.vertices, y.data.vertices)
        assert x.is_leaf() == y.is_leaf()
        assert x.data.is_epsilon_suboptimal == y.data.is_epsilon_suboptimal
        assert hasattr(x.data, 'commutation') == hasattr(y.data, 'commutation')
        if hasattr(x.data, 'commutation') and x.is_leaf():
            assert np.array_equal(x.data.commutation, y.data.commutation)
            assert np.allclose(x.data.vertex_costs, y.data.vertex_costs, rtol=1e-9, atol=1e-9)
    # the native share of the work: the interpreter sees only the cells handed back open (this
    # small cell is the lcss-heaviest of the first roots: 6 % of its visits)
    assert st['slow_path_cells'] <= 0.1 * st['visits']
    assert st['seconds_solvers'] >= 0.5 * st['seconds_total']
    nat.close()
    orc.close()


def test_two_handles_grow_different_roots_at_the_same_time():
    """
    bench.py --host-streams: two driver handles (own device tables, streams and host threads) grow
    different Delaunay roots from two interpreter threads at the same time -- the native calls
    release the GIL.  Each tree must be the one the same handle grows alone.
    """
    import threading
    from explicit_hybrid_mpc_amd import examples, frontier
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from explicit_hybrid_mpc_amd.tree import NodeData, Tree
    mpc = examples.pwa4_mpc(N=8, seed=0)
    V = examples.box_vertices(examples.theta_box(mpc))
    nats = [frontier.NativeFrontier(mpc, 1., 1., slots=4096) for _ in range(2)]
    eps_a = max(j for _, _, j in nats[0].p_theta(0.2 * V))
    for nat in nats:
        nat.set_eps(eps_a, 1e-3)
    roots, _ = ehm_tools.delaunay_roots(V)
    cells = (2, 1)                      # 1.5 k and 22.6 k regions
    alone = []
    for nat, c in zip(nats, cells):
        t = Tree(NodeData(vertices=roots[c].copy()))
        st = frontier.grow_cells(nat, t)
        assert st['slow_path_cells'] == 0
        alone.append((t, st['regions']))
    both, errs = [None, None], []

    def work(k):
        try:
            t = Tree(NodeData(vertices=roots[cells[k]].copy()))
            st = frontier.grow_cells(nats[k], t)
            both[k] = (t, st['regions'])
        except BaseException as e:
            errs.append(e)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for (ta, ra), (tb, rb) in zip(alone, both):
        a, b = _nodes(ta), _nodes(tb)
        assert ra == rb and set(a) == set(b)
        for loc, x in a.items():
            y = b[loc]
            assert np.array_equal(x.data.vertices, y.data.vertices)
            assert x.is_leaf() == y.is_leaf()
            assert x.data.is_epsilon_suboptimal == y.data.is_epsilon_suboptimal
    for nat in nats:
        nat.close()


def test_cells_move_between_two_device_handles():
    """
    ehm_frontier_take / ehm_frontier_give on the device tables: a handle that grows a configs[4]
    Delaunay root is stopped after 3 000 visits, the shallowest half of its pending cells moves to
    a second handle (own tables, own stream), the two finish AT THE SAME TIME from two threads, and
    the merged tree (frontier.merge_taken) is the tree one handle grows alone.
    """
    import threading
    from explicit_hybrid_mpc_amd import examples, frontier
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    mpc = examples.pwa4_mpc(N=8, seed=0)
    V = examples.box_vertices(examples.theta_box(mpc))
    nats = [frontier.NativeFrontier(mpc, 1., 1., slots=4096) for _ in range(2)]
    eps_a = max(j for _, _, j in nats[0].p_theta(0.2 * V))
    for nat in nats:
        nat.set_eps(eps_a, 1e-3)
    roots, _ = ehm_tools.delaunay_roots(V)
    a, b = nats
    a.add_roots(roots[1:2])                      # 22.6 k regions
    st0 = a.run()
    ref = a.export()
    a.reset()
    a.add_roots(roots[1:2])
    st = a.run(max_visits=3000)
    assert st['truncated'] and a.pending() >= 4
    cells = a.take(a.pending() // 2)
    assert len(cells['node']) >= 2 and np.all(np.diff(cells['depth']) >= 0)
    b.give(cells)
    out, errs = {}, []

    def work(name, nat):
        try:
            out[name] = nat.run()
        except BaseException as e:
            errs.append(e)
    threads = [threading.Thread(target=work, args=(n, h)) for n, h in (('a', a), ('b', b))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    assert out['a']['regions'] + out['b']['regions'] == st0['regions']
    assert out['b']['regions'] > 0 and out['a']['open_cells'] == out['b']['open_cells'] == 0
    merged = frontier.merge_taken(a.export(), cells, b.export())
    assert merged['n_nodes'] == ref['n_nodes']

    def by_location(flat):
        loc, stack = {}, [(0, '')]
        while stack:
            k, name = stack.pop()
            loc[name] = k
            if flat['left'][k] >= 0:
                stack.append((int(flat['left'][k]), name + '0'))
                stack.append((int(flat['right'][k]), name + '1'))
        return loc
    lr, lm = by_location(ref), by_location(merged)
    assert set(lr) == set(lm)
    for name, k in lr.items():
        j = lm[name]
        assert np.array_equal(ref['vertices'][k], merged['vertices'][j]), name
        assert ref['flags'][k] == merged['flags'][j], name
        assert np.array_equal(ref['sequence'][k], merged['sequence'][j]), name
        if ref['flags'][k] & frontier.FR_HAS_RECORD:
            assert np.allclose(ref['vertex_costs'][k], merged['vertex_costs'][j],
                               rtol=1e-7, atol=1e-7), name
    for nat in nats:
        nat.close()


def _rank_shares_a_device_root(rank, world, port, out_dir):
    import os
    import pickle
    import time
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from explicit_hybrid_mpc_amd import distributed, examples, frontier
    from explicit_hybrid_mpc_amd import tools as ehm_tools
    from explicit_hybrid_mpc_amd.tree import NodeData, Tree
    distributed.init_process_group('gloo')          # both ranks on the one GPU of the box
    mpc = examples.pwa4_mpc(N=8, seed=0)
    V = examples.box_vertices(examples.theta_box(mpc))
    nat = frontier.NativeFrontier(mpc, 1., 1., slots=4096)
    eps_a = max(j for _, _, j in nat.p_theta(0.2 * V))
    nat.set_eps(eps_a, 1e-3)
    roots, _ = ehm_tools.delaunay_roots(V)
    alone = None
    if rank == 0:
        t = Tree(NodeData(vertices=roots[2].copy()))
        frontier.grow_cells(nat, t)
        alone = [(loc, nd.is_leaf(), bool(nd.data.is_epsilon_suboptimal)) for nd, loc in t.walk()]
    dist.barrier()
    if rank == 1:
        time.sleep(0.5)             # rank 0 claims the only root
    trees = [Tree(NodeData(vertices=roots[2].copy()))]
    trees, stats, counts = distributed.grow_roots_sharded(
        None, trees, 'ecc', deal='dynamic', native=nat, steal=True, steal_slice=256,
        claim_key='share_one_root')
    got = [(loc, nd.is_leaf(), bool(nd.data.is_epsilon_suboptimal),
            bool(getattr(nd.data, 'remote', False))) for nd, loc in trees[0].walk()] \
        if stats['mine'] else None
    with open(os.path.join(out_dir, 'share%d.pkl' % rank), 'wb') as f:
        pickle.dump(dict(alone=alone, got=got, counts=counts,
                         stats={k: v for k, v in stats.items() if k != 'mine'},
                         mine=stats['mine']), f)
    nat.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_root_on_the_device(tmp_path):
    """
    distributed.grow_roots_sharded(deal='dynamic', steal=True) with DEVICE handles: two processes
    (gloo; they share the box's one GPU), ONE configs[4] root.  Rank 0 claims it, rank 1 asks for
    cells through the store, grows them on its own tables and sends the sub-trees back; rank 0's
    tree is the tree it grows alone.
    """
    import pickle
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank_shares_a_device_root, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [pickle.load(open(str(tmp_path / ('share%d.pkl' % r)), 'rb')) for r in range(2)]
    assert outs[0]['mine'] == [0] and outs[1]['mine'] == []
    s0, s1 = outs[0]['stats'], outs[1]['stats']
    assert s0['parcels_given'] >= 1 and s1['cells_adopted'] >= 1 and s1['regions'] > 0
    assert s0['subtrees_attached'] == s1['cells_adopted'] + s0['cells_adopted']
    alone = {loc: (leaf, closed) for loc, leaf, closed in outs[0]['alone']}
    got = {}
    for loc, leaf, closed, remote in outs[0]['got']:
        assert not remote, loc
        got[loc] = (leaf, closed)
    assert got == alone
    assert s0['regions'] + s1['regions'] == sum(1 for leaf, closed in alone.values()
                                                if leaf and closed)
