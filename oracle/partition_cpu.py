"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

CPU restatement of the partition algorithms ``Worker.ecc`` / ``Worker.lcss``
(reference lib/worker.py:241-417): the same oracle sequence per visited node, the same
child construction (left child gets the midpoint in slot v_idx[0], right child in slot
v_idx[1], lib/worker.py:356-365), written with an explicit work list instead of Python
recursion and MPI off-loading (the tree does not depend on the visiting order because a
node's fate depends only on its own record).

Nodes are plain dicts keyed by their location string ('0' = left, '1' = right, relative
to the root they descend from):
    vertices, commutation (or None), vertex_costs, vertex_inputs,
    is_epsilon_suboptimal, leaf (bool)
"""

import numpy as np

from . import geometry


def _new(vertices, commutation=None, vertex_costs=None, vertex_inputs=None):
    return dict(vertices=np.array(vertices, dtype=np.float64), commutation=commutation,
                vertex_costs=vertex_costs, vertex_inputs=vertex_inputs,
                is_epsilon_suboptimal=False, leaf=True)


class PartitionCPU:
    def __init__(self, oracle, max_nodes=None):
        self.oracle = oracle
        self.max_nodes = max_nodes
        self.nodes = {}
        self.visits = 0
        self.volume_closed = 0.
        self.min_margin = np.inf
        self.truncated = False
        # path codes of the nodes (root k: k; children 2 c, 2 c + 1 modulo 2^32 -- DevTree::code of
        # the device), handed to the oracle before V_R / bar_D for its rule 'hash'
        self.codes = {}

    def _code(self, loc):
        c = self.codes.get(loc)
        if c is None:
            # a child: the nearest ancestor with a known code, then one bit per turn.  (A run whose
            # nodes were injected without ``run`` -- bench.py's CPU-baseline workers -- has no root
            # codes: code 0 at the top; only the rule 'hash' of the oracle reads them.)
            k = len(loc)
            while k > 0 and loc[:k] not in self.codes:
                k -= 1
            c = self.codes.get(loc[:k], 0)
            for ch in loc[k:]:
                c = (2 * c + (1 if ch == '1' else 0)) & 0xffffffff
            self.codes[loc] = c
        return c

    # -- one node visit each -------------------------------------------------------------
    def _ecc_visit(self, loc, work):
        node = self.nodes[loc]
        c_R = np.average(node['vertices'], axis=0)             # lib/worker.py:264
        if not self.oracle.P_theta(theta=c_R, check_feasibility=True):
            raise RuntimeError('STOP, Theta contains infeasible regions')
        self.oracle.node_code = self._code(loc)
        delta_hat, vx = self.oracle.V_R(node['vertices'])       # lib/worker.py:268
        if delta_hat is None:
            S_1, S_2 = geometry.split_along_longest_edge(node['vertices'])[:2]
            node['leaf'] = False
            self.nodes[loc + '0'] = _new(S_1)
            self.nodes[loc + '1'] = _new(S_2)
            work.append((loc + '1', 'ecc'))
            work.append((loc + '0', 'ecc'))
        else:
            node['commutation'] = delta_hat
            node['vertex_costs'] = np.array([v[1] for v in vx])
            node['vertex_inputs'] = np.array([v[0] for v in vx])
            work.append((loc, 'lcss'))

    def _lcss_visit(self, loc, work):
        node = self.nodes[loc]
        orc = self.oracle
        closed = orc.bar_E_delta_R(R=node['vertices'], V_delta_R=node['vertex_costs'])
        self.min_margin = min(self.min_margin, getattr(orc, 'last_margin', np.inf))
        if closed:                                              # lib/worker.py:369-375
            node['is_epsilon_suboptimal'] = True
            self.volume_closed += geometry.simplex_volume(node['vertices'])
            return
        orc.node_code = self._code(loc)
        delta_star, theta_star, new_vx, varies_little = orc.bar_D_delta_R(
            R=node['vertices'], V_delta_R=node['vertex_costs'],
            delta_ref=node['commutation'])
        feasible = delta_star is not None
        if not feasible:                                        # lib/worker.py:381-386
            delta_star = node['commutation']
            new_costs = node['vertex_costs']
            new_inputs = node['vertex_inputs']
        else:
            new_costs = np.array([v[1] for v in new_vx])
            new_inputs = np.array([v[0] for v in new_vx])
        if feasible and varies_little:                          # lib/worker.py:396-401
            node['commutation'] = delta_star
            node['vertex_costs'] = new_costs
            node['vertex_inputs'] = new_inputs
            work.append((loc, 'lcss'))
            return
        S_1, S_2, v_idx = geometry.split_along_longest_edge(node['vertices'])
        v_mid = S_1[v_idx[0]]                                   # lib/worker.py:406
        u_mid, V_mid = orc.P_theta_delta(theta=v_mid, delta=delta_star)[:2]
        in_1, in_2 = new_inputs.copy(), new_inputs.copy()
        co_1, co_2 = new_costs.copy(), new_costs.copy()
        in_1[v_idx[0]] = u_mid
        in_2[v_idx[1]] = u_mid
        co_1[v_idx[0]] = V_mid
        co_2[v_idx[1]] = V_mid
        node['leaf'] = False
        self.nodes[loc + '0'] = _new(S_1, delta_star, co_1, in_1)
        self.nodes[loc + '1'] = _new(S_2, delta_star, co_2, in_2)
        work.append((loc + '1', 'lcss'))
        work.append((loc + '0', 'lcss'))

    # -- driver ------------------------------------------------------------------------------
    def run(self, roots, locations, action='ecc'):
        """
        roots: list of (p+1, p) vertex arrays or of node dicts (for action 'lcss' they
        must already carry commutation / vertex costs / vertex inputs).
        """
        work = []
        for k, (R, loc) in enumerate(zip(roots, locations)):
            self.codes[loc] = k
            self.nodes[loc] = R if isinstance(R, dict) else _new(R)
            work.append((loc, action))
        work.reverse()
        self._work = work
        return self.resume()

    def resume(self):
        """Continue a run that stopped at ``max_nodes`` (raise the limit first)."""
        work = self._work
        self.truncated = False
        while work:
            if self.max_nodes is not None and self.visits >= self.max_nodes:
                self.truncated = True
                break
            loc, act = work.pop()
            self.visits += 1
            if act == 'ecc':
                self._ecc_visit(loc, work)
            else:
                self._lcss_visit(loc, work)
        return self.nodes

    def leaves(self):
        return {k: v for k, v in self.nodes.items() if v['leaf']}
