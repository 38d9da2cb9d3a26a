"""
Binary partition tree and node payload with the reference's API (lib/tree.py:12-95), so
that pickles written here load in the reference's consumers (``ExplicitMPC``,
``PostProcessor``, ``build_tree``) and vice versa.

Compatibility notes that matter to those consumers:
* optional node attributes (``commutation``, ``vertex_costs``, ``vertex_inputs``) are
  ABSENT rather than ``None`` when not supplied (lib/tree.py:34-39);
* a node is a leaf iff it has no ``left`` attribute (lib/tree.py:86-95);
* ``copy`` shares the payload and the children (lib/tree.py:57-70).

Everything that walks a tree here is iterative: the Delaunay right spine alone is
hundreds to tens of thousands of levels deep for p >= 6 (hard part 7).
"""

import time


class NodeData:
    """Payload of one partition cell (a simplex and what is known on its vertices)."""

    def __init__(self, vertices, commutation=None, vertex_costs=None, vertex_inputs=None):
        self.timestamp = time.time()
        self.vertices = vertices
        self.is_epsilon_suboptimal = False
        for name, value in (('commutation', commutation), ('vertex_costs', vertex_costs),
                            ('vertex_inputs', vertex_inputs)):
            if value is not None:
                setattr(self, name, value)


class Tree:
    """Node of the binary partition tree; ``data`` may be None on Delaunay spine nodes."""

    def __init__(self, data, top=True):
        self.data = data
        self.top = top

    def is_leaf(self):
        return not hasattr(self, 'left')

    def grow(self, left, right):
        """Attach two children holding the payloads ``left`` and ``right``."""
        self.left = Tree(left, top=False)
        self.right = Tree(right, top=False)

    def copy(self, node):
        """Make ``node`` an alias of this node (payload and children are shared)."""
        node.data = self.data
        node.top = self.top
        if not self.is_leaf():
            node.left = self.left
            node.right = self.right

    # -- iterative helpers (not in the reference) --------------------------------------------
    def walk(self, location=''):
        """Yield (node, location) for every node, depth first, left before right."""
        stack = [(self, location)]
        while stack:
            node, loc = stack.pop()
            yield node, loc
            if not node.is_leaf():
                stack.append((node.right, loc + '1'))
                stack.append((node.left, loc + '0'))

    def leaves(self, location=''):
        for node, loc in self.walk(location):
            if node.is_leaf():
                yield node, loc

    def descend(self, location):
        """Node reached by following the '0'/'1' string ``location`` from here."""
        node = self
        for ch in location:
            node = node.left if ch == '0' else node.right
        return node
