"""
Reading and writing the reference's on-disk tree format.

The reference pickles nested ``tree.Tree`` / ``tree.NodeData`` objects (``tree.pkl``,
lib/scheduler.py:388-391; ``branch_<location>.pkl`` task dicts, lib/worker.py:456-458) and
its consumers (``ExplicitMPC``, ``PostProcessor``, ``build_tree``) unpickle them by the
module path ``tree``.  ``dump_reference`` writes pickles that name exactly that module and
those classes; ``load_reference`` reads the reference's pickles into this package's
classes.  Deep Delaunay spines need a raised recursion limit for the pickle itself
(SURVEY.md section 7 hard part 7); the limit is restored afterwards.
"""

import contextlib
import pickle
import sys

from . import tree as _tree


@contextlib.contextmanager
def _as_reference_module():
    saved_mod = sys.modules.get('tree')
    saved_names = (_tree.Tree.__module__, _tree.NodeData.__module__)
    sys.modules['tree'] = _tree
    _tree.Tree.__module__ = 'tree'
    _tree.NodeData.__module__ = 'tree'
    try:
        yield
    finally:
        _tree.Tree.__module__, _tree.NodeData.__module__ = saved_names
        if saved_mod is None:
            del sys.modules['tree']
        else:
            sys.modules['tree'] = saved_mod


@contextlib.contextmanager
def _deep_recursion(depth_hint):
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 10 * depth_hint + 1000))
    try:
        yield
    finally:
        sys.setrecursionlimit(old)


def tree_depth(root):
    depth = 0
    for _, loc in root.walk():
        depth = max(depth, len(loc))
    return depth


def dump_reference(obj, path, depth_hint=None):
    """Pickle ``obj`` (a Tree, or a task dict holding one) for the reference's consumers."""
    if depth_hint is None:
        root = obj['branch_root'] if isinstance(obj, dict) else obj
        depth_hint = tree_depth(root)
    with _as_reference_module(), _deep_recursion(depth_hint), open(path, 'wb') as f:
        pickle.dump(obj, f)


def load_reference(path, depth_hint=100000):
    """Unpickle a file written by the reference (or by ``dump_reference``)."""
    with _as_reference_module(), _deep_recursion(depth_hint), open(path, 'rb') as f:
        return pickle.load(f)


def branch_task(branch_root, location, action):
    """The reference's task dict (lib/scheduler.py:637-639, lib/worker.py:232-233)."""
    return dict(branch_root=branch_root, location=location, action=action)
