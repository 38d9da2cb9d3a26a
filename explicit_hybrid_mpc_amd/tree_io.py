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
import threading

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


# The pickle of a nested tree recurses once per level, and so does the Delaunay right spine
# (34 573 levels for the 8-dimensional box).  A raised recursion limit alone lets the C stack
# overflow (a segfault instead of RecursionError), so the (un)pickling runs in a thread whose
# stack is sized for the depth; beyond MAX_DEPTH the call fails with a clear error.
MAX_DEPTH = 200000
_STACK_BYTES_PER_LEVEL = 8192      # C stack per TREE level: the pickler nests ~4 save() calls per
                                   # level (Tree -> __dict__ -> child list -> Tree) at ~0.5 KB
                                   # each, 4x margin
_STACK_MAX = 2 << 30
_lock = threading.Lock()           # sys.modules['tree'] / the recursion limit are process-global


def _run_deep(fn, depth_hint):
    if depth_hint > MAX_DEPTH:
        raise ValueError('tree depth %d exceeds the supported %d levels' % (depth_hint, MAX_DEPTH))
    out = {}

    def work():
        try:
            out['value'] = fn()
        except BaseException as e:       # re-raised in the caller's thread
            out['error'] = e
    with _lock:
        old_limit, old_stack = sys.getrecursionlimit(), threading.stack_size()
        sys.setrecursionlimit(max(old_limit, 10 * depth_hint + 1000))
        # (the recursion limit is process-wide while this runs; the lock serialises callers)
        stack = min(_STACK_MAX, max(64 << 20, (depth_hint + 1000) * _STACK_BYTES_PER_LEVEL))
        threading.stack_size(stack)
        try:
            with _as_reference_module():
                t = threading.Thread(target=work)
                try:
                    t.start()
                except (RuntimeError, MemoryError) as e:
                    raise MemoryError('cannot start a thread with a %d MB stack for a tree of '
                                      'depth %d (address-space limit?): %s' %
                                      (stack >> 20, depth_hint, e))
                t.join()
        finally:
            threading.stack_size(old_stack)
            sys.setrecursionlimit(old_limit)
    if 'error' in out:
        raise out['error']
    return out.get('value')


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
    def write():
        with open(path, 'wb') as f:
            pickle.dump(obj, f)
    _run_deep(write, depth_hint)


def load_reference(path, depth_hint=40000):
    """
    Unpickle a file written by the reference (or by ``dump_reference``).  ``depth_hint``: an
    upper bound of the nesting depth (default: enough for the 8-dimensional Delaunay spine).
    """
    def read():
        with open(path, 'rb') as f:
            return pickle.load(f)
    return _run_deep(read, depth_hint)


def branch_task(branch_root, location, action):
    """The reference's task dict (lib/scheduler.py:637-639, lib/worker.py:232-233)."""
    return dict(branch_root=branch_root, location=location, action=action)
