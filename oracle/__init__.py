"""
CPU oracle for the partitioning hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package; the product package
``explicit_hybrid_mpc_amd`` never does and fails loudly when its HIP library is missing.

What is restated and from where (paths into the reference tree):

* geometry  -- ``lib/tools.py:134-257`` (``simplex_volume``, ``delaunay``,
  ``split_along_longest_edge``): ``oracle/geometry.py`` (+ ``oracle/csrc/geom_ref.c``
  for the fused-multiply-add accumulation numpy/OpenBLAS performs in ``x.dot(x)``).
  PINNED: checked against fixtures produced by the unmodified reference functions
  (``tests/golden/make_geometry_golden.py`` imports ``/root/reference/lib/tools.py``).
* node types -- ``lib/tree.py:12-95``: ``oracle/partition_cpu.py`` keeps plain dict nodes
  and is compared structurally with the product's ``Tree`` objects.
* the six optimisation oracles -- ``lib/oracle.py:23-443``: ``oracle/oracle_cpu.py``.
  The reference delegates the arithmetic to CVXPY 1.0.21 -> MOSEK 9.0.87 (closed source,
  ``requirements.txt:2``, ``README.md:39-41``, selected at ``lib/global_vars.py:25``);
  neither is installable here and the reference ships no tests, fixtures or golden
  vectors for any oracle.  The restatement therefore poses the same
  optimisation problems (same constraint sets, same return conventions) on the
  *uncondensed* state/input variables and solves them with SciPy's HiGHS
  (``scipy.optimize.linprog``).  PARITY UNPINNED for this half: there is nothing of the
  reference's to pin it to; it is cross-checked against a second, independent solver
  path (``oracle/ipm_numpy.py``) instead.
* node semantics of the partition algorithms -- ``lib/worker.py:241-417`` (``ecc``,
  ``lcss``): ``oracle/partition_cpu.py`` (iterative, same per-node oracle sequence).

Canonical choices where the reference leaves the answer to the solver (mixed-integer
feasibility problems return *a* feasible commutation, lib/oracle.py:201,347) are
documented in DESIGN.md ("canonical commutation rule") and implemented identically
here and in the HIP kernels.
"""
