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
  (``scipy.optimize.linprog``) for infinity-norm costs and with ``oracle/qp_numpy.py`` for
  the reference's quadratic costs.  PINNED for the quadratic cost class: the only numbers
  of this half the reference ships -- the five ``P_theta`` optima hard-coded in
  ``lib/post_process.py:484-485`` (absolute tolerances of its cwh_z runs, rule
  ``lib/examples.py:42-45``, job parameters ``make_jobs.sh:60-66``) -- are reproduced by
  ``oracle/satellite_cpu.py`` (line-by-line restatement of ``SatelliteZ``,
  ``lib/mpc_library.py:61-272``) + ``OracleCPU.P_theta`` to <= 7e-8 absolute
  (``tests/test_oracle_satellite.py``, fixture ``tests/golden/known_answers.json`` made by
  ``tests/golden/make_known_answers.py``).  The synthetic infinity-norm LP instances of
  BASELINE.json have no counterpart in the reference: for them the oracle is cross-checked
  against a second, independent solver path (``oracle/ipm_numpy.py``).
* ``oracle/cut_bound.py`` restates the device's tangent-plane bound of the suboptimality-test
  optimum with HiGHS duals (not a reference function: a check that the bound never falls below
  the optimum the reference's ``bar_E_delta_R`` problem has).
* ``oracle/prefix_bb.py`` restates on the uncondensed models the relaxations and searches over
  mode prefixes the product uses where the reference relies on its solver's branch-and-bound
  (``lib/oracle.py:42-46, 89-102``; product: ``sequences.py``, ``bnb.py``), and offers a HiGHS
  stand-in of the device table so the CPU tests can run those searches;
  ``oracle/milp_check.py`` states ``P_theta`` and the ``bar_E`` problem each as ONE
  mixed-integer LP (binary mode indicators, big-M dynamics -- the reference's own
  formulation, lib/oracle.py:42-46, 89-97) for HiGHS' branch-and-bound: the independent pin
  of enumeration and prefix search on small instances.
* node semantics of the partition algorithms -- ``lib/worker.py:241-417`` (``ecc``,
  ``lcss``): ``oracle/partition_cpu.py`` (iterative, same per-node oracle sequence).

Canonical choices where the reference leaves the answer to the solver (mixed-integer
feasibility problems return *a* feasible commutation, lib/oracle.py:201,347) are
documented in DESIGN.md ("canonical commutation rule") and implemented identically
here and in the HIP kernels.
"""
