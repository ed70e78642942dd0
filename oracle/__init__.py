"""CPU oracle for the filter hot path.  TEST INFRASTRUCTURE ONLY.

This package is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``multimodalfilter_amd/`` imports from here.

What it restates, and what pins it
----------------------------------
* ``oracle.tf`` / ``oracle.fp`` -- the subset of the third-party ``torchfilter`` and
  ``fannypack`` packages that the reference subclasses and calls
  (``/root/reference/setup.py:12-15`` declares them un-pinned:
  ``torchfilter @ .../tarball/master``; neither is vendored, installed or
  downloadable here).  Their published algorithm is restated from the reference's
  call sites (``crossmodal/eval_helpers.py:125-142``, ``door_models/pf.py:14-27``,
  ``door_models/kf.py:14-28``, ``base_models/crossmodal_kf.py:147-149,180``).
  **Parity for the recursion itself (T1/T2/T3) is unpinned by the reference** -- it
  holds no tests, golden vectors or fixtures.  It is pinned instead by analytic
  known answers (linear-Gaussian Kalman closed form, Jacobian vs. finite
  differences, resampling invariants) in ``tests/test_oracle_known_answers.py``.
* ``oracle.models`` -- the crossmodal layer (dynamics / measurement / virtual-sensor
  / weight models and the crossmodal + unimodal fusion math) restated from
  ``/root/reference/crossmodal/{base_models,door_models,push_models}``.  This part
  IS pinned: ``oracle/capture_golden.py`` imports the reference's own ``crossmodal``
  package in the build container (on top of ``oracle.tf``/``oracle.fp``) and writes
  ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` holds the oracle to them.
* ``oracle.resample`` -- the normative fixed-point systematic / multinomial
  resampler (integer CDF; bit-exact by construction for any scan order).
* ``oracle.evalmetrics`` -- the RMSE arithmetic of ``crossmodal/eval_helpers.py:149-160``.
"""
