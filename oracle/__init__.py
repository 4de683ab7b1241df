"""CPU oracle for the EpipolarPose hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package restates, in float64 NumPy (and plain fp32 torch-CPU for the network),
the algorithms of the reference's training hot path so that the HIP kernels can
be checked against them.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
``epipolarpose_amd`` never imports ``oracle`` and has no CPU fallback.

Pinning status (see DESIGN.md section "Oracle"):

* ``oracle.integral``      -- pinned: golden vectors produced by importing the
  reference's ``lib/core/integral_loss.py`` (torch CPU, autograd) in
  ``tests/golden/make_golden.py``.
* ``oracle.network``       -- pinned: golden vectors produced by the reference's
  ``lib/models/pose3d_resnet.py`` on deterministic weights.
* ``oracle.triangulation``, ``oracle.geometry`` -- pinned against the reference's
  OWN Python control flow (``lib/utils/triangulation.py``, ``lib/utils/img_utils.py``,
  ``lib/utils/prep_h36m.py``, ``lib/utils/cameras.py`` executed live), with the
  absent third-party OpenCV primitives (``cv2.solve(DECOMP_SVD)``,
  ``cv2.getAffineTransform``, ``cv2.triangulatePoints``, ``cv2.invert``,
  ``cv2.correctMatches``, ``cv2.findFundamentalMat(FM_8POINT)``; conda pin
  opencv=4.1.0) substituted by float64 NumPy equivalents of their published
  algorithms.  The OpenCV primitives themselves are therefore *restated*, not
  executed: "parity unpinned" applies to that third-party layer only.  For
  ``correctMatches`` the golden generator deliberately runs a DIFFERENT algorithm
  (Kanatani's iterated optimal correction) under the reference's glue, so that the
  oracle's Hartley-Sturm polynomial is cross-checked by an independent solver;
  the 8-point estimate is checked by its defining properties (exact on noise-free
  matches, rank 2, transpose / translation covariance).
* ``oracle.evaluation``    -- pinned: ``H36M_Integral.evaluate`` and
  ``compute_similarity_transform`` executed live for the golden vectors.
* ``oracle.inference``     -- pinned: ``get_max_preds`` is pure NumPy in the
  reference and is executed live for the golden vectors.
"""
