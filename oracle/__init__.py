"""oracle/ -- CPU restatement of the reference's 3D-RetinaNet hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker /
the timed CPU baseline.  The product path (``3d-playground_amd/``) never
imports it and fails loudly when the HIP library is missing.

Every function cites the reference file:line it restates
(paths relative to the reference checkout; ``D/`` =
``pytorch_retinanet_detector_directional/retinanet/``, ``R/`` = ``retinanet/``).

Pinning: the restatement is checked against golden vectors produced by
importing the reference itself in the build container
(``tools/make_golden.py`` -> ``tests/golden/*.npz``) and against the
reference's own CSV result files (known-answer rows for the homography
transforms).  Third-party pieces the reference calls but does not ship
(``torchvision.ops.nms``) are restated from their documented contract and are
"parity unpinned" -- see ``oracle/boxes.py``.
"""
