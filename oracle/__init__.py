"""CPU oracle for the CenterFusion forward path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (pure torch fp32 / numpy fp64, our own code)
of the algorithm the reference implements on the hot path named by
BASELINE.json's north_star.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the
checker - never as the thing that is measured or shipped.  The product package
(`centerfusiondetect3d_amd`) must not import anything from here.

Pinning status (see DESIGN.md "Oracle"):
  * model forward / heads / frustum association / topk / decode: pinned by
    golden vectors generated here from the reference's own Python, imported
    from /root/reference (tests/golden/make_golden.py).
  * deform_conv2d arithmetic (third-party torchvision, absent from
    /root/reference and from this image, version not pinned by the reference):
    PARITY UNPINNED - restated from torchvision's documented semantics and
    held by known-answer tests only (tests/test_oracle_dcn.py).
  * pillar expansion, radar ingest, nuScenes result serialisation: PINNED by
    fixtures generated from the reference's own dataset classes
    (tests/golden/make_golden_dataset.py; arithmetic stand-ins for absent
    libraries: cv2.transform / getAffineTransform, devkit view_points - stated
    in the generator); bit-exact.  The orientation quaternion of the result
    file (pyquaternion / devkit Box, absent) alone is restated, unpinned.
  * image pre-processing (cv2.warpAffine, absent): PARITY UNPINNED - restated
    from OpenCV's published fixed-point algorithm, held by known-answer tests.
"""
