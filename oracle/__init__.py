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
  * pillar expansion (reference module cannot be imported: cv2 / pycocotools /
    lightning at module top): PARITY UNPINNED - restated from the source text
    and held by hand-computable cases + property tests.
"""
