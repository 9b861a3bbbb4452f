"""Import the UNMODIFIED reference (``/root/reference/diffdrr``) -- test-only.

The reference is pure Python/PyTorch but needs three third-party modules that
are absent in the build container (fastcore, roma, torchio) plus three that
its task modules import at module level (timm, torchvision, kornia).  The
shims in ``oracle/ref_shims`` stand in for them (see that directory's README).

``/root/reference`` exists only in the build container, never on the GPU box:
this module is used by ``tests/golden/make_golden.py`` (to generate committed
fixtures) and by CPU tests that skip when the reference is absent.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DIFFDRR_REFERENCE", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shims")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "diffdrr"))


def load() -> types.SimpleNamespace:
    """Return a namespace with the reference's DRR, Siddon, Trilinear, ... classes."""
    if not available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    for p in (_SHIMS, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import diffdrr.drr as drr
    import diffdrr.detector as detector
    import diffdrr.metrics as metrics
    import diffdrr.pose as pose
    import diffdrr.registration as registration
    import diffdrr.renderers as renderers
    from torchio import LabelMap, ScalarImage, Subject

    return types.SimpleNamespace(
        DRR=drr.DRR, Detector=detector.Detector, Siddon=renderers.Siddon,
        Trilinear=renderers.Trilinear, renderers=renderers, pose=pose, convert=pose.convert,
        RigidTransform=pose.RigidTransform, Registration=registration.Registration,
        NCC=metrics.NormalizedCrossCorrelation2d, metrics=metrics,
        Subject=Subject, ScalarImage=ScalarImage, LabelMap=LabelMap,
    )
