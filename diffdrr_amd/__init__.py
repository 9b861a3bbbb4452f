"""diffdrr_amd: DiffDRR's rendering hot path on AMD Instinct MI355X (gfx950).

Drop-in for the Siddon ray-caster and trilinear ray-marcher of
eigenvivek/DiffDRR (``diffdrr/renderers.py``) as hand-written HIP kernels behind
the reference's own module API (``DRR``, ``Detector``, ``RigidTransform`` /
``convert``, ``Siddon``, ``Trilinear``, ``Registration``).  See DESIGN.md.
"""
__version__ = "0.1.0"

from .detector import Detector  # noqa: F401
from .drr import DRR  # noqa: F401
from .metrics import (  # noqa: F401
    GradientNormalizedCrossCorrelation2d,
    MultiscaleNormalizedCrossCorrelation2d,
    NormalizedCrossCorrelation2d,
)
from .pose import RigidTransform, convert  # noqa: F401
from .registration import GraphedIteration, PoseAdam, Registration  # noqa: F401
from .renderers import Siddon, Trilinear  # noqa: F401
