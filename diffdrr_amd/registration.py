"""2D/3D registration module (reference ``diffdrr/registration.py:14-50``):
learnable pose parameters in front of a ``DRR``.  ``PoseRegressor`` (a CNN) is
out of scope."""
from __future__ import annotations

import torch
import torch.nn as nn

from .pose import convert


class Registration(nn.Module):
    def __init__(self, drr, rotation: torch.Tensor, translation: torch.Tensor,
                 parameterization: str, convention: str | None = None):
        super().__init__()
        self.drr = drr
        self._rotation = nn.Parameter(rotation)
        self._translation = nn.Parameter(translation)
        self.parameterization = parameterization
        self.convention = convention

    def forward(self, **kwargs):
        # (same result as the reference's `self.drr(self.pose, **kwargs)`; handing the raw
        # parameters over lets DRR take its fused pose -> rays path for Euler angles)
        return self.drr(self._rotation, self._translation,
                        parameterization=self.parameterization, convention=self.convention,
                        **kwargs)

    @property
    def pose(self):
        return convert(self._rotation, self._translation,
                       parameterization=self.parameterization, convention=self.convention)

    @property
    def rotation(self):
        return self._rotation

    @property
    def translation(self):
        return self._translation
