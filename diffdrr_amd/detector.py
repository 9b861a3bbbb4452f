"""C-arm detector: turns a camera pose into ray endpoints.

API-compatible restatement of reference ``diffdrr/detector.py:17-154``.  Kept
in PyTorch (12 bytes per ray, differentiable w.r.t. the pose); the renderers'
HIP kernels consume its ``(source, target)`` output.

Geometry: the source sits at the origin, the detector
plane at z = 1 is spanned by x (columns, flipped when ``reverse_x_axis``) and y
(rows); ray ``n = i * width + j`` (row-major over height then width,
detector.py:126); ``calibration = [[delx,0,0,x0],[0,dely,0,y0],[0,0,sdd,0],
[0,0,0,1]]`` scales the plane to world units (detector.py:50-60) and
``reorient.compose(pose)`` places it (detector.py:151-153).
"""
from __future__ import annotations

import torch

from .pose import RigidTransform


class Detector(torch.nn.Module):
    def __init__(
        self,
        sdd: float,
        height: int,
        width: int,
        delx: float,
        dely: float,
        x0: float,
        y0: float,
        reorient: torch.Tensor,
        n_subsample: int | None = None,
        reverse_x_axis: bool = False,
    ):
        super().__init__()
        self.height = height
        self.width = width
        self.n_subsample = n_subsample
        if self.n_subsample is not None:
            self.subsamples = []
        self.reverse_x_axis = reverse_x_axis

        source, target = self._initialize_carm()
        self.register_buffer("source", source)
        self.register_buffer("target", target)
        self.register_buffer("_reorient", reorient)
        self.register_buffer(
            "_calibration",
            torch.tensor(
                [
                    [delx, 0, 0, x0],
                    [0, dely, 0, y0],
                    [0, 0, sdd, 0],
                    [0, 0, 0, 1],
                ],
                dtype=torch.float32,
            ),
        )

    # NB: like the reference (detector.py:74-80) the x0 / y0 *properties* return
    # the negated matrix entries.
    @property
    def sdd(self):
        return self._calibration[2, 2].item()

    @property
    def delx(self):
        return self._calibration[0, 0].item()

    @property
    def dely(self):
        return self._calibration[1, 1].item()

    @property
    def x0(self):
        return -self._calibration[0, -1].item()

    @property
    def y0(self):
        return -self._calibration[1, -1].item()

    @property
    def reorient(self):
        return RigidTransform(self._reorient)

    @property
    def calibration(self):
        """4x4 matrix that rescales the unit detector plane to world units."""
        return RigidTransform(self._calibration)

    @property
    def intrinsic(self):
        """3x3 pinhole intrinsic matrix."""
        return make_intrinsic_matrix(self).to(self.source)

    def _unit_grid(self):
        """(1, H W, 3) unit-spaced pixel centres on the plane z = 1, row-major (detector.py:97-131)."""
        h_off = 1.0 if self.height % 2 else 0.5
        w_off = 1.0 if self.width % 2 else 0.5
        rows = -(torch.arange(-self.height // 2, self.height // 2) + h_off)
        cols = -(torch.arange(-self.width // 2, self.width // 2) + w_off)
        if not self.reverse_x_axis:
            cols = -cols
        yy, xx = torch.meshgrid(rows, cols, indexing="ij")
        return torch.stack([xx, yy, torch.ones_like(xx)], dim=-1).reshape(1, -1, 3).to(torch.float32)

    def full_target(self):
        """The WHOLE detector grid's unit points, whatever ``n_subsample`` kept of it in ``target``
        (the volume-stationary kernels render grids: a subsample is rendered as its grid and
        gathered, ``DRR._render_fused_Mw``).  Not a registered buffer -- ``state_dict`` stays the
        reference's -- but cached per device."""
        if self.n_subsample is None:
            return self.target
        cached = getattr(self, "_full_target", None)
        if cached is None or cached.device != self.target.device or cached.dtype != self.target.dtype:
            cached = self._unit_grid().to(self.target)  # (device and dtype of the module)
            self._full_target = cached
        return cached

    def subsample_index(self):
        """The pixel indices ``target`` holds (``subsamples[-1]``, detector.py:134-137) as an int64
        tensor on the detector's device, in the order the renderer's output has; None without
        ``n_subsample``."""
        if self.n_subsample is None:
            return None
        dev = self.target.device
        cached = getattr(self, "_subsample_index", None)
        if cached is None or cached.device != dev or cached.numel() != len(self.subsamples[-1]):
            cached = torch.tensor(self.subsamples[-1], dtype=torch.int64, device=dev)
            self._subsample_index = cached
        return cached

    def subsample_mask(self):
        """One bit per pixel of the whole grid, set for the subsample's pixels (``ops.pixel_mask_of``):
        what the brick kernels take to render the subsample alone; cached per device."""
        idx = self.subsample_index()
        if idx is None:
            return None
        cached = getattr(self, "_subsample_mask", None)
        if cached is None or cached[0] is not idx:
            from . import ops
            cached = (idx, ops.pixel_mask_of(idx, self.height * self.width))
            self._subsample_mask = cached
        return cached[1]

    def _initialize_carm(self):
        """Unit-spaced pixel centres on the plane z = 1, centred on the optical axis."""
        target = self._unit_grid()
        source = torch.zeros(1, 1, 3)
        if self.n_subsample is not None:
            sample = torch.randperm(self.height * self.width)[: int(self.n_subsample)]
            target = target[:, sample, :]
            self.subsamples.append(sample.tolist())
        return source, target.to(torch.float32)

    def forward(self, extrinsic: RigidTransform, calibration: RigidTransform | None):
        """``(source (B,1,3), target (B,N,3))`` in world coordinates."""
        cal = self.calibration if calibration is None else calibration
        target = cal(self.target)
        pose = self.reorient.compose(extrinsic)
        return pose(self.source), pose(target)


def get_focal_length(intrinsic, delx: float, dely: float) -> float:
    fx, fy = intrinsic[0, 0], intrinsic[1, 1]
    return abs((fx * delx) + (fy * dely)).item() / 2.0


def get_principal_point(intrinsic, height: int, width: int, delx: float, dely: float):
    x0 = delx * (intrinsic[0, 2] - width / 2)
    y0 = dely * (intrinsic[1, 2] - height / 2)
    return x0.item(), y0.item()


def parse_intrinsic_matrix(intrinsic, height: int, width: int, delx: float, dely: float):
    focal_length = get_focal_length(intrinsic, delx, dely)
    x0, y0 = get_principal_point(intrinsic, height, width, delx, dely)
    return focal_length, x0, y0


def make_intrinsic_matrix(detector: Detector):
    fx = detector.sdd / detector.delx
    fy = detector.sdd / detector.dely
    u0 = detector.x0 / detector.delx + detector.width / 2
    v0 = detector.y0 / detector.dely + detector.height / 2
    return torch.tensor([[fx, 0.0, u0], [0.0, fy, v0], [0.0, 0.0, 1.0]])
