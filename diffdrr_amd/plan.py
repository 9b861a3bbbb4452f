"""Scheduling plan for the slab-march Siddon kernel (device-side, no host sync).

For every pose of a detector-grid render this picks
  * the march axis: the volume axis (x or y) the central ray is most aligned with;
    z-dominant poses (2) take the generic per-crossing walk,
  * the wave composition: rays that lie in one plane through the source containing the
    volume's z axis stay in the same (x, y) row of the volume at equal depth, so their
    voxels differ only along z, the contiguous axis.  On the detector those planes are
    straight lines; ``major`` says whether they run closer to the detector's row index
    (0) or column index (1) and ``shear[b, strip]`` is their local slope at each strip
    of 64 pixels (csrc/ddrr_common.h ShearMap).
Nothing here affects results -- only which rays share a wavefront.
"""
from __future__ import annotations

import torch


def slab_plan(source: torch.Tensor, target: torch.Tensor, det_h: int, det_w: int):
    """source (B,1,3), target (B,H*W,3) in voxel coordinates ->
    plan (B,2) int32, shear (B,S) float32 with S = ceil(max(H,W)/64)."""
    B = target.shape[0]
    H, W = det_h, det_w
    S = (max(H, W) + 63) // 64
    dev = target.device
    if H < 2 or W < 2:
        plan = torch.zeros(B, 2, dtype=torch.int32, device=dev)
        plan[:, 0] = 2
        return plan, torch.zeros(B, S, dtype=torch.float32, device=dev)
    with torch.no_grad():
        t = target.detach().view(B, H, W, 3)
        s = source.detach().reshape(B, 1, 3)
        ic, jc = min(H // 2, H - 2), min(W // 2, W - 2)
        d_c = t[:, ic, jc] - s[:, 0]
        march = d_c.abs().argmax(dim=-1).to(torch.int32)

        def slopes(pts_i, pts_j):
            """pts_*: pixel coordinates (S,) of the strip centres; returns the in-plane
            coefficients a = e_i . n, b = e_j . n at those pixels, n = d x z_hat."""
            p = t[:, pts_i, pts_j]                         # (B,S,3)
            d = p - s
            n = torch.stack([d[..., 1], -d[..., 0], torch.zeros_like(d[..., 0])], dim=-1)
            e_i = t[:, pts_i + 1, pts_j] - p
            e_j = t[:, pts_i, pts_j + 1] - p
            return (e_i * n).sum(-1), (e_j * n).sum(-1)

        r = torch.arange(S, device=dev)
        ci = (64 * r + 32).clamp_max(H - 2)
        cj = (64 * r + 32).clamp_max(W - 2)
        a0, b0 = slopes(ci, torch.full_like(ci, jc))       # strips of rows (major 0)
        a1, b1 = slopes(torch.full_like(cj, ic), cj)       # strips of columns (major 1)
        ac, bc = slopes(torch.tensor([ic], device=dev), torch.tensor([jc], device=dev))
        # the epipolar line direction in pixel space is (di, dj) ~ (b, -a)
        major = (bc.abs() < ac.abs()).to(torch.int32).reshape(B)  # 0: along rows index i
        tiny = 1e-20
        sig0 = (-a0 / torch.where(b0.abs() < tiny, torch.full_like(b0, tiny), b0))  # dj/di
        sig1 = (-b1 / torch.where(a1.abs() < tiny, torch.full_like(a1, tiny), a1))  # di/dj
        shear = torch.where(major[:, None] == 0, sig0, sig1).clamp(-4.0, 4.0)
        plan = torch.stack([march, major], dim=-1).contiguous()
    return plan, shear.to(torch.float32).contiguous()
