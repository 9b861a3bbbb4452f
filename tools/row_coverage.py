"""VERDICT r05 next 8, the estimate asked for before anything is built: at ONE pose, which fraction of a
brick's z-rows (runs of 64 voxels along z at one (x, y): what the staging copies) do the pose's rays touch at all?
If it were small, a one-pose launch could stage only the rows inside the pose's footprint and cut the 273 MB of
packed bricks that set its ~0.05 ms floor.  Host script (numpy): the bench's geometry (512^3, 256 x 256 detector,
delx 2.4, sdd 1020, source 850 mm away), its base pose and perturbed poses; every ray is sampled every 0.25 voxel
and marks the (x, y, z // 64) rows it enters; rows of bricks no ray enters at all are counted separately (the kernel
passes over those bricks' walks anyway, it still stages them)."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, convert  # noqa: E402
from diffdrr_amd.data import make_subject  # noqa: E402

D, H = 512, 256
drr = DRR(make_subject(torch.zeros(8, 8, 8), spacing=(1.0, 1.0, 1.0), orientation="AP"), sdd=1020.0, height=H, delx=2.4)
# (the rays only need the affine of a 512^3 volume: centred, unit spacing)
A = torch.eye(4)
A[:3, 3] = -(D - 1) / 2
Ainv = torch.linalg.inv(A)
g = torch.Generator().manual_seed(2)
B = 6
rot = (torch.rand(B, 3, generator=g) - 0.5) * (math.pi / 2)
xyz = torch.tensor([0.0, 850.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 60.0
rot[0] = 0.0
xyz[0] = torch.tensor([0.0, 850.0, 0.0])
with torch.no_grad():
    src, tgt = drr.detector(convert(rot, xyz, parameterization="euler_angles", convention="ZXY"), None)
    s = (src @ Ainv[:3, :3].T + Ainv[:3, 3]).numpy().astype(np.float64)
    t = (tgt @ Ainv[:3, :3].T + Ainv[:3, 3]).numpy().astype(np.float64)
print("pose, rows touched / rows of the volume, rows touched / rows of the bricks that are entered, bricks entered / 2048")
for b in range(B):
    rows = np.zeros((D, D, D // 64), bool)
    d = t[b] - s[b, 0]                                   # (N, 3)
    L = np.linalg.norm(d, axis=1)
    # slab clip of every ray against the volume [-0.5, D - 0.5]^3
    with np.errstate(divide="ignore", invalid="ignore"):
        a0 = (-0.5 - s[b, 0]) / d
        a1 = (D - 0.5 - s[b, 0]) / d
    lo = np.minimum(a0, a1).max(axis=1)
    hi = np.maximum(a0, a1).min(axis=1)
    ok = hi > lo
    n_steps = int(np.ceil(((hi - lo) * L)[ok].max() / 0.25)) + 1
    for c0 in range(0, d.shape[0], 8192):
        sel = np.arange(c0, min(c0 + 8192, d.shape[0]))
        sel = sel[ok[sel]]
        if not sel.size:
            continue
        al = lo[sel, None] + (hi[sel] - lo[sel])[:, None] * (np.arange(n_steps) + 0.5)[None, :] / n_steps
        p = s[b, 0][None, None, :] + al[:, :, None] * d[sel][:, None, :]
        idx = np.rint(p).astype(np.int64)
        inside = ((idx >= 0) & (idx < D)).all(-1)
        idx = idx[inside]
        rows[idx[:, 0], idx[:, 1], idx[:, 2] // 64] = True
    bricks = rows.reshape(D // 32, 32, D // 32, 32, D // 64).any(axis=(1, 3))       # (16, 16, 8)
    in_entered = np.broadcast_to(bricks[:, None, :, None, :], (D // 32, 32, D // 32, 32, D // 64)).reshape(D, D, D // 64)
    print(f"pose {b} (rot {rot[b].numpy().round(2)}): {rows.mean():.3f}   {rows[in_entered].mean():.3f}   "
          f"{int(bricks.sum())} / {bricks.size}", flush=True)
