"""Run one fixed launch configuration a few times (target for rocprofv3 runs)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("exp")  # experiment switches live in the tools build only
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from tools.kernel_sweep import poses, rays  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="pert32")
ap.add_argument("--tile", default="16x4")
ap.add_argument("--xcd", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--kernel", default="generic", choices=["generic", "brick", "volgrad", "trifwd", "trivol"])
ap.add_argument("--det", type=int, default=256)
ap.add_argument("--aux", type=int, default=0)
ap.add_argument("--storage", default="q16p", choices=["q16p", "q16", "f32"])
a = ap.parse_args()
dev = torch.device("cuda:0")
D, H = a.size, a.det
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4 * (256 / H) * D / 512).to(dev)
_lib.get_lib().cdll.ddrr_set_xcd_swizzle(a.xcd)
if a.case.startswith("base"):
    B = int(a.case[4:])
    one = rays(drr, torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 850.0, 0.0]], device=dev))
    s, t, L = (x.expand(B, *x.shape[1:]).contiguous() for x in one)
elif a.case.startswith("pert"):
    s, t, L = rays(drr, *poses(int(a.case[4:]), 2, dev))
elif a.case.startswith("same"):  # one perturbed pose replicated
    B = int(a.case[4:])
    one = tuple(x[4:5] for x in rays(drr, *poses(8, 2, dev)))
    s, t, L = (x.expand(B, *x.shape[1:]).contiguous() for x in one)
th, tw = (int(v) for v in a.tile.split("x"))
if a.kernel in ("trifwd", "trivol"):  # config 3's launches: the marcher on the bricks, 512 samples per ray
    from diffdrr_amd.renderers import get_alpha_minmax

    lo, hi = get_alpha_minmax(s, t, torch.tensor(drr.density.shape, device=dev), 0.5, 1e-8)
    amin, amax = lo.min().reshape(1).contiguous(), hi.max().reshape(1).contiguous()
    go = torch.rand(L.shape, device=dev)
for _ in range(a.reps):
    if a.kernel == "trifwd":
        ops.trilinear_forward_bricks(drr.density, s, t, L, amin, amax, (H, H), n_points=512)
    elif a.kernel == "trivol":
        ops.trilinear_backward_volume_bricks(drr.density.shape, s, t, L, go, amin, amax, (H, H), n_points=512)
    elif a.kernel == "generic":
        ops.siddon_forward(drr.density, s, t, L, det=(H, H), tile=(th, tw))
    elif a.kernel == "volgrad":
        ops.siddon_backward_volume_bricks(drr.density.shape, s, t, L, torch.ones_like(L), (H, H))
    else:
        ops.siddon_forward_bricks(drr.density, s, t, L, (H, H), want_aux=bool(a.aux), storage=a.storage)
torch.cuda.synchronize()
