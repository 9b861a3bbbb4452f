"""`p_subsample` / `patch_size` (reference drr.py:36-39, 142-147, 218-225) on the volume-stationary
kernels against the per-ray kernels they took until round 5 (development tool, GPU).

The reference's example geometry (README.md:67-87): a CT-like 512 x 512 x 133 volume -> 200 x 200,
p_subsample = 0.1 (the reference publishes 5.15 ms for it on an RTX 2080 Ti,
notebooks/tutorials/introduction.ipynb:611) at 1 and 8 poses, forward (no grad) and forward +
backward to the pose; and the timing notebook's patched renders (timing.ipynb:182-254: 500^2 / 250,
750^2 / 150, 1000^2 / 250) for both renderers.  Module-level wall time per call (HIP events
around 50 calls after 20 warm-up calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR  # noqa: E402
from diffdrr_amd.data import ct_like_hu_volume, make_subject, transform_hu_to_density  # noqa: E402
from bench import perturbed_poses  # noqa: E402

dev = torch.device("cuda:0")
density = transform_hu_to_density(ct_like_hu_volume((512, 512, 133), seed=0))
subject = make_subject(density, spacing=(0.703, 0.703, 2.5), orientation="AP")


def timed(fn, warm=20, n=50):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(drr, B, kw, grad):
    rot0, xyz0 = perturbed_poses(B, seed=2, device=dev)
    if not grad:
        def fn():
            with torch.no_grad():
                drr(rot0, xyz0, parameterization="euler_angles", convention="ZXY", **kw)
    else:
        rot, xyz = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()

        def fn():
            rot.grad = xyz.grad = None
            drr(rot, xyz, parameterization="euler_angles", convention="ZXY", **kw).sum().backward()
    return timed(fn)


print("== p_subsample = 0.1, 512 x 512 x 133 -> 200 x 200 (ms per call; per-ray kernels -> on the bricks)")
for renderer, kw in (("siddon", {}), ("trilinear", {"n_points": 200})):
    torch.manual_seed(0)
    sub = DRR(subject, sdd=1020.0, height=200, delx=2.0, renderer=renderer, p_subsample=0.1).to(dev)
    dense = DRR(subject, sdd=1020.0, height=200, delx=2.0, renderer=renderer).to(dev)
    for B in (1, 8):
        for grad in (False, True):
            sub.fuse_ray_generation = False
            old = run(sub, B, kw, grad)
            sub.fuse_ray_generation = True
            new = run(sub, B, kw, grad)
            full = run(dense, B, kw, grad)
            print(f"{renderer:9s} B={B} {'fwd+bwd' if grad else 'forward'}: per-ray {old:7.3f} -> bricks {new:7.3f} ms"
                  f"   (the dense 200 x 200 render: {full:7.3f} ms)", flush=True)

print("== patch_size (timing.ipynb:182-254), one pose, forward (ms per call; patch loop on per-ray kernels -> bricks)")
for renderer, kw in (("siddon", {}), ("trilinear", {"n_points": 200})):
    for H, ps in ((500, 250), (750, 150), (1000, 250)):
        pat = DRR(subject, sdd=1020.0, height=H, delx=2.0 * 200 / H, renderer=renderer, patch_size=ps).to(dev)
        pat.fuse_ray_generation = False
        old = run(pat, 1, kw, False)
        pat.fuse_ray_generation = True
        new = run(pat, 1, kw, False)
        print(f"{renderer:9s} {H}^2 / patch {ps} ({pat.n_patches} chunks): per-ray {old:8.3f} -> bricks {new:8.3f} ms",
              flush=True)
