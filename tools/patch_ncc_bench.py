"""Patch-wise / multiscale / gradient NCC (reference metrics.py:16-107) with the fused window kernels against the
`to_patches` composition they replace (development tool, GPU): 256 x 256 images, forward + backward w.r.t. the moving
image, ms per call."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import metrics as M  # noqa: E402

dev = torch.device("cuda:0")
gc.collect()
gc.disable()  # (a generation-2 pass inside a 20-call timed region reads as +2 ms per call)


def timed(fn, warm=10, n=20):
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


def unfused(crit):
    """the same module with the window kernels switched off (the reference's own composition)"""
    import copy
    c = copy.deepcopy(crit)
    for m in [c] + list(getattr(c, "nccs", [])):
        m._no_patch_kernel = True
    if hasattr(c, "sobel"):
        c.sobel._no_blur_kernel = True
    return c


g = torch.Generator().manual_seed(0)
CASES = (("NCC(patch 13)", lambda: M.NormalizedCrossCorrelation2d(patch_size=13)),
         ("Multiscale([13, None], [0.5, 0.5])", lambda: M.MultiscaleNormalizedCrossCorrelation2d([13, None], [0.5, 0.5])),
         ("GradientNCC(patch 9, sigma 1)", lambda: M.GradientNormalizedCrossCorrelation2d(patch_size=9, sigma=1.0)),
         ("GradientNCC(sigma 1)", lambda: M.GradientNormalizedCrossCorrelation2d(sigma=1.0)))
for B in (1, 8, 32):
    fixed = (torch.rand(1, 1, 256, 256, generator=g) * 50).to(dev)
    moving = (torch.rand(B, 1, 256, 256, generator=g) * 50).to(dev)
    rows = {}
    # (every fused criterion first, then the compositions: those leave gigabytes in the caching allocator)
    for which in ("fused", "composition"):
        for name, make in CASES:
            c = make() if which == "fused" else unfused(make())
            x = moving.clone().requires_grad_()

            def fn():
                x.grad = None
                c(fixed.expand(B, -1, -1, -1), x).sum().backward()
            try:
                ms = timed(fn)
                vals = c(fixed.expand(B, -1, -1, -1), x).detach()[:2].tolist()
            except torch.OutOfMemoryError:
                ms, vals = float("nan"), None
            rows.setdefault(name, {})[which] = (ms, vals)
            del x
        torch.cuda.empty_cache()
    for name, r in rows.items():
        print(f"B = {B:2d}  {name:38s} fused {r['fused'][0]:8.3f} ms   composition {r['composition'][0]:8.3f} ms   "
              f"values {r['fused'][1]} / {r['composition'][1]}", flush=True)
