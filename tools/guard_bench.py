"""What the fp32 fallback of the 16-bit brick storages costs: 512^3 -> 256^2, 32 poses, forward and
forward + record, storage "f32" against "q16p" on volumes with none / some / most of their bricks on
the fp32 path (csrc/brick_step.h q16_usable).  Product library.
Usage: python tools/guard_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume, phantom_volume  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

dev = torch.device("cuda:0")
D, H = 512, 256
noise = noise_volume(D, 0)
vols = {"noise U[0,1)": noise}
metal = noise.clone()
g = torch.Generator().manual_seed(3)
for _ in range(40):  # 40 bright markers: their bricks leave the quantised path
    c = (torch.rand(3, generator=g) * (D - 8)).long() + 4
    metal[c[0] - 2:c[0] + 2, c[1] - 2:c[1] + 2, c[2] - 2:c[2] + 2] = 200.0
vols["noise + 40 markers at 200x"] = metal
vols["phantom (smooth ellipsoids + 1 % noise)"] = phantom_volume(D, 0)
sparse = noise * (torch.rand(D, D, D, generator=g) < 0.02)  # 98 % exact zeros: level of the non-zero voxels
vols["noise, 98 % of the voxels zero"] = sparse
hu = (noise - 0.5) * 2000.0
vols["zero-mean values +-1000"] = hu
drr = DRR(make_subject(noise), sdd=1020.0, height=H, delx=2.4).to(dev)
s, t, L = rays(drr, *poses(32, 2, dev))
print(f"# {torch.cuda.get_device_name(0)}: {D}^3 -> {H}^2, 32 poses; kernel ms by back-to-back launches")
for name, v in vols.items():
    V = v.to(dev).contiguous()
    row = []
    for st in ("f32", "q16p"):
        f, _ = timeit(lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H), storage=st))
        a, _ = timeit(lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H), storage=st, want_aux=True))
        row.append((f, a))
    fb = ops.brick_fallbacks(V, "q16p")
    o32 = ops.siddon_forward_bricks(V, s[:4], t[:4], L[:4], (H, H), storage="f32")[0]
    o16 = ops.siddon_forward_bricks(V, s[:4], t[:4], L[:4], (H, H), storage="q16p")[0]
    err = float((o16 - o32).abs().max() / o32.abs().max())
    print(f"{name:42s} bricks on the fp32 path {fb[0]:5d} / {fb[1]} | forward f32 {row[0][0]:6.3f} q16p {row[1][0]:6.3f} ms "
          f"| forward + record f32 {row[0][1]:6.3f} q16p {row[1][1]:6.3f} ms | q16p vs f32 image {err:.1e}", flush=True)
    del V
