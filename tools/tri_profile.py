"""Where the wave-cycles of the marcher's brick kernels go at BASELINE config 3 (512^3 -> 512^2, 512 samples
per ray, one pose): forward and volume gradient, profile build (-DDDRR_BRICK_PROFILE; development tool).
Usage: python tools/tri_profile.py [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("prof")
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from diffdrr_amd.renderers import get_alpha_minmax  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
D, P, H = 512, 512, 512
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=1.2, renderer="trilinear").to(dev)
V = drr.density
s, t, L = rays(drr, *poses(B, 2, dev))
lo, hi = get_alpha_minmax(s, t, torch.tensor(V.shape, device=dev), 0.5, 1e-8)
amin, amax = lo.min().reshape(1).contiguous(), hi.max().reshape(1).contiguous()
go = torch.rand(B, H * H, device=dev)
NAMES = ["barrier+prefix", "unit pull", "phase A", "batch pop", "ray loads", "setup", "walk", "deliver",
         "barrier wait", "#batches", "#wave-steps", "#units", "#hits"]
for name, fn in (("forward", lambda: ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (H, H), n_points=P)),
                 ("volume gradient", lambda: ops.trilinear_backward_volume_bricks(V.shape, s, t, L, go, amin, amax,
                                                                                  (H, H), n_points=P))):
    med, _ = timeit(fn)
    lib.cdll.ddrr_brick_profile_reset()
    fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 20)()
    lib.cdll.ddrr_brick_profile_read20(buf)
    v = list(buf)
    tot = sum(v[:9]) + sum(v[13:16])
    print(f"## {name}, {B} pose(s): kernel {med:.3f} ms (profiling build); {tot / 4096:.0f} ticks per wave")
    for i, n in zip((13, 14, 15), ("  claim", "  rows/issue", "  LDS store")):
        print(f"  {n:14s} {100 * v[i] / tot:5.1f} %")
    for i, n in enumerate(NAMES):
        print(f"  {n:14s} {100 * v[i] / tot:5.1f} %" if i < 9 else f"  {n:14s} {v[i]}")
    print(f"  hits per batch {v[12] / max(1, v[9]):.1f}; wave-steps per batch {v[10] / max(1, v[9]):.1f}; "
          f"walk ticks per wave-step {v[6] / max(1, v[10]):.1f}; phase A ticks per unit {v[2] / max(1, v[11]):.0f}", flush=True)
