"""Stage stamps of the marcher's brick kernels at BASELINE config 3 (512^3 -> 512^2, 512 samples per ray): where a
brick's time goes from its claim to its end, mean over the bricks (profile build; development tool).
Usage: python tools/tri_stamps.py [B]"""
import ctypes
import os
import sys

import numpy as np
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
NAMES = ("claimed", "wave 0 staged", "last wave staged", "behind staging barrier", "wave 0 out of units",
         "last wave out of units", "wave 0 walks done", "last walk done", "behind gradient barrier", "wave 0 stored",
         "last wave stored")
for name, nb, fn in (("forward", 17 ** 3, lambda: ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (H, H), n_points=P)),
                     ("volume gradient", 16 ** 3 * 4, lambda: ops.trilinear_backward_volume_bricks(
                         V.shape, s, t, L, go, amin, amax, (H, H), n_points=P))):
    nb = 17 ** 3 if name == "forward" else 16 ** 3
    times = torch.zeros(nb * 17, dtype=torch.int32, device=dev)
    lib.cdll.ddrr_set_brick_times(ctypes.c_void_p(0))
    med, _ = timeit(fn)
    lib.cdll.ddrr_set_brick_times(ctypes.c_void_p(times.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.cdll.ddrr_set_brick_times(ctypes.c_void_p(0))
    tr = times.cpu().float().numpy()[nb:nb + 16 * nb].reshape(nb, 16) * 0.01
    live = tr[:, 0] > 0
    tr = tr[live]
    end = tr[:, :11].max(axis=1)
    print(f"## {name}, {B} pose(s): kernel {med * 1e3:.0f} us (profile build, stamps off), {live.sum()} bricks, "
          f"{live.sum() / 256:.1f} per workgroup; a brick ends {end.mean():.1f} us after its claim was asked for "
          f"(p5 {np.percentile(end, 5):.1f}, p50 {np.percentile(end, 50):.1f}, p95 {np.percentile(end, 95):.1f})")
    print("   us from the brick's start, mean: " + ", ".join(f"{n} {tr[:, k].mean():.1f}" for k, n in enumerate(NAMES)),
          flush=True)
