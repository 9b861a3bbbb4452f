"""Timing of the two Siddon volume-gradient kernels (development tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("exp")  # experiment switches live in the tools build only
from diffdrr_amd import DRR, ops
from diffdrr_amd.data import make_subject, noise_volume
from tools.kernel_sweep import poses, rays, timeit
dev = torch.device("cuda:0")
from diffdrr_amd import _lib
if len(sys.argv) > 1:
    _lib.get_lib().cdll.ddrr_set_brick_debug(int(sys.argv[1]))
for D, H, Bs in ((512, 256, (1, 8, 32)), (256, 256, (32,))):
    drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4 * D / 512).to(dev)
    V = drr.density
    for B in Bs:
        s, t, L = rays(drr, *poses(B, 2, dev))
        go = torch.rand(B, H * H, device=dev)
        a, _ = timeit(lambda: ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (H, H)))
        b, _ = timeit(lambda: ops.siddon_backward_volume(V, s, t, L, go, det=(H, H)), reps=3, warm=1)
        print(f"{D}^3 det {H}^2 B={B:3d}: bricks {a:8.3f} ms ({B / a * 1e3:7.0f} DRR/s)   global atomics {b:8.3f} ms", flush=True)
