"""The marcher's volume gradient at config 3 with integer LDS atomics (product), plain LDS stores (what the atomics cost;
garbage result) and float LDS atomics: tools build (development tool).  Usage: python tools/volgrad_noatomic.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib
tools.explib.use("exp")
from diffdrr_amd import DRR, ops, _lib
from diffdrr_amd.data import make_subject, noise_volume
from diffdrr_amd.renderers import get_alpha_minmax
from tools.kernel_sweep import poses, rays, timeit
dev = torch.device("cuda:0")
D, P, H = 512, 512, 512
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=1.2, renderer="trilinear").to(dev)
V = drr.density
for B in (1, 4):
    s, t, L = rays(drr, *poses(B, 2, dev))
    lo, hi = get_alpha_minmax(s, t, torch.tensor(V.shape, device=dev), 0.5, 1e-8)
    amin, amax = lo.min().reshape(1).contiguous(), hi.max().reshape(1).contiguous()
    go = torch.rand(B, H * H, device=dev)
    for dbg, label in ((0, "integer LDS atomics (product)"), (1 << 23, "plain LDS stores instead (garbage result)"), (32, "float LDS atomics")):
        _lib.get_lib().cdll.ddrr_set_brick_debug(dbg)
        a, _ = timeit(lambda: ops.trilinear_backward_volume_bricks(V.shape, s, t, L, go, amin, amax, (H, H), n_points=P))
        print(f"volume gradient, {B} pose(s), {label}: {a:.3f} ms", flush=True)
