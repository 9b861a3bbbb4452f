"""What ONE module call costs at a pose or two (product library): `drr(rot, xyz, parameterization="euler_angles")`
under no_grad -- the reference's everyday call -- wall time per call in a tight loop (the host launches ahead of the
GPU) against the brick kernel's own time, 512^3 -> 256^2, noise volume and phantom (development tool).
Usage: python tools/module_call_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume, phantom_volume  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

dev = torch.device("cuda:0")
for name, vol in (("noise", noise_volume(512, 0)), ("phantom", phantom_volume(512, 0)), ("noise again", noise_volume(512, 0))):
    drr = DRR(make_subject(vol), sdd=1020.0, height=256, delx=2.4).to(dev)
    for B in (1, 4):
        rot, xyz = poses(B, 2, dev)
        s, t, L = rays(drr, rot, xyz)
        with torch.no_grad():
            call = lambda: drr(rot, xyz, parameterization="euler_angles", convention="ZXY")  # noqa: E731
            for _ in range(400):  # (the GPU's clocks drop while the host builds a volume)
                call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 300
            for _ in range(n):
                call()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e3
            k, _ = timeit(lambda: ops.siddon_forward_bricks(drr.density, s, t, L, (256, 256), storage="q16p"))
        print(f"{name:11s} B {B}: module call {wall:.4f} ms wall per call ({B / wall * 1e3:.0f} DRRs/s) | brick entry alone "
              f"{k:.4f} ms", flush=True)
