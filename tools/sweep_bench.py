"""BASELINE.json config 5 on ONE GPU: the per-GPU share of the 4096-pose sweep
(512 candidate poses, 512^3 volume, 256^2 detector): forward + per-pose NCC through
diffdrr_amd.dist.sweep (no process group = world size 1).  Prints DRRs/s."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import perturbed_poses  # noqa: E402
from diffdrr_amd import DRR, NormalizedCrossCorrelation2d  # noqa: E402
from diffdrr_amd import dist as ddist  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402

dev = torch.device("cuda:0")
drr = DRR(make_subject(noise_volume(512, seed=0)), sdd=1020.0, height=256, delx=2.4).to(dev)
ncc = NormalizedCrossCorrelation2d()
with torch.no_grad():
    fixed = drr(torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 850.0, 0.0]], device=dev),
                parameterization="euler_angles", convention="ZXY")
for P, chunk in ((512, 128), (512, 256), (512, 512)):
    rot, xyz = perturbed_poses(P, 2, dev)
    ddist.sweep(drr, ncc, fixed, rot[:chunk], xyz[:chunk], chunk=chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vals = ddist.sweep(drr, ncc, fixed, rot, xyz, chunk=chunk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"sweep of {P} poses, {chunk} per launch: {dt * 1e3:.1f} ms = {P / dt:.0f} DRRs/s "
          f"(best NCC {vals.max().item():.4f})", flush=True)
