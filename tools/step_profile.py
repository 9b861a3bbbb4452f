"""Where one bench step spends its time: torch.profiler over a few steps of bench.py's
workload (development tool).  Prints GPU kernels by total time and the step wall time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import perturbed_poses  # noqa: E402
from diffdrr_amd import DRR, NormalizedCrossCorrelation2d  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402

dev = torch.device("cuda:0")
D, H, B = 512, 256, 32
drr = DRR(make_subject(noise_volume(D, seed=0)), sdd=1020.0, height=H, delx=2.4).to(dev)
ncc = NormalizedCrossCorrelation2d()
rot0, xyz0 = perturbed_poses(B, 2, dev)
with torch.no_grad():
    base = drr(torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 850.0, 0.0]], device=dev),
               parameterization="euler_angles", convention="ZXY")
rot = rot0.clone().requires_grad_()
xyz = xyz0.clone().requires_grad_()


def step():
    rot.grad = None
    xyz.grad = None
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    loss = ncc(base.expand(B, -1, -1, -1), img)
    loss.sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"step wall {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0]
ev.sort(key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in ev)
print(f"GPU kernel time per step {tot / 5 / 1e3:.3f} ms in {sum(e.count for e in ev) / 5:.0f} launches")
for e in ev[:25]:
    print(f"{e.device_time_total / 5 / 1e3:8.3f} ms  x{e.count / 5:5.1f}  {e.key[:100]}")
