"""The reference's registration loop exactly as a user writes it (notebooks/tutorials/registration.ipynb:240-316:
Registration + criterion + torch.optim.Adam, eager, no HIP graph) on the MI355X, 512^3 -> 256^2, one pose: ms per
iteration for three criteria, with the differentiable Euler render as one autograd node (DRR._render_euler_differentiable)
and as the composition it replaces (FUSED_NCC_MAX_POSES = 0), alternating in one process (development tool, GPU).
The loop is bound by the host (Python, autograd, torch's optimizer): GraphedIteration + PoseAdam are the fast way."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, Registration  # noqa: E402
from diffdrr_amd.data import synthetic_subject  # noqa: E402
from diffdrr_amd.metrics import (GradientNormalizedCrossCorrelation2d, MultiscaleNormalizedCrossCorrelation2d,  # noqa: E402
                                 NormalizedCrossCorrelation2d)

dev = torch.device("cuda:0")
drr = DRR(synthetic_subject(512, kind="phantom", seed=0), sdd=1020.0, height=256, delx=2.4).to(dev)
true_rot = torch.zeros(1, 3, device=dev)
true_xyz = torch.tensor([[0.0, 850.0, 0.0]], device=dev)
with torch.no_grad():
    gt = drr(true_rot, true_xyz, parameterization="euler_angles", convention="ZXY")
g = torch.Generator().manual_seed(1)
r0 = true_rot + ((torch.rand(1, 3, generator=g) - 0.5) * 0.4).to(dev)
x0 = true_xyz + ((torch.rand(1, 3, generator=g) - 0.5) * 60.0).to(dev)
gc.collect()
gc.disable()
CRITERIA = (("NCC", NormalizedCrossCorrelation2d), ("Multiscale([None, 9])", lambda: MultiscaleNormalizedCrossCorrelation2d([None, 9], [0.5, 0.5])),
            ("GradientNCC()", GradientNormalizedCrossCorrelation2d))
for fused_adam in (False, True):
    for cap in (32, 0, 32, 0):
        DRR.FUSED_NCC_MAX_POSES = cap
        row = []
        for name, make in CRITERIA:
            crit = make()
            reg = Registration(drr, r0.clone(), x0.clone(), parameterization="euler_angles", convention="ZXY")
            opt = torch.optim.Adam([{"params": [reg._rotation], "lr": 1e-2}, {"params": [reg._translation], "lr": 1e0}],
                                   maximize=True, fused=fused_adam)

            def it():
                opt.zero_grad()
                loss = crit(gt, reg())
                loss.backward()
                opt.step()
                return loss
            for _ in range(100):
                it()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                last = it()
            torch.cuda.synchronize()
            row.append(f"{name} {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms (value {float(last.detach()):.4f})")
        print(f"torch.optim.Adam(fused={fused_adam})  render as {'one node' if cap else 'composition'}:  " + "   ".join(row), flush=True)
