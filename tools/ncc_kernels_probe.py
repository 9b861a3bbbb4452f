"""The image-space similarity kernels under rocprofv3 (development tool, GPU): 30 forward + backward calls each of
GradientNormalizedCrossCorrelation2d(sigma = 1) and NormalizedCrossCorrelation2d(patch_size = 13, 15, 9, 10) on 32
pairs of 256 x 256 -- 13 and 9 are instantiated window sizes, 15 and 10 take the run-time loop.
    rocprofv3 --kernel-trace --stats -d gpurun_out/ncc_kernels -o p -- python tools/ncc_kernels_probe.py
(profiles/r06/gradient_ncc.txt, patch_ncc.txt)."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import metrics as M  # noqa: E402

gc.collect()
gc.disable()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = 32
fixed = (torch.rand(1, 1, 256, 256, generator=g) * 50).to(dev)
moving = (torch.rand(B, 1, 256, 256, generator=g) * 50).to(dev)
for c in (M.GradientNormalizedCrossCorrelation2d(sigma=1.0), M.NormalizedCrossCorrelation2d(patch_size=13),
          M.NormalizedCrossCorrelation2d(patch_size=15), M.NormalizedCrossCorrelation2d(patch_size=9),
          M.NormalizedCrossCorrelation2d(patch_size=10)):
    x = moving.clone().requires_grad_()
    for _ in range(30):
        x.grad = None
        c(fixed.expand(B, -1, -1, -1), x).sum().backward()
torch.cuda.synchronize()
