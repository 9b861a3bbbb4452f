"""Which PyTorch ops launch the small kernels of a headline step (development tool, GPU):
torch.profiler over a few steps, aten ops with device time and calls per step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, NormalizedCrossCorrelation2d
from diffdrr_amd.data import make_subject, noise_volume
dev = torch.device("cuda:0")
D, H, B = 512, 256, 32
drr = DRR(make_subject(noise_volume(D, seed=0), spacing=(1.0, 1.0, 1.0), orientation="AP"), sdd=1020.0, height=H,
          delx=2.4, renderer="siddon").to(dev)
g = torch.Generator().manual_seed(2)
rot = ((torch.rand(B, 3, generator=g) - 0.5) * 1.5).to(dev).requires_grad_()
xyz = (torch.tensor([0.0, 850.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 60).to(dev).requires_grad_()
ncc = NormalizedCrossCorrelation2d()
with torch.no_grad():
    base = drr(torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 850.0, 0.0]], device=dev),
               parameterization="euler_angles", convention="ZXY")
def step():
    rot.grad = None; xyz.grad = None
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    loss = ncc(base.expand(B, -1, -1, -1), img)
    loss.sum().backward()
for _ in range(10): step()
torch.cuda.synchronize()
N = 5
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
rows = [(e.key, e.count / N, e.device_time_total / N) for e in prof.key_averages() if e.device_time_total > 0]
for k, c, t in sorted(rows, key=lambda r: -r[2])[:40]:
    print(f"{t:9.1f} us/step  x{c:5.1f}  {k[:100]}")
