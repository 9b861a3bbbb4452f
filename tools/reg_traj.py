"""Registration trajectory of the reference fixture replayed on the GPU (diagnostic)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden, rel_err
from diffdrr_amd import DRR, Registration
from diffdrr_amd.data import Image, Subject
from diffdrr_amd.metrics import NormalizedCrossCorrelation2d
from test_gpu_parity import _geo
gpu = torch.device("cuda:0")
g = golden("registration")
T = torch.from_numpy
vol = T(g["volume"])
print("volume", vol.shape, "geo", _geo(g))
subject = Subject(Image(vol.unsqueeze(0), g["affine"]), Image(vol.unsqueeze(0), g["affine"]), T(g["reorient"]))
for path in ("bricks", "generic"):
  for fused in (True, False):
    drr = DRR(subject, **_geo(g)).to(gpu)
    drr.renderer.grid_path = path
    drr.fuse_ray_generation = fused
    gt = T(g["gt"]).to(gpu)
    reg = Registration(drr, T(g["rot0"]).clone().to(gpu), T(g["xyz0"]).clone().to(gpu), parameterization="euler_angles", convention="ZXY")
    crit = NormalizedCrossCorrelation2d()
    opt = torch.optim.SGD([{"params": [reg._rotation], "lr": 5e-2}, {"params": [reg._translation], "lr": 1e2}], maximize=True)
    print(f"== path {path} fused {fused}")
    for k in range(len(g["losses_full"])):
        opt.zero_grad()
        loss = crit(gt, reg()).mean()
        loss.backward()
        print(k, f"loss {loss.item():.6f} ref {g['losses_full'][k]:.6f} | rot {reg._rotation.detach().cpu().numpy().round(5)} ref {g['rots_full'][k].round(5)} | xyz {reg._translation.detach().cpu().numpy().round(3)} ref {g['xyzs_full'][k].round(3)} | grot {reg._rotation.grad.cpu().numpy().round(4)} gxyz {reg._translation.grad.cpu().numpy().round(6)}")
        opt.step()
