"""Where the wave time of the channel render on the bricks goes, next to the plain render of the same
kernel (bricks.hip siddon_brick_kernel: 512 x 512 x 133 takes the general 32^3 fp32 path): phase
profile of the profiling build (s_memtime per phase and wave).
Usage: python tools/channels_profile.py [--real-mask] [--poses 8]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("prof")
from diffdrr_amd import DRR, _lib, convert, ops  # noqa: E402
from diffdrr_amd.data import make_subject  # noqa: E402
from diffdrr_amd.renderers import _labels_u8  # noqa: E402
from tools.kernel_sweep import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--real-mask", action="store_true")
ap.add_argument("--poses", type=int, default=8)
ap.add_argument("--dbg", default="0")
a = ap.parse_args()
dev = torch.device("cuda:0")
dims, C, H = (512, 512, 133), 119, 200
g = torch.Generator().manual_seed(0)
vol = torch.rand(*dims, generator=g)
if a.real_mask:
    fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                              "reference_mask_ds2.npz"))
    mask = torch.from_numpy(fx["labels"]).repeat_interleave(2, 0).repeat_interleave(2, 1)[:512, :512]
else:
    coarse = torch.randint(0, C, (16, 16, 8), generator=g)
    mask = coarse
    for ax, d in enumerate(dims):
        idx = (torch.arange(d) * coarse.shape[ax] // d).clamp_max(coarse.shape[ax] - 1)
        mask = mask.index_select(ax, idx)
    mask[0, 0, 0] = C - 1
drr = DRR(make_subject(vol, spacing=(0.703, 0.703, 2.5), mask=mask), sdd=1020.0, height=H, delx=2.0).to(dev)
(labels, _, _), = _labels_u8(drr.mask)
B = a.poses
rot = torch.zeros(B, 3, device=dev) + torch.linspace(0, 0.3, B, device=dev)[:, None]
xyz = torch.tensor([[0.0, 850.0, 0.0]], device=dev).expand(B, 3).contiguous()
with torch.no_grad():
    pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
    source, target = drr.detector(pose, None)
    L = (target - source).norm(dim=-1).contiguous()
    s_, t_ = drr.affine_inverse(source).contiguous(), drr.affine_inverse(target).contiguous()
lib = _lib.get_lib()
NAMES = ["barrier+prefix", "unit pull", "phase A", "batch pop", "ray loads", "setup", "walk", "deliver",
         "barrier wait", "#batches", "#wave-steps", "#units", "#hits"]
for dbg in a.dbg.split(","):
    lib.cdll.ddrr_set_brick_debug(int(dbg))
    for name, fn in (("plain", lambda: ops.siddon_forward_bricks(drr.density, s_, t_, L, (H, H))),
                     ("channels", lambda: ops.siddon_forward_channels_bricks(drr.density, labels, C, s_, t_, L, (H, H)))):
        med, _ = timeit(fn)
        lib.cdll.ddrr_brick_profile_reset()
        fn()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        lib.cdll.ddrr_brick_profile_read(buf)
        v = list(buf)
        tot = sum(v[:9]) + sum(v[13:16])
        print(f"## dbg {dbg} {name} B {B}: {med:.3f} ms (profiling build); {tot / 4096:.0f} ticks per wave")
        for i, n in zip((13, 14, 15), ("  claim", "  rows/issue", "  LDS store")):
            print(f"  {n:14s} {100 * v[i] / tot:5.1f} %  ({v[i] / 4096:.0f} ticks per wave)")
        for i, n in enumerate(NAMES):
            if i < 9:
                print(f"  {n:14s} {100 * v[i] / tot:5.1f} %  ({v[i] / 4096:.0f} ticks per wave)")
            else:
                print(f"  {n:14s} {v[i]}")
