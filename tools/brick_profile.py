"""Where the wave-cycles of the Siddon brick kernel go: builds the library with
-DDDRR_BRICK_PROFILE (s_memtime deltas per phase, per wave; tools only, never the product
library) and prints the phase totals for a launch.
Usage: python tools/brick_profile.py [--cases pert32,pert32aux] [--size 512]"""
import argparse
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--det", type=int, default=256)
ap.add_argument("--cases", default="pert32,pert32aux")
ap.add_argument("--build-only", action="store_true")
ap.add_argument("--variants", default="-2", help="bricks_fwd.hip variants (see tools/brick_bench.py)")
ap.add_argument("--storage", default=None, help="brick storage handed to the entry point (with --variants -2: f32 | q16 | q16p)")
a = ap.parse_args()

import tools.explib  # noqa: E402

tools.explib.use("prof")
if a.build_only:
    sys.exit(0)
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
D, H = a.size, a.det
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4 * (256 / H) * (D / 512)).to(dev)
V = drr.density
NAMES = ["barrier+prefix", "unit pull", "phase A", "batch pop", "ray loads", "setup", "walk", "deliver",
         "barrier wait", "#batches", "#wave-steps", "#units", "#hits"]
import itertools  # noqa: E402

for var, case in itertools.product(a.variants.split(","), a.cases.split(",")):
    var = int(var)
    lib.cdll.ddrr_set_brick_variant(var)
    storage = a.storage or ("q16" if var >= 0 and var % 16 in (1, 2, 4, 5, 6, 10) else "f32")
    aux = case.endswith("aux")
    name = case[:-3] if aux else case
    if name.startswith("base"):
        B = int(name[4:])
        one = rays(drr, torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 850.0, 0.0]], device=dev))
        s, t, L = (x.expand(B, *x.shape[1:]).contiguous() for x in one)
    else:
        s, t, L = rays(drr, *poses(int(name[4:]), 2, dev))
    fn = lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H), want_aux=aux, storage=storage)  # noqa: E731
    med, best = timeit(fn)
    lib.cdll.ddrr_brick_profile_reset()
    fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 20)()
    lib.cdll.ddrr_brick_profile_read20(buf)
    v = list(buf)
    tot = sum(v[:9]) + sum(v[13:16])
    waves = 4096  # (every variant runs 4096 waves: 256 x 16 or 512 x 8)
    print(f"## variant {var} {case}: kernel {med:.3f} ms (profiling build); {tot / 4096:.0f} ticks per wave")
    for i, n in zip((13, 14, 15), ("  claim", "  rows/issue", "  LDS store")):
        print(f"  {n:14s} {100 * v[i] / tot:5.1f} %")
    for i, n in enumerate(NAMES):
        if i < 9:
            print(f"  {n:14s} {100 * v[i] / tot:5.1f} %")
        else:
            print(f"  {n:14s} {v[i]}")
    if v[19]:
        first = (~v[18]) & (2 ** 64 - 1)
        span = v[19] - first
        print(f"  launch span {span / 100:.1f} us (first wave start -> last wave end, 100 MHz clock); waves start on average "
              f"{100 * (v[16] / 4096 - first) / span:.1f} % of it after the first and end {100 * (v[19] - v[17] / 4096) / span:.1f} % "
              f"of it before the last")
    print(f"  hits per batch {v[12] / max(1, v[9]):.1f}; wave-steps per batch {v[10] / max(1, v[9]):.1f}; "
          f"walk ticks per wave-step {v[6] / max(1, v[10]):.1f}; setup ticks per batch {v[5] / max(1, v[9]):.0f}; "
          f"load ticks per batch {v[4] / max(1, v[9]):.0f}; phase A ticks per unit {v[2] / max(1, v[11]):.0f}; "
          f"deliver ticks per batch {v[7] / max(1, v[9]):.0f}", flush=True)
