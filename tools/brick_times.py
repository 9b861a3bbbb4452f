"""Per-brick durations of one forward (+ record) launch (profiling build) and what a schedule of
them over the persistent workgroups can achieve: the measured launch, the bound sum / workgroups,
the longest brick, and list schedules in id order / by decreasing measured duration.
Usage: python tools/brick_times.py [--cases pert32,pert32aux] [--variant 5]"""
import argparse
import ctypes
import heapq
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("prof")
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="pert32,pert32aux")
ap.add_argument("--variant", type=int, default=5)
ap.add_argument("--storage", default=None, help="f32 | q16 | q16p (default: by variant)")
a = ap.parse_args()
dev = torch.device("cuda:0")
D, H = 512, 256
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4).to(dev)
V = drr.density
lib = _lib.get_lib()
lib.cdll.ddrr_set_brick_variant(a.variant)
storage = a.storage or ("q16" if a.variant in (1, 2, 4, 5, 6, 10) else "f32")
times = torch.zeros(2048 * 17, dtype=torch.int32, device=dev)


def schedule(durs, n_wg=256):
    heap = [0.0] * n_wg
    heapq.heapify(heap)
    for d in durs:
        heapq.heappush(heap, heapq.heappop(heap) + d)
    return max(heap)


for case in a.cases.split(","):
    aux = case.endswith("aux")
    name = case[:-3] if aux else case
    s, t, L = rays(drr, *poses(int(name[4:]), 2, dev))
    fn = lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H), want_aux=aux, storage=storage)  # noqa: E731
    for dbg, label in ((512, "bricks in id order"), (0, "heaviest projected area first (product)")):
        lib.cdll.ddrr_set_brick_debug(dbg)
        lib.cdll.ddrr_set_brick_times(ctypes.c_void_p(0))
        med, _ = timeit(fn)
        times.zero_()
        lib.cdll.ddrr_set_brick_times(ctypes.c_void_p(times.data_ptr()))
        fn()
        torch.cuda.synchronize()
        raw = times.cpu().float().numpy() * 0.01  # us
        d = raw[:2048]
        live = d > 0
        tr = raw[2048:2048 + 16 * 2048].reshape(2048, 16)[live]
        d = d[live]
        import numpy as np
        print(f"## {case}, {label}: launch {med * 1e3:.0f} us (median, profiling build), {len(d)} bricks: sum / 256 = "
              f"{d.sum() / 256:.0f} us, longest {d.max():.0f} us, mean {d.mean():.0f} us, p99 {np.percentile(d, 99):.0f} us | "
              f"list schedule in id order {schedule(d):.0f} us, by decreasing duration {schedule(sorted(d, reverse=True)):.0f} us",
              flush=True)
        names = ("staged", "at pool barrier", "behind pool barrier", "wave 0 out of work",
                 "wave 0 at the staging barrier", "last wave out of work", "last walk done",
                 "last wave at the staging barrier", "wave 0 image stored", "wave 0 rows written",
                 "last wave image stored")
        q = np.percentile(d, [5, 25, 50, 75, 95])
        print("   durations: p5 %.1f p25 %.1f p50 %.1f p75 %.1f p95 %.1f us; the quarter of shortest bricks: "
              % tuple(q) + ", ".join(f"{n} {tr[d <= q[1]][:, k].mean():.1f}" for k, n in enumerate(names)), flush=True)
        print("   stages (us from the brick's start, mean over the bricks): "
              + ", ".join(f"{n} {tr[:, k].mean():.1f}" for k, n in enumerate(names)), flush=True)
