"""How well can the hand-out order of the bricks be predicted?  Measures the duration of every brick
of one launch (profiling build, bricks in id order) and evaluates candidate weights offline: the
product's (projected pixel-box area summed over the poses), an estimate of the walk's steps from
the geometry (volume x ray density x l1 norm of the direction), and their least-squares mix;
for each the makespan of the list schedule over 256 workgroups it would give, next to the bound
sum / 256 and the schedule by the measured durations themselves.
Usage: python tools/brick_weights.py [--cases pert32,pert32aux,pert128]"""
import argparse
import ctypes
import heapq
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("prof")
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="pert32,pert32aux,pert128")
a = ap.parse_args()
dev = torch.device("cuda:0")
D, H = 512, 256
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4).to(dev)
V = drr.density
lib = _lib.get_lib()
times = torch.zeros(32768, dtype=torch.int32, device=dev)
BX, BY, BZ = 32, 32, 64


def schedule(durs, n_wg=256):
    heap = [0.0] * n_wg
    heapq.heapify(heap)
    for d in durs:
        heapq.heappush(heap, heapq.heappop(heap) + d)
    return max(heap)


def features(s_, t_):
    """per brick (id order: z fastest): clipped pixel-box area and estimated walk steps, summed over poses"""
    n = [-(-D // b) for b in (BX, BY, BZ)]
    ix, iy, iz = torch.meshgrid(*[torch.arange(k, device=dev) for k in n], indexing="ij")
    lo = torch.stack([ix * BX, iy * BY, iz * BZ], -1).reshape(-1, 3).double()
    hi = lo + torch.tensor([BX, BY, BZ], device=dev).double()
    B = t_.shape[0]
    tg = t_.reshape(B, H, H, 3).double()
    t00, ei, ej = tg[:, 0, 0], (tg[:, -1, 0] - tg[:, 0, 0]) / (H - 1), (tg[:, 0, -1] - tg[:, 0, 0]) / (H - 1)
    src = s_[:, 0].double()
    nvec = torch.linalg.cross(ei, ej)                     # (B, 3)
    r = t00 - src
    corners = torch.stack([torch.where(torch.tensor([(c >> k) & 1 for k in range(3)], device=dev).bool(), hi, lo)
                           for c in range(8)], 1) - 0.5   # (nb, 8, 3)
    w = corners[None] - src[:, None, None]                # (B, nb, 8, 3)
    det = (w * nvec[:, None, None]).sum(-1)
    rxej = torch.linalg.cross(r, ej)
    rxei = torch.linalg.cross(r, ei)
    i = -(w * rxej[:, None, None]).sum(-1) / det
    j = (w * rxei[:, None, None]).sum(-1) / det
    i0u, i1u, j0u, j1u = i.amin(-1), i.amax(-1), j.amin(-1), j.amax(-1)
    i0, i1 = i0u.floor().clamp(0, H - 1), i1u.ceil().clamp(0, H - 1)
    j0, j1 = j0u.floor().clamp(0, H - 1), j1u.ceil().clamp(0, H - 1)
    area = ((i1 - i0 + 1) * (j1 - j0 + 1)) * ((i1u >= 0) & (i0u <= H - 1) & (j1u >= 0) & (j0u <= H - 1))
    frac = (area / ((i1u - i0u + 1) * (j1u - j0u + 1))).clamp(0, 1)   # part of the silhouette on the detector
    wc = (lo + hi)[None] / 2 - 0.5 - src[:, None]        # (B, nb, 3)
    wn = (wc * nvec[:, None]).sum(-1).abs()
    rn = (r * nvec).sum(-1).abs()[:, None]
    steps = (BX * BY * BZ) * rn ** 2 * wc.abs().sum(-1) / wn ** 3 * frac
    return area.sum(0).cpu().numpy(), steps.sum(0).cpu().numpy()


for case in a.cases.split(","):
    aux = case.endswith("aux")
    name = case[:-3] if aux else case
    s, t, L = rays(drr, *poses(int(name[4:]), 2, dev))
    fn = lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H), want_aux=aux, storage="q16p")  # noqa: E731
    lib.cdll.ddrr_set_brick_debug(512)  # id order: durations undisturbed by the order under test
    lib.cdll.ddrr_set_brick_times(ctypes.c_void_p(0))
    fn()
    times.zero_()
    lib.cdll.ddrr_set_brick_times(ctypes.c_void_p(times.data_ptr()))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    d = times.cpu().float().numpy()[:2048] * 0.01  # us
    lib.cdll.ddrr_set_brick_times(ctypes.c_void_p(0))
    lib.cdll.ddrr_set_brick_debug(0)
    med, _ = timeit(fn)
    area, steps = features(s, t)
    A = np.stack([area, steps, np.ones_like(area)], 1)
    coef, *_ = np.linalg.lstsq(A, d, rcond=None)
    print(f"## {case}: product launch {med * 1e3:.0f} us (profiling build); sum / 256 = {d.sum() / 256:.0f} us; "
          f"schedule by measured duration {schedule(sorted(d, reverse=True)):.0f} us; in id order {schedule(d):.0f} us")
    for label, wgt in (("pixel-box area (product)", area), ("estimated steps", steps),
                       (f"fit {coef[0]:.3g} area + {coef[1]:.3g} steps + {coef[2]:.3g}", A @ coef),
                       ("area + steps / 5", area + steps / 5), ("area + steps / 10", area + steps / 10),
                       ("area + steps / 20", area + steps / 20)):
        order = np.argsort(-wgt)
        cc = np.corrcoef(wgt, d)[0, 1]
        print(f"   {label:48s} corr {cc:.3f}   list schedule {schedule(d[order]):.0f} us", flush=True)
