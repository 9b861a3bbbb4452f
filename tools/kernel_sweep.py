"""Kernel-only timing sweep on one MI355X (development tool, not the bench).

Times the C-ABI launches directly (HIP events, median of several launches) for
the Siddon forward kernel over wave-tile shapes, XCD mapping, batch size and
pose sets, plus the other kernels once, and prints algorithmic GB/s
(SURVEY.md section 8d definition).  Usage: python tools/kernel_sweep.py [--size 512] [--det 256]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

if __name__ == "__main__":
    # the experiment switches of main() live in the tools build only; importing this module for
    # its helpers (poses, rays, timeit) leaves the caller on the product library
    tools.explib.use("exp")
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from diffdrr_amd.pose import convert  # noqa: E402


def poses(B, seed, dev, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    rot = (torch.rand(B, 3, generator=g) - 0.5) * (math.pi / 2) * spread
    xyz = torch.tensor([0.0, 850.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 60.0 * spread
    return rot.to(dev), xyz.to(dev)


def rays(drr, rot, xyz):
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        s, t = drr.detector(pose, None)
        L = (t - s).norm(dim=-1).contiguous()
        return drr.affine_inverse(s).contiguous(), drr.affine_inverse(t).contiguous(), L


def timeit(fn, reps=7, warm=2, group=None):
    """-> (median, best) milliseconds per call.  Calls are timed in back-to-back groups between
    two events, after a warm-up long enough for the clocks of an idle board (a launch timed alone
    after a synchronise runs up to 8 % slower: the first tool timings of a process were)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1), 1e-3)
    if group is None:
        group = int(min(20, max(1, 5.0 / one)))          # ~5 ms per group
    for _ in range(max(warm, int(min(200, 60.0 / one)))):  # ~60 ms of warm-up
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(group):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / group)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    D, H = a.size, a.det
    delx = 2.4 * (256 / H) * (D / 512)
    drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=delx).to(dev)
    V = drr.density
    lib = _lib.get_lib()
    print(f"# {torch.cuda.get_device_name(0)}  volume {D}^3  detector {H}^2  delx {delx:.2f}")

    def fwd_case(label, s, t, L, tile, xcd, aux=False):
        lib.cdll.ddrr_set_xcd_swizzle(int(xcd))
        B = t.shape[0]
        _, _, nv = ops.siddon_forward(V, s, t, L, count_voxels=True, det=(H, H), tile=tile)
        nvox = int(nv.sum())
        alg = 4 * nvox + B * H * H * 20 + 12 * B
        med, best = timeit(lambda: ops.siddon_forward(V, s, t, L, det=(H, H), tile=tile,
                                                      want_aux=aux))
        print(f"{label:34s} tile {tile[0]:2d}x{tile[1]:<2d} xcd {int(xcd)} aux {int(aux)} B {B:4d} "
              f"vox/ray {nvox / (B * H * H):6.1f}  {med:8.3f} ms (best {best:7.3f})  "
              f"{B / med * 1e3:9.0f} DRR/s  {alg / med / 1e6:8.1f} GB/s alg "
              f"({alg / med / 1e6 / 8000 * 100:5.1f}% of 8 TB/s)", flush=True)
        return med

    def slab_case(*args, **kw):  # the lockstep slab march was removed in round 2
        return None


    def brick_case(label, s, t, L):
        B = t.shape[0]
        _, _, nv = ops.siddon_forward(V, s, t, L, count_voxels=True, det=(H, H))
        nvox = int(nv.sum())
        alg = 4 * nvox + B * H * H * 20 + 12 * B
        med, best = timeit(lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H)))
        ref = ops.siddon_forward(V, s, t, L, det=(H, H))[0]
        out = ops.siddon_forward_bricks(V, s, t, L, (H, H))[0]
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        print(f"{label:34s} BRICK (LDS) B {B:4d} "
              f"vox/ray {nvox / (B * H * H):6.1f}  {med:8.3f} ms (best {best:7.3f})  "
              f"{B / med * 1e3:9.0f} DRR/s  {alg / med / 1e6:8.1f} GB/s alg "
              f"({alg / med / 1e6 / 8000 * 100:5.1f}% of 8 TB/s)  err vs generic {err:.1e}",
              flush=True)
        return med

    base = rays(drr, torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 850.0, 0.0]], device=dev))
    pert32 = rays(drr, *poses(32, 2, dev))
    tiles = [(64, 1), (32, 2), (16, 4), (8, 8), (4, 16), (1, 64)]
    if a.quick:
        tiles = [(16, 4), (8, 8)]
    print("## Siddon forward: base AP pose, B=1 (one DRR in flight: latency / occupancy bound)")
    for tile in tiles:
        fwd_case("base pose", *base, tile, True)
    print("## Siddon forward: base AP pose replicated x32")
    rep = tuple(x.expand(32, *x.shape[1:]).contiguous() for x in base)
    for tile in tiles:
        fwd_case("base pose x32", *rep, tile, True)
    fwd_case("base pose x32", *rep, (16, 4), False)
    for xcd in (1, 0):
        slab_case("base pose x32", *rep, xcd, passes=1)
        slab_case("base pose x32", *rep, xcd)
    brick_case("base pose x32", *rep)
    brick_case("base pose B=1", *base)
    print("## Siddon forward: 32 perturbed poses (bench workload)")
    brick_case("perturbed x32", *pert32)
    brick_case("perturbed x8", *(x[:8].contiguous() for x in pert32))
    brick_case("perturbed x128", *rays(drr, *poses(128, 3, dev)))
    for xcd in (1, 0):
        for passes in (1, 2, 4, 8, 16):
            slab_case("perturbed x32", *pert32, xcd, passes=passes)
    slab_case("perturbed x32 + aux", *pert32, 0, aux=True)
    slab_case("perturbed x32 + aux", *pert32, 0, aux=True, passes=1)
    slab_case("base pose B=1", *base, 0)
    for tile in tiles:
        fwd_case("perturbed x32", *pert32, tile, True)
    fwd_case("perturbed x32", *pert32, (16, 4), False)
    fwd_case("perturbed x32 + aux record", *pert32, (16, 4), True, aux=True)
    fwd_case("perturbed x32 + aux record", *pert32, (8, 8), True, aux=True)
    if not a.quick:
        print("## Siddon forward: per-pose timings of the perturbed set (tile 16x4 vs 8x8)")
        for b in range(0, 32, 4):
            one = tuple(x[b:b + 1].contiguous() for x in pert32)
            rep1 = tuple(x.expand(16, *x.shape[1:]).contiguous() for x in one)
            fwd_case(f"pose {b} x16", *rep1, (16, 4), True)
            slab_case(f"pose {b} x16", *rep1, False)
        big = rays(drr, *poses(128, 3, dev))
        fwd_case("perturbed x128", *big, (16, 4), True)
        slab_case("perturbed x128", *big, False, passes=1)
        slab_case("perturbed x128", *big, False)
        slab_case("perturbed x128", *big, False, passes=8)

    lib.cdll.ddrr_set_xcd_swizzle(1)
    s, t, L = pert32
    B = 32
    print("## other kernels, 32 perturbed poses, tile 16x4")
    go = torch.randn(B, H * H, device=dev)
    _, aux, _ = ops.siddon_forward(V, s, t, L, want_aux=True, det=(H, H))
    med, _ = timeit(lambda: ops.siddon_backward_rays(aux, go, s, t, L))
    print(f"siddon_backward_rays (elementwise)      {med:8.3f} ms")
    med, _ = timeit(lambda: ops.siddon_backward_volume(V, s[:8], t[:8], L[:8], go[:8], det=(H, H)),
                    reps=3, warm=1)
    print(f"siddon_backward_volume B=8 (atomics)    {med:8.3f} ms  ({8 / med * 1e3:.0f} DRR/s)")
    med, _ = timeit(lambda: ops.siddon_forward(V, s[:8], t[:8], L[:8], lookup="mid_nearest",
                                               det=(H, H)), reps=3, warm=1)
    print(f"siddon_forward midpoint lookup B=8      {med:8.3f} ms  ({8 / med * 1e3:.0f} DRR/s)")
    a0, a1 = torch.tensor(0.0, device=dev), torch.tensor(1.0, device=dev)
    for P in (512,):
        med, _ = timeit(lambda: ops.trilinear_forward(V, s[:8], t[:8], L[:8], a0, a1, n_points=P,
                                                      det=(H, H)), reps=3, warm=1)
        samples = 8 * H * H * P
        print(f"trilinear_forward P={P} B=8            {med:8.3f} ms  ({8 / med * 1e3:.0f} DRR/s, "
              f"{samples / med / 1e6:.1f} Gsamples/s)")
        med, _ = timeit(lambda: ops.trilinear_backward(V, s[:4], t[:4], L[:4], go[:4], a0, a1,
                                                       n_points=P, want_volume=True, det=(H, H)),
                        reps=3, warm=1)
        print(f"trilinear_backward(+vol) P={P} B=4     {med:8.3f} ms  ({4 / med * 1e3:.0f} DRR/s)")


if __name__ == "__main__":
    main()
