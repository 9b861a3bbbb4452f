"""Experiment: how much do (A) z-epipolar ray ordering and (B) sub-volume passes buy?
Both are emulated on top of the current kernel without code changes:
 (A) permute the ray list so that 64 consecutive rays lie in one plane through the
     source that contains the volume's z axis (same (x, y) row at equal depth);
 (B) render x-slabs of the volume one after the other (each slab fits the 256 MiB
     Infinity Cache), all poses per slab, and add the partial images.
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("exp")  # experiment switches live in the tools build only
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402


def epipolar_perm(s, t, bin_vox=1.0, D=512):
    """(B,N) permutation: sort rays by (azimuth bin about z through the source, polar slope)."""
    d = t - s  # (B,N,3)
    phi = torch.atan2(d[..., 1], d[..., 0])
    rho = d[..., :2].norm(dim=-1)
    theta = d[..., 2] / rho
    c = torch.tensor([(D - 1) / 2] * 2, device=s.device)
    rmax = (s[..., :2] - c).norm(dim=-1) + D / math.sqrt(2)  # (B,1)
    nbin = (phi * (rmax / bin_vox)).floor().to(torch.int64)
    nbin = nbin - nbin.amin(dim=1, keepdim=True)
    tq = ((theta - theta.amin(dim=1, keepdim=True)) /
          (theta.amax(dim=1, keepdim=True) - theta.amin(dim=1, keepdim=True) + 1e-12) * (2**30)
          ).to(torch.int64)
    key = (nbin << 32) | tq
    return key.argsort(dim=1)


def main():
    dev = torch.device("cuda:0")
    D, H = 512, 256
    drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4).to(dev)
    V = drr.density
    lib = _lib.get_lib()
    sets = {
        "base x32": tuple(x.expand(32, *x.shape[1:]).contiguous() for x in rays(
            drr, torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 850.0, 0.0]], device=dev))),
        "perturbed x32": rays(drr, *poses(32, 2, dev)),
    }
    for name, (s, t, L) in sets.items():
        B, N = t.shape[:2]
        ref = ops.siddon_forward(V, s, t, L, det=(H, H), tile=(16, 4))[0]
        for xcd in (1, 0):
            lib.cdll.ddrr_set_xcd_swizzle(xcd)
            med, _ = timeit(lambda: ops.siddon_forward(V, s, t, L, det=(H, H), tile=(16, 4)))
            print(f"{name:14s} xcd {xcd} tile 16x4 baseline           {med:8.3f} ms", flush=True)
            for bv in (0.5, 1.0, 2.0, 4.0):
                tp = timeit(lambda: epipolar_perm(s, t, bv), reps=3)[0]
                perm = epipolar_perm(s, t, bv)
                t2 = t.gather(1, perm[..., None].expand(-1, -1, 3)).contiguous()
                L2 = L.gather(1, perm).contiguous()
                med, _ = timeit(lambda: ops.siddon_forward(V, s, t2, L2))
                out = torch.empty_like(ref).scatter_(1, perm, ops.siddon_forward(V, s, t2, L2)[0])
                ok = torch.equal(out, ref)
                print(f"{name:14s} xcd {xcd} epipolar order bin {bv:3.1f} vox   {med:8.3f} ms   "
                      f"(perm build {tp:.3f} ms, exact={ok})", flush=True)
        # (B) x-slab passes with the best ordering
        perm = epipolar_perm(s, t, 1.0)
        t2 = t.gather(1, perm[..., None].expand(-1, -1, 3)).contiguous()
        L2 = L.gather(1, perm).contiguous()
        for xcd in (1, 0):
            lib.cdll.ddrr_set_xcd_swizzle(xcd)
            for K in (2, 4, 8, 16):
                w = D // K

                def slabs(tt=t2, LL=L2):
                    acc = None
                    for k in range(K):
                        sub = V[k * w:(k + 1) * w]
                        off = torch.tensor([k * w, 0.0, 0.0], device=dev)
                        o = ops.siddon_forward(sub, s - off, tt - off, LL)[0]
                        acc = o if acc is None else acc + o
                    return acc

                med, _ = timeit(slabs, reps=5)
                out = torch.empty_like(ref).scatter_(1, perm, slabs())
                err = (out - ref).abs().max().item() / ref.abs().max().item()
                print(f"{name:14s} xcd {xcd} epipolar + {K:2d} x-slab passes     {med:8.3f} ms   "
                      f"(err {err:.1e})", flush=True)


if __name__ == "__main__":
    main()
