"""Kernel-only timing of the volume-stationary brick kernel (development tool).
Usage: python tools/brick_bench.py [--size 512] [--det 256] [--layouts 33:1057,36:1168]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("exp")  # the switches below only exist in the tools build (or DDRR_LIB=...)
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--det", type=int, default=256)
ap.add_argument("--layouts", default="33:1057")
ap.add_argument("--cases", default="pert32,pert32aux,base32,pert1,pert8,pert128")
ap.add_argument("--dbg", default="0")
ap.add_argument("--classes", default="18:40")
ap.add_argument("--order", default="id", help="hand-out order of the bricks: id | center | weight (comma list)")
ap.add_argument("--split", default="0:1", help="T:S -- the last T bricks are handed out in S pose parts (comma list)")
ap.add_argument("--sqw", default="8", help="class width(s) of the shared rings (variants >= 16)")
ap.add_argument("--variants", default="-2",
                help="bricks_fwd.hip variants: -2 product default, -1 the general 32^3 fp32 kernel of "
                     "bricks.hip, 0 32^3 fp32 (DDRR_BRICKS_F32), 1 = 5 32x32x64 16-bit (DDRR_BRICKS_Q16), "
                     "2 32^3 16-bit, 10 32^3 16-bit x 2 workgroups per CU, 3 32x32x16 fp32 x 2, 4 / 6 double "
                     "16-bit bricks long in x / y, 7-9 anisotropic fp32 bricks 16x64x32, 64x16x32, 16x32x64; "
                     "16 + v / 32 + v: workgroup-shared rings of 12 / 8 length classes")
ap.add_argument("--storage", default="auto", help="auto (by variant) | f32 | q16 | q16p")
a = ap.parse_args()
dev = torch.device("cuda:0")
D, H = a.size, a.det
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4 * (256 / H) * (D / 512)).to(dev)
V = drr.density
lib = _lib.get_lib()
print(f"# {torch.cuda.get_device_name(0)}  volume {D}^3  detector {H}^2")
base = rays(drr, torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 850.0, 0.0]], device=dev))
sets = {}
for case in a.cases.split(","):
    aux = case.endswith("aux") or case.endswith("auxp")
    name = case[:-4] if case.endswith("auxp") else (case[:-3] if aux else case)
    if name.startswith("base"):
        B = int(name[4:])
        s, t, L = (x.expand(B, *x.shape[1:]).contiguous() for x in base)
    else:
        s, t, L = rays(drr, *poses(int(name[4:]), 2, dev))
    sets[case] = (s, t, L, aux)
import itertools
import ctypes

BRICK = {0: (32, 32, 32), 2: (32, 32, 32), 10: (32, 32, 32), 3: (32, 32, 16), 4: (64, 32, 32), 5: (32, 32, 64),
         1: (32, 32, 64), 6: (32, 64, 32), 7: (16, 64, 32), 8: (64, 16, 32), 9: (16, 32, 64)}


def brick_order(kind, var, s_, t_):
    """int32 device tensor: k-th brick handed out.  center: nearest to the volume centre first;
    weight: largest sum over the poses of the projected pixel-box area first."""
    if kind == "id" or var not in BRICK:
        return None
    bx, by, bz = BRICK[var]
    n = [-(-D // b) for b in (bx, by, bz)]
    ix, iy, iz = torch.meshgrid(*[torch.arange(k, device=dev) for k in n], indexing="ij")
    lo = torch.stack([ix * bx, iy * by, iz * bz], -1).reshape(-1, 3).float()      # id order: z fastest
    hi = torch.minimum(lo + torch.tensor([bx, by, bz], device=dev), torch.tensor([D, D, D], device=dev).float())
    if kind == "center":
        key = (((lo + hi) / 2 - D / 2) ** 2).sum(-1)
        return torch.argsort(key).int().contiguous()
    # weight: project the 8 corners of every brick through every pose's source onto its pixel lattice
    B = t_.shape[0]
    tg = t_.reshape(B, H, H, 3)
    t00, ei, ej = tg[:, 0, 0], (tg[:, -1, 0] - tg[:, 0, 0]) / (H - 1), (tg[:, 0, -1] - tg[:, 0, 0]) / (H - 1)
    src = s_[:, 0]
    corners = torch.stack([torch.where(torch.tensor([(c >> a) & 1 for a in range(3)], device=dev).bool(), hi, lo)
                           for c in range(8)], 1) - 0.5                           # (nb, 8, 3), planes at k - shift
    w = corners[None] - src[:, None, None]                                          # (B, nb, 8, 3)
    A = torch.stack([w, -ei[:, None, None].expand_as(w), -ej[:, None, None].expand_as(w)], -1)  # lambda w - i ei - j ej = r
    r = (t00 - src)[:, None, None, :, None].expand(B, w.shape[1], 8, 3, 1)
    sol = torch.linalg.solve(A, r)[..., 0]
    i, j = sol[..., 1], sol[..., 2]
    i0, i1 = i.amin(-1).clamp(0, H - 1), i.amax(-1).clamp(0, H - 1)
    j0, j1 = j.amin(-1).clamp(0, H - 1), j.amax(-1).clamp(0, H - 1)
    area = ((i1 - i0 + 1) * (j1 - j0 + 1)).sum(0)
    return torch.argsort(area, descending=True).int().contiguous()


keep_alive = []
for lay, dbg, cl, var, sqw, order, split in itertools.product(
        a.layouts.split(","), a.dbg.split(","), a.classes.split(","), a.variants.split(","), a.sqw.split(","),
        a.order.split(","), a.split.split(",")):
    sy, sx = (int(v) for v in lay.split(":"))
    var = int(var)
    if var < 16 and sqw != a.sqw.split(",")[0]:
        continue
    lib.cdll.ddrr_set_brick_variant(var)
    lib.cdll.ddrr_set_brick_sq_width(ctypes.c_float(float(sqw)))
    storage = "q16" if var >= 0 and var % 16 in (1, 2, 4, 5, 6, 10) else "f32"
    if a.storage != "auto":
        storage = a.storage
    lib.cdll.ddrr_set_brick_debug(int(dbg))
    t1, t2 = (float(v) for v in cl.split(":"))
    import ctypes
    lib.cdll.ddrr_set_brick_classes(ctypes.c_float(t1), ctypes.c_float(t2))
    T_, S_ = (int(v) for v in split.split(":"))
    lib.cdll.ddrr_set_brick_split(T_, S_)
    lay = f"dbg{dbg} cls{cl if var < 16 else sqw} var{var:2d} {storage} ord {order} split {split}"
    rc = lib.cdll.ddrr_set_brick_layout(sy, sx)
    if rc != 0:
        print(f"layout {lay}: rejected")
        continue
    for case, (s, t, L, aux) in sets.items():
        B = t.shape[0]
        tab = brick_order(order, var, s, t)
        keep_alive.append(tab)
        lib.cdll.ddrr_set_brick_order(ctypes.c_void_p(tab.data_ptr() if tab is not None else 0))
        _, _, nv = ops.siddon_forward(V, s, t, L, count_voxels=True, det=(H, H))
        nvox = int(nv.sum())
        alg = 4 * nvox + B * H * H * 20 + 12 * B
        vm = ops.volume_absmax(V) if case.endswith("auxp") else 0.0
        med, best = timeit(lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H), want_aux=aux, record_vmax=vm,
                                                             storage=storage))
        ref = ops.siddon_forward(V, s, t, L, det=(H, H))[0]
        out = ops.siddon_forward_bricks(V, s, t, L, (H, H), want_aux=aux, storage=storage)[0]
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        if aux:  # the record: per-pose sums of the target gradients against the generic walk's
            _, gaux, _ = ops.siddon_forward(V, s[:4], t[:4], L[:4], want_aux=True, det=(H, H))
            _, baux = ops.siddon_forward_bricks(V, s[:4], t[:4], L[:4], (H, H), want_aux=True, storage=storage)
            go = torch.ones_like(L[:4])
            g1 = ops.siddon_backward_rays(gaux, go, s[:4], t[:4], L[:4])[1].double().sum(1)
            g2 = ops.siddon_backward_rays(baux, go, s[:4], t[:4], L[:4])[1].double().sum(1)
            err = max(err, -((g1 - g2).abs().max() / g1.abs().max()).item())  # (negative: gradient error)
            lay = lay + f" gerr {((g1 - g2).abs().max() / g1.abs().max()).item():.1e}"
        print(f"{lay:58s} {case:12s} B {B:4d} vox/ray {nvox / (B * H * H):6.1f}  {med:8.3f} ms "
              f"(best {best:7.3f})  {B / med * 1e3:9.0f} DRR/s  {alg / med / 1e6:8.1f} GB/s alg "
              f"({alg / med / 1e6 / 8000 * 100:5.1f}% of 8 TB/s)  err vs generic {err:.1e}", flush=True)
