"""Randomised consistency sweep (development tool, GPU): the volume-stationary brick kernels
against the per-ray generic kernels over random volume shapes, detector sizes and poses
(including sources inside / next to the volume and large rotations).
Usage: python tools/fuzz_bricks.py [--cases 40] [--seed 0]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, convert, ops  # noqa: E402
from diffdrr_amd.data import make_subject  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--smooth", action="store_true", help="smooth volume instead of noise")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(a.seed)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def ru(lo, hi, *shape):
    return lo + (hi - lo) * torch.rand(*shape, generator=g)


worst = {"fwd": 0.0, "aux": 0.0, "volgrad": 0.0, "tri": 0.0, "trivol": 0.0, "q16": 0.0, "chan": 0.0, "chan_bwd": 0.0, "chan_bwd_sum": 0.0, "chan_volgrad": 0.0, "chan_trivolgrad": 0.0}
for case in range(a.cases):
    dims = (ri(20, 150), ri(20, 150), ri(20, 150))
    if case % 2 == 0:  # z a multiple of 4 (aligned rows; any other D.z is staged from dword-aligned quads)
        dims = (dims[0], dims[1], 4 * ri(5, 40))
    H, W = ri(2, 90), ri(2, 90)
    B = ri(1, 5)
    spacing = tuple(float(x) for x in ru(0.5, 2.0, 3))
    vol = torch.rand(*dims, generator=g)
    dist = (case // 4) % 4
    if dist == 1:    # CT-like: air = 0, soft tissue ~1, a few bright voxels (metal) -> guarded bricks
        vol = torch.where(vol < 0.3, torch.zeros(()), 0.9 + 0.2 * vol)
        vol[torch.rand(*dims, generator=g) < 1e-3] = 40.0
    elif dist == 2:  # mixed sign, wide range
        vol = (vol - 0.5) * torch.exp(4.0 * torch.rand(*dims, generator=g))
    elif dist == 3:  # sparse: mostly zeros (air bricks are skipped)
        vol = vol * (torch.rand(*dims, generator=g) < 0.05)
    if a.smooth:  # a smooth field: neighbouring voxels differ by ~1 %
        ax = [torch.linspace(0, float(ru(2, 6, 1)), d) for d in dims]
        vol = 0.5 + 0.5 * torch.sin(ax[0])[:, None, None] * torch.cos(ax[1])[None, :, None] * torch.sin(ax[2] + 1.0)[None, None, :]
    drr = DRR(make_subject(vol, spacing=spacing), sdd=float(ru(300, 1500, 1)), height=H, width=W,
              delx=float(ru(0.5, 4.0, 1))).to(dev)
    kind = case % 4
    rot = ru(-3.1, 3.1, B, 3) if kind != 0 else ru(-0.3, 0.3, B, 3)
    if kind == 3:   # source inside or right next to the volume
        xyz = ru(-0.6, 0.6, B, 3) * torch.tensor(dims) * torch.tensor(spacing)
    else:
        xyz = torch.stack([ru(-50, 50, B), ru(200, 1200, B), ru(-50, 50, B)], -1)
    with torch.no_grad():
        pose = convert(rot.to(dev), xyz.to(dev), parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    V = drr.density
    ref, aux_ref, _ = ops.siddon_forward(V, s, t, L, want_aux=True)
    out, aux = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=True)
    scale = ref.abs().max().item() + 1e-30
    e_f = (out - ref).abs().max().item() / scale
    go = torch.rand(ref.shape, generator=g).to(dev)
    gsb, gtb, gib = ops.siddon_backward_rays(aux, go, s, t, L)
    gsg, gtg, gig = ops.siddon_backward_rays(aux_ref, go, s, t, L)
    e_a = ((gtb.sum(1) - gtg.sum(1)).abs().max() / (gtg.sum(1).abs().max() + 1e-30)).item()
    e_a = max(e_a, ((gib - gig).abs().max() / (gig.abs().max() + 1e-30)).item())
    gv_ref = ops.siddon_backward_volume(V, s, t, L, go)
    gv = ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (H, W))
    e_v = ((gv - gv_ref).abs().max() / (gv_ref.abs().max() + 1e-30)).item()
    # trilinear
    from diffdrr_amd.renderers import get_alpha_minmax
    lo, hi = get_alpha_minmax(s, t, torch.tensor(V.shape, device=dev), 0.5, 1e-8)
    amin, amax = lo.min().reshape(1), hi.max().reshape(1)
    P = ri(20, 300)
    if min(H, W) >= 2 and amax.item() > amin.item():
        tr = ops.trilinear_forward(V, s, t, L, amin, amax, n_points=P)
        tb = ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (H, W), n_points=P)
        e_t = ((tb - tr).abs().max() / (tr.abs().max() + 1e-30)).item()
        rv = ops.trilinear_backward(V, s, t, L, go, amin, amax, n_points=P, want_rays=False,
                                    want_img=False, want_alpha=False, want_volume=True)["g_volume"]
        bv = ops.trilinear_backward_volume_bricks(V.shape, s, t, L, go, amin, amax, (H, W), n_points=P)
        e_tv = ((bv - rv).abs().max() / (rv.abs().max() + 1e-30)).item()
        _, taux = ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (H, W), n_points=P, want_aux=True)
        rec = ops.trilinear_backward_rays(taux, go, s, t, L, amin, amax, n_points=P)
        rem = ops.trilinear_backward(V, s, t, L, go, amin, amax, n_points=P)
        for k in ("g_source", "g_target", "g_img"):
            e_t = max(e_t, ((rec[k] - rem[k]).abs().max() / (rem[k].abs().max() + 1e-30)).item())
    else:
        e_t = e_tv = 0.0
    # 16-bit bricks, from the volume and from the packed copy (first and cached call), with record
    e_q = 0.0
    if min(H, W) >= 2:
        for st in ("q16", "q16p", "q16p"):
            oq, auxq = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=True, storage=st)
            op, _ = ops.siddon_forward_bricks(V, s, t, L, (H, W), storage=st)
            e_q = max(e_q, (oq - ref).abs().max().item() / scale, (op - ref).abs().max().item() / scale)
            giq = ops.siddon_backward_rays(auxq, go, s, t, L)[2]
            e_q = max(e_q, ((giq - gig).abs().max() / (gig.abs().max() + 1e-30)).item())
        # a subsample through the kernels' pixel mask (every storage, with and without the record):
        # the drawn pixels as in the unmasked render, exact zeros at the others
        keep = torch.rand(H * W, generator=g) < (0.1 if case % 2 else 0.5)
        idx = keep.nonzero().flatten().to(dev)
        if idx.numel():
            pm = ops.pixel_mask_of(idx, H * W)
            for st in ("f32", "q16p"):
                for aux_ in (False, True):
                    om, _ = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=aux_, storage=st, pixel_mask=pm)
                    od, _ = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=aux_, storage=st)
                    assert float(om[:, ~keep.to(dev)].abs().max()) == 0.0 if (~keep).any() else True
                    e_q = max(e_q, (om[:, idx] - od[:, idx]).abs().max().item() / scale)
    # mask_to_channels on the bricks against the per-ray channel kernels
    e_c = e_cb = e_cg = e_cv = e_ctv = 0.0
    if min(H, W) >= 2:
        C = ri(2, 40)
        lab = torch.randint(0, C, tuple((d + 5) // 6 for d in dims), generator=g).to(torch.uint8)
        lab = lab.repeat_interleave(6, 0).repeat_interleave(6, 1).repeat_interleave(6, 2)
        lab = lab[:dims[0], :dims[1], :dims[2]].contiguous().to(dev)
        if case % 3 == 0:
            C = max(1, C - ri(1, 5))  # the highest labels have no channel
        cb = ops.siddon_forward_channels_bricks(V, lab, C, s, t, L, (H, W))
        cr = ops.siddon_forward_channels(V, lab, C, s, t, L)
        e_c = ((cb - cr).abs().max() / (cr.abs().max() + 1e-30)).item()
        # the channel backward on the bricks (weighted record) against the per-ray kernel: d/d img
        # to rounding, per-pose sums of d/d target (single rays may book a tie on the other axis)
        goc = torch.rand(B, C, H * W, generator=g).to(dev)
        bs, bt, bi = ops.siddon_backward_channels_bricks(V, lab, s, t, L, goc, (H, W))
        ps, pt, pi, _ = ops.siddon_backward_channels(V, lab, s, t, L, goc, det=(H, W))
        e_cb = ((bi - pi).abs().max() / (pi.abs().max() + 1e-30)).item()
        e_cg = ((bt.sum(1) - pt.sum(1)).abs().max() / (pt.sum(1).abs().max() + 1e-30)).item()
        # ... and its volume gradient (the LDS brick as accumulator, labels in the words' low byte)
        vb = ops.siddon_backward_channels_volume_bricks(lab, s, t, L, goc, (H, W))
        vr = ops.siddon_backward_channels(V, lab, s, t, L, goc, det=(H, W), want_rays=False, want_img=False,
                                          want_volume=True)[3]
        e_cv = ((vb - vr).abs().max() / (vr.abs().max() + 1e-30)).item()
        if amax.item() > amin.item():
            tcb = ops.trilinear_forward_channels_bricks(V, lab, C, s, t, L, amin, amax, (H, W), n_points=P)
            tcr = ops.trilinear_forward_channels(V, lab, C, s, t, L, amin, amax, n_points=P)
            e_c = max(e_c, ((tcb - tcr).abs().max() / (tcr.abs().max() + 1e-30)).item())
            rb = ops.trilinear_backward_channels_bricks(V, lab, s, t, L, goc, amin, amax, (H, W), n_points=P)
            rp = ops.trilinear_backward_channels(V, lab, s, t, L, goc, amin, amax, n_points=P)
            for k in ("g_img", "g_target", "g_source", "g_alpha"):
                e_cb = max(e_cb, ((rb[k] - rp[k]).abs().max() / (rp[k].abs().max() + 1e-30)).item())
            tvb = ops.trilinear_backward_channels_volume_bricks(lab, s, t, L, goc, amin, amax, (H, W), n_points=P)
            tvr = ops.trilinear_backward_channels(V, lab, s, t, L, goc, amin, amax, n_points=P, want_rays=False,
                                                  want_img=False, want_alpha=False, want_volume=True)["g_volume"]
            e_ctv = ((tvb - tvr).abs().max() / (tvr.abs().max() + 1e-30)).item()
    for k, e in zip(worst, (e_f, e_a, e_v, e_t, e_tv, e_q, e_c, e_cb, e_cg, e_cv, e_ctv)):
        worst[k] = max(worst[k], e if e == e else float("inf"))
    # (a voxel's Siddon gradient is one or two segment lengths, each a difference of two fp32 alphas:
    # two fp32 walks agree to ~1e-4 of it, more where alpha is large against the voxel)
    flag = " <<<" if max(e_f, e_t, e_tv, e_q, e_c, e_cb, e_ctv) > 2e-4 or max(e_v, e_cv) > 1e-3 or max(e_a, e_cg) > 5e-3 or e_f != e_f else ""
    print(f"case {case:3d} kind {kind} dims {dims} det {H}x{W} B {B} P {P}: fwd {e_f:.1e} "
          f"pose-grad {e_a:.1e} volgrad {e_v:.1e} tri {e_t:.1e} trivol {e_tv:.1e} q16 {e_q:.1e} "
          f"channels {e_c:.1e} chan-bwd {e_cb:.1e} / sums {e_cg:.1e} chan-volgrad {e_cv:.1e} tri {e_ctv:.1e} dist {dist}{flag}", flush=True)
print("worst", {k: f"{v:.1e}" for k, v in worst.items()})
