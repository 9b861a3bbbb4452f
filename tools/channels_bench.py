"""mask_to_channels at the reference's published size (introduction.ipynb:230-272: 512x512x133
CT, 119 label channels, 200x200 detector: 38.8 ms vs 25.4 ms for the plain render on an RTX
2080 Ti, i.e. +54 %): the plain Siddon render against the channel render, one MI355X.
The label map is synthetic (piecewise constant blocks of 32 x 32 x 17 voxels, 119 labels), the
volume uniform noise.  The reference's example label map (diffdrr/data/mask.nii.gz, 512x512x133,
90 of 119 labels present, 81 % background) changes label every 110 voxel steps along x / y and
every 38 along z; the synthetic one every 32 / 17: three times as many label runs per ray to
flush, i.e. the channel render timed here is on the pessimistic side."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR  # noqa: E402
from diffdrr_amd.data import make_subject  # noqa: E402
from tools.kernel_sweep import timeit  # noqa: E402

import numpy as np  # noqa: E402

dev = torch.device("cuda:0")
dims, C, H = (512, 512, 133), 119, 200
g = torch.Generator().manual_seed(0)
vol = torch.rand(*dims, generator=g)
if "--real-mask" in sys.argv:
    # the reference's own example label map (diffdrr/data/mask.nii.gz), committed down-sampled
    # 2 x 2 in-plane (tests/golden/make_golden_mask.py) and repeated back to 512 x 512 x 133:
    # label runs of the original lengths; 81 % background, 90 of 119 labels present
    fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                              "reference_mask_ds2.npz"))
    mask = torch.from_numpy(fx["labels"]).repeat_interleave(2, 0).repeat_interleave(2, 1)[:512, :512]
    assert tuple(mask.shape) == dims and int(mask.max()) == C - 1
    what = "the reference's example label map (TotalSegmentator, 2x2 in-plane repeat of the fixture)"
else:
    coarse = torch.randint(0, C, (16, 16, 8), generator=g)
    mask = coarse
    for ax, d in enumerate(dims):
        idx = (torch.arange(d) * coarse.shape[ax] // d).clamp_max(coarse.shape[ax] - 1)
        mask = mask.index_select(ax, idx)
    mask[0, 0, 0] = C - 1
    what = "synthetic label blocks of 32 x 32 x 17 voxels"
subject = make_subject(vol, spacing=(0.703, 0.703, 2.5), mask=mask)
drr = DRR(subject, sdd=1020.0, height=H, delx=2.0).to(dev)
print(f"# {torch.cuda.get_device_name(0)}: {dims} volume, {C} labels ({what}), {H}x{H} detector")
for B in (1, 8):
    rot = torch.zeros(B, 3, device=dev) + torch.linspace(0, 0.3, B, device=dev)[:, None]
    xyz = torch.tensor([[0.0, 850.0, 0.0]], device=dev).expand(B, 3).contiguous()
    with torch.no_grad():
        plain, _ = timeit(lambda: drr(rot, xyz, parameterization="euler_angles", convention="ZXY"))
        drr.renderer.grid_path = "generic"
        plain_g, _ = timeit(lambda: drr(rot, xyz, parameterization="euler_angles", convention="ZXY"))
        chan_g, _ = timeit(lambda: drr(rot, xyz, parameterization="euler_angles", convention="ZXY",
                                       mask_to_channels=True))
        cg = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", mask_to_channels=True)
        drr.renderer.grid_path = "bricks"
        chan, _ = timeit(lambda: drr(rot, xyz, parameterization="euler_angles", convention="ZXY",
                                     mask_to_channels=True))
        a = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
        c = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", mask_to_channels=True)
    err = ((c.sum(1, keepdim=True) - a).abs().max() / a.abs().max()).item()
    err_g = ((c - cg).abs().max() / cg.abs().max()).item()
    print(f"B {B}: plain render (bricks) {plain:7.3f} ms | plain render (per-ray walk) {plain_g:7.3f} ms | "
          f"{C}-channel render: bricks {chan:7.3f} ms = {chan / plain:5.2f} x plain bricks, per-ray "
          f"kernel {chan_g:7.3f} ms | channel sum vs plain {err:.1e}, bricks vs per-ray channels "
          f"{err_g:.1e}", flush=True)

# the kernels alone (rays precomputed, labels converted): what the entry points cost
from diffdrr_amd import convert, ops  # noqa: E402
from diffdrr_amd.renderers import _labels_u8  # noqa: E402

(labels, _, _), = _labels_u8(drr.mask)
for B in (1, 8):
    rot = torch.zeros(B, 3, device=dev) + torch.linspace(0, 0.3, B, device=dev)[:, None]
    xyz = torch.tensor([[0.0, 850.0, 0.0]], device=dev).expand(B, 3).contiguous()
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s_, t_ = drr.affine_inverse(source).contiguous(), drr.affine_inverse(target).contiguous()
        k_plain, _ = timeit(lambda: ops.siddon_forward_bricks(drr.density, s_, t_, L, (H, H)))
        k_chb, _ = timeit(lambda: ops.siddon_forward_channels_bricks(drr.density, labels, C, s_, t_,
                                                                      L, (H, H)))
        k_chr, _ = timeit(lambda: ops.siddon_forward_channels(drr.density, labels, C, s_, t_, L,
                                                              det=(H, H)))
    print(f"B {B} entry points only: plain bricks {k_plain:7.3f} ms | channels on bricks {k_chb:7.3f} ms "
          f"= {k_chb / k_plain:5.2f} x | per-ray channel kernel {k_chr:7.3f} ms", flush=True)
    # the backward of the channel render (per-ray re-walk, gathers of grad_out[b, label, n]):
    # ray / img gradients only, and with the volume gradient (atomics)
    go = torch.rand(B, C, H * H, device=dev)
    with torch.no_grad():
        k_bwd, _ = timeit(lambda: ops.siddon_backward_channels(drr.density, labels, s_, t_, L, go,
                                                               want_volume=False, det=(H, H)))
        k_bwdv, _ = timeit(lambda: ops.siddon_backward_channels(drr.density, labels, s_, t_, L, go,
                                                                want_volume=True, det=(H, H)))
        k_aux, _ = timeit(lambda: ops.siddon_forward_bricks(drr.density, s_, t_, L, (H, H), want_aux=True))
        k_bwdb, _ = timeit(lambda: ops.siddon_backward_channels_bricks(drr.density, labels, s_, t_, L, go, (H, H)))
        gb = ops.siddon_backward_channels_bricks(drr.density, labels, s_, t_, L, go, (H, H))
        gr = ops.siddon_backward_channels(drr.density, labels, s_, t_, L, go, want_volume=False, det=(H, H))
        k_vb, _ = timeit(lambda: ops.siddon_backward_channels_volume_bricks(labels, s_, t_, L, go, (H, H)))
        k_pv, _ = timeit(lambda: ops.siddon_backward_volume_bricks(drr.density.shape, s_, t_, L, go[:, 0].contiguous(), (H, H)))
        gvb = ops.siddon_backward_channels_volume_bricks(labels, s_, t_, L, go, (H, H))
        gvr = ops.siddon_backward_channels(drr.density, labels, s_, t_, L, go, want_rays=False, want_img=False,
                                           want_volume=True, det=(H, H))[3]
    print(f"B {B} channel backward, VOLUME gradient: per-ray kernel (global atomics, with the ray gradients) "
          f"{k_bwdv:7.3f} ms | ON THE BRICKS {k_vb:7.3f} ms = {k_bwdv / k_vb:4.1f} x faster (plain volume gradient "
          f"on the bricks {k_pv:7.3f} ms) | bricks vs per-ray {float((gvb - gvr).abs().max() / gvr.abs().max()):.1e}",
          flush=True)
    with torch.no_grad():
        pass
    same = float(((gb[1] - gr[1]).abs().amax(-1) <= 1e-3 * gr[1].abs().max()).float().mean())
    print(f"B {B} channel backward, rays + img: per-ray kernel {k_bwd:7.3f} ms | ON THE BRICKS {k_bwdb:7.3f} ms "
          f"= {k_bwd / k_bwdb:4.1f} x faster (rays agreeing to 1e-3: {100 * same:.2f} %, d/d img "
          f"{float((gb[2] - gr[2]).abs().max() / gr[2].abs().max()):.1e}) | per-ray with the volume gradient "
          f"{k_bwdv:7.3f} ms | for scale: plain forward + record on the bricks {k_aux:7.3f} ms", flush=True)

# the marcher (Trilinear.forward renderers.py:205-254): plain render (volume-stationary bricks)
# against its mask branch (per-ray kernel ddrr_trilinear_forward_channels), 500 samples per ray
tri = DRR(subject, sdd=1020.0, height=H, delx=2.0, renderer="trilinear").to(dev)
for B in (1, 8):
    rot = torch.zeros(B, 3, device=dev) + torch.linspace(0, 0.3, B, device=dev)[:, None]
    xyz = torch.tensor([[0.0, 850.0, 0.0]], device=dev).expand(B, 3).contiguous()
    with torch.no_grad():
        kw = dict(parameterization="euler_angles", convention="ZXY", n_points=500)
        t_plain, _ = timeit(lambda: tri(rot, xyz, **kw))
        tri.renderer.use_bricks = False
        t_plain_r, _ = timeit(lambda: tri(rot, xyz, **kw))
        tri.renderer.use_bricks = True
        t_chan, _ = timeit(lambda: tri(rot, xyz, mask_to_channels=True, **kw))
        tri.renderer.channels_on_bricks = False
        t_chan_r, _ = timeit(lambda: tri(rot, xyz, mask_to_channels=True, **kw))
        cr = tri(rot, xyz, mask_to_channels=True, **kw)
        tri.renderer.channels_on_bricks = True
        a = tri(rot, xyz, **kw)
        c = tri(rot, xyz, mask_to_channels=True, **kw)
    err = ((c.sum(1, keepdim=True) - a).abs().max() / a.abs().max()).item()
    err_r = ((c - cr).abs().max() / cr.abs().max()).item()
    print(f"trilinear B {B}: plain render bricks {t_plain:7.3f} ms, per-ray {t_plain_r:7.3f} ms | {C}-channel "
          f"render: bricks {t_chan:7.3f} ms = {t_chan / t_plain:5.2f} x plain bricks, per-ray kernel "
          f"{t_chan_r:7.3f} ms | channel sum vs plain {err:.1e}, bricks vs per-ray channels {err_r:.1e}",
          flush=True)
    # the marcher's channel backward for rays / img / range: per-ray re-march against the weighted
    # record on the bricks (ddrr_trilinear_backward_channels_bricks + ddrr_trilinear_backward_rays)
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = tri.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s_, t_ = tri.affine_inverse(source).contiguous(), tri.affine_inverse(target).contiguous()
        a0, a1 = (x.reshape(1) for x in ops.trilinear_alpha_range(s_, t_, tri.density.shape))
        go = torch.rand(B, C, H * H, device=dev)
        tb_r, _ = timeit(lambda: ops.trilinear_backward_channels(tri.density, labels, s_, t_, L, go, a0, a1,
                                                                 n_points=500, det=(H, H)))
        tb_b, _ = timeit(lambda: ops.trilinear_backward_channels_bricks(tri.density, labels, s_, t_, L, go,
                                                                        a0, a1, (H, H), n_points=500))
        rb = ops.trilinear_backward_channels_bricks(tri.density, labels, s_, t_, L, go, a0, a1, (H, H),
                                                    n_points=500)
        rp = ops.trilinear_backward_channels(tri.density, labels, s_, t_, L, go, a0, a1, n_points=500,
                                             det=(H, H))
        tv_b, _ = timeit(lambda: ops.trilinear_backward_channels_volume_bricks(labels, s_, t_, L, go, a0, a1,
                                                                               (H, H), n_points=500))
        tv_r, _ = timeit(lambda: ops.trilinear_backward_channels(tri.density, labels, s_, t_, L, go, a0, a1,
                                                                 n_points=500, det=(H, H), want_rays=False,
                                                                 want_img=False, want_alpha=False,
                                                                 want_volume=True))
        gvb = ops.trilinear_backward_channels_volume_bricks(labels, s_, t_, L, go, a0, a1, (H, H), n_points=500)
        gvr = ops.trilinear_backward_channels(tri.density, labels, s_, t_, L, go, a0, a1, n_points=500,
                                              det=(H, H), want_rays=False, want_img=False, want_alpha=False,
                                              want_volume=True)["g_volume"]
    print(f"trilinear B {B} channel backward, VOLUME gradient: per-ray kernel (global atomics) {tv_r:7.3f} ms | ON THE "
          f"BRICKS {tv_b:7.3f} ms = {tv_r / tv_b:4.1f} x faster | bricks vs per-ray "
          f"{float((gvb - gvr).abs().max() / gvr.abs().max()):.1e}", flush=True)
    with torch.no_grad():
        pass
    errs = {k: float((rb[k] - rp[k]).abs().max() / rp[k].abs().max()) for k in ("g_img", "g_target", "g_alpha")}
    print(f"trilinear B {B} channel backward, rays + img + range: per-ray kernel {tb_r:7.3f} ms | ON THE "
          f"BRICKS {tb_b:7.3f} ms = {tb_r / tb_b:4.1f} x faster | bricks vs per-ray: "
          + ", ".join(f"{k} {v:.1e}" for k, v in errs.items()), flush=True)
