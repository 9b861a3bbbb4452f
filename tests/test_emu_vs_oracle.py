"""The per-ray kernel cores (diffdrr_amd/csrc/*_core.h), compiled for the host by
tests/emu, against the oracle and the reference goldens.  This checks the
traversal *logic* the HIP kernels execute (3-way merge, software-pipelined
fetch slots, aux record, scatter, tile map) in the GPU-less container; the same
comparisons run on the real kernels in test_gpu_parity.py."""
import numpy as np
import pytest

import oracle
import conftest
from conftest import golden, rel_err
from diffdrr_amd._lib import SIDDON_AUX

FWD_TOL = 1e-4    # north-star tolerance: image-normalised error vs the reference fp32
GRAD_TOL = 1e-3


def P(a):
    return None if a is None else a.ctypes.data


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def load(name):
    g = golden(name)
    vol, src, tgt = f32(g["volume"]), f32(g["source"]), f32(g["target"])
    B, N, _ = tgt.shape
    return g, vol, src, tgt, f32(g["img_f32"].reshape(B, N)), B, N


def emu_siddon(emu, vol, src, tgt, img, shift=0.5, red=0, lookup=0, align=0, aux=False,
               cnt=False, det=(0, 0), tile=(1, 64)):
    B, N, _ = tgt.shape
    out = np.zeros((B, N), np.float32)
    a = np.zeros((B, N, SIDDON_AUX), np.float32) if aux else None
    c = np.zeros((B, N), np.int32) if cnt else None
    emu.call("ddrr_siddon_forward", P(vol), *vol.shape, P(src), src.shape[1], P(tgt), P(img), B,
             N, shift, 1e-8, red, lookup, align, det[0], det[1], tile[0], tile[1], P(out), P(a),
             P(c), None)
    return out, a, c


@pytest.mark.parametrize("name,red,shift", [
    ("siddon_sum", 0, 0.5), ("siddon_sum_oblique", 0, 0.5), ("siddon_max", 1, 0.5),
    ("siddon_per_ray_source", 0, 0.5), ("siddon_shift0", 0, 0.0), ("siddon_stopgrad", 0, 0.5),
])
def test_siddon_forward_and_gradients(emu_lib, name, red, shift):
    g, vol, src, tgt, img, B, N = load(name)
    out, aux, _ = emu_siddon(emu_lib, vol, src, tgt, img, shift=shift, red=red, aux=True)
    assert rel_err(out, g["out_f32"].reshape(B, N)) < FWD_TOL
    # no worse than twice the reference's own fp32 error w.r.t. its fp64 run (+ slack)
    ref_err = rel_err(g["out_f32"], g["out_f64"])
    assert rel_err(out, g["out_f64"].reshape(B, N)) < 2 * ref_err + 5e-6
    # without the aux record the result is bit-identical
    out2, _, _ = emu_siddon(emu_lib, vol, src, tgt, img, shift=shift, red=red)
    assert np.array_equal(out, out2)

    go = f32(g["grad_out_f32"].reshape(B, N))
    gs, gt = np.zeros((B, N, 3), np.float32), np.zeros((B, N, 3), np.float32)
    gi = np.zeros((B, N), np.float32)
    emu_lib.call("ddrr_siddon_backward_rays", P(aux), 0, P(go), P(src), src.shape[1], P(tgt), P(img),
                 B, N, 1e-8, red, P(gs), P(gt), P(gi), None)
    gs = gs.sum(1, keepdims=True) if src.shape[1] == 1 else gs
    assert rel_err(gs, g["g_source_f64"]) < GRAD_TOL
    assert rel_err(gt, g["g_target_f64"]) < GRAD_TOL
    if "g_img_f64" in g.files:
        assert rel_err(gi, g["g_img_f64"].reshape(B, N)) < GRAD_TOL
        gv = np.zeros_like(vol)
        emu_lib.call("ddrr_siddon_backward_volume", P(vol), *vol.shape, P(src), src.shape[1],
                     P(tgt), P(img), P(go), B, N, shift, 1e-8, red, 0, 0, 1, 64, P(gv), None)
        assert rel_err(gv, g["g_volume_f64"]) < GRAD_TOL


def test_siddon_special_rays(emu_lib):
    g, vol, src, tgt, img, B, N = load("siddon_special_rays")
    out, _, cnt = emu_siddon(emu_lib, vol, src, tgt, img, cnt=True)
    assert rel_err(out, g["out_f64"].reshape(B, N)) < 1e-5
    assert out[5, 0] == 0 and out[6, 0] == 0 and cnt[5, 0] == 0 and cnt[6, 0] == 0
    # generic midpoint path agrees as well
    out_mid, _, _ = emu_siddon(emu_lib, vol, src, tgt, img, lookup=1)
    # ray 11 runs exactly inside an x-plane (x = 2.5): which of the two voxels a
    # midpoint rounds to is decided by the 1e-8 eps, which fp32 cannot resolve; the
    # midpoint path then makes the reference's fp32 choice, the stepping path its
    # fp64 choice.  Every other ray is unambiguous.
    keep = np.arange(B) != 11
    assert rel_err(out_mid[keep], g["out_f64"].reshape(B, N)[keep]) < 1e-5
    assert abs(out_mid[11, 0] - g["out_f32"].reshape(B, N)[11, 0]) < 1e-4


@pytest.mark.parametrize("name,lookup,align", [
    ("siddon_bilinear", 2, 0), ("siddon_align_corners", 1, 1), ("siddon_sum", 1, 0),
    ("siddon_sum_oblique", 1, 0),
])
def test_siddon_generic_lookup(emu_lib, name, lookup, align):
    g, vol, src, tgt, img, B, N = load(name)
    out, _, _ = emu_siddon(emu_lib, vol, src, tgt, img, lookup=lookup, align=align)
    assert rel_err(out, g["out_f32"].reshape(B, N)) < FWD_TOL


def test_voxel_count_equals_oracle(emu_lib):
    for name in ("siddon_sum", "siddon_sum_oblique"):
        g, vol, src, tgt, img, B, N = load(name)
        _, _, cnt = emu_siddon(emu_lib, vol, src, tgt, img, cnt=True)
        ref = oracle.siddon(vol, src, tgt, img, count_voxels=True)["n_inside"]
        # identical up to zero-length segments at exact ties
        assert np.abs(cnt - ref).max() <= 1
        assert abs(int(cnt.sum()) - int(ref.sum())) <= 0.002 * ref.sum() + 2


def test_tile_map_is_result_invariant(emu_lib):
    """Any detector tiling must give bit-identical images (it only permutes lanes)."""
    rng = np.random.default_rng(0)
    vol = f32(rng.random((10, 12, 9)))
    H, W = 12, 20  # not multiples of every tile shape -> padding lanes
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    tgt = np.stack([ii.ravel() * 0.9 - 0.5, np.full(H * W, 30.0), jj.ravel() * 0.45], -1)
    tgt = f32(tgt[None])
    src = f32(np.array([[[5.0, -40.0, 4.0]]]))
    img = f32(np.linalg.norm(tgt - src, axis=-1))
    base, _, _ = emu_siddon(emu_lib, vol, src, tgt, img)
    for th, tw in [(64, 1), (32, 2), (16, 4), (8, 8), (4, 16), (2, 32), (1, 64)]:
        out, _, _ = emu_siddon(emu_lib, vol, src, tgt, img, det=(H, W), tile=(th, tw))
        assert np.array_equal(out, base), (th, tw)


def test_siddon_channels(emu_lib):
    g, vol, src, tgt, img, B, N = load("siddon_mask")
    labels = np.ascontiguousarray(g["mask"].astype(np.uint8))
    C = int(labels.max()) + 1
    out = np.full((B, C, N), np.nan, np.float32)
    emu_lib.call("ddrr_siddon_forward_channels", P(vol), P(labels), *vol.shape, P(src),
                 src.shape[1], P(tgt), P(img), B, N, C, 0.5, 1e-8, 0, 0, 1, 64, P(out), None)
    assert rel_err(out, g["out_f32"]) < FWD_TOL
    plain, _, _ = emu_siddon(emu_lib, vol, src, tgt, img)
    assert rel_err(out.sum(1), plain) < 1e-5  # channels add up to the DRR


def test_siddon_channels_on_bricks(emu_lib):
    """The channel render on the volume-stationary bricks (value | label packed in one staged
    word, brick_step.h) against the reference fixture ..."""
    g, vol, src, tgt, img, B, N = load("siddon_mask")
    labels = np.ascontiguousarray(g["mask"].astype(np.uint8))
    C = int(labels.max()) + 1
    out = np.full((B, C, N), np.nan, np.float32)
    emu_lib.call("ddrr_siddon_forward_channels_bricks", P(vol), P(labels), *vol.shape, P(src),
                 P(tgt), P(img), B, 4, N // 4, C, 0.5, 1e-8, P(out), None, None)
    assert rel_err(out, g["out_f32"]) < FWD_TOL
    # fewer channels than labels: the others are dropped (the per-ray kernel does the same)
    out3 = np.full((B, 3, N), np.nan, np.float32)
    emu_lib.call("ddrr_siddon_forward_channels_bricks", P(vol), P(labels), *vol.shape, P(src),
                 P(tgt), P(img), B, 4, N // 4, 3, 0.5, 1e-8, P(out3), None, None)
    # (labels without a channel are staged as value 0 under label 0: channel 0's runs are cut
    # differently, i.e. summed in a different order)
    assert rel_err(out3, out[:, :3]) < 1e-6


@pytest.mark.parametrize("kind", ["noise", "phantom"])
def test_siddon_channels_on_bricks_vs_oracle(emulated_ops, kind):
    """... and, on a volume of several bricks with up to 256 labels, against the oracle's
    channel render and the per-ray channel kernel; the channels add up to the plain render.
    The packed word keeps a 16-bit mantissa of the value: 2^-17 relative per voxel."""
    import torch

    from diffdrr_amd import DRR, convert
    from diffdrr_amd.data import synthetic_subject

    ops = emulated_ops
    D, H, W = (40, 70, 36), 24, 31
    drr = DRR(synthetic_subject(D, kind=kind, seed=5), sdd=600.0, height=H, width=W, delx=3.0)
    rng = np.random.default_rng(11)
    blocks = rng.integers(0, 256, size=(5, 9, 5)).astype(np.uint8)  # 8^3-voxel label blocks
    labels = np.kron(blocks, np.ones((8, 8, 8), np.uint8))[:D[0], :D[1], :D[2]].copy()
    rot = torch.tensor([[0.3, 0.2, -0.1], [1.5, 0.1, 0.0], [0.0, 1.45, 0.2]])
    xyz = torch.tensor([[5.0, 420.0, -3.0], [0.0, 400.0, 0.0], [2.0, 380.0, 1.0]])
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    V, lab = drr.density, torch.from_numpy(labels)
    C = 256
    ch = ops.siddon_forward_channels_bricks(V, lab, C, s, t, L, (H, W)).numpy()
    per_ray = ops.siddon_forward_channels(V, lab, C, s, t, L).numpy()
    ref = oracle.siddon_channels(V.numpy(), labels.astype(np.float32), s.numpy(), t.numpy(),
                                 L.numpy(), n_channels=C)
    assert rel_err(ch, ref) < FWD_TOL
    assert rel_err(ch, per_ray) < 3e-5
    plain, _ = ops.siddon_forward_bricks(V, s, t, L, (H, W))
    assert rel_err(ch.sum(1), plain.numpy()) < 3e-5
    # a channel is exactly zero wherever the per-ray render says no ray meets the label
    assert np.all(ch[per_ray == 0] == 0)


TRI = [
    ("trilinear_global_range", 41, None, 0.5), ("trilinear_explicit_range", 64, (0.31, 0.77), 0.5),
    ("trilinear_oblique", 50, None, 0.5), ("trilinear_shift0", 40, None, 0.0),
]


@pytest.mark.parametrize("name,npts,rng,shift", TRI)
def test_trilinear_forward_and_gradients(emu_lib, name, npts, rng, shift):
    g, vol, src, tgt, img, B, N = load(name)
    if rng is None:
        lo, hi = oracle.alpha_minmax(src, tgt, vol.shape, voxel_shift=shift)
        rng = (lo.min(), hi.max())
    am, aM = np.array([rng[0]], np.float32), np.array([rng[1]], np.float32)
    out = np.zeros((B, N), np.float32)
    emu_lib.call("ddrr_trilinear_forward", P(vol), *vol.shape, P(src), src.shape[1], P(tgt),
                 P(img), B, N, shift, 1e-8, npts, P(am), P(aM), 0, 0, 0, 0, 0, 1, 64, P(out), None)
    assert rel_err(out, g["out_f32"].reshape(B, N)) < FWD_TOL

    go = f32(g["grad_out_f32"].reshape(B, N))
    ref = oracle.trilinear(vol.astype(np.float64), src.astype(np.float64), tgt.astype(np.float64),
                           img.astype(np.float64), n_points=npts, alphamin=float(am[0]),
                           alphamax=float(aM[0]), voxel_shift=shift,
                           grad_out=go.astype(np.float64), want_volume_grad=True)
    gs, gt = np.zeros((B, N, 3), np.float32), np.zeros((B, N, 3), np.float32)
    gi, ga = np.zeros((B, N), np.float32), np.zeros((B, N, 2), np.float32)
    gv = np.zeros_like(vol)
    emu_lib.call("ddrr_trilinear_backward", P(vol), *vol.shape, P(src), src.shape[1], P(tgt),
                 P(img), P(go), B, N, shift, 1e-8, npts, P(am), P(aM), 0, 0, 0, 0, 1, 64, P(gs),
                 P(gt), P(gi), P(ga), P(gv), None)
    assert rel_err(gs.sum(1, keepdims=True), ref["g_source"]) < GRAD_TOL
    assert rel_err(gt, ref["g_target"]) < GRAD_TOL
    assert rel_err(gi, ref["g_img"].reshape(B, N)) < GRAD_TOL
    assert rel_err(gv, ref["g_volume"]) < GRAD_TOL
    assert abs(ga[..., 0].sum() / ref["g_alphamin"] - 1) < GRAD_TOL
    assert abs(ga[..., 1].sum() / ref["g_alphamax"] - 1) < GRAD_TOL
    assert rel_err(gv, g["g_volume_f64"]) < GRAD_TOL  # and the reference's own autograd


def test_trilinear_nearest_max(emu_lib):
    g, vol, src, tgt, img, B, N = load("trilinear_nearest_max")
    lo, hi = oracle.alpha_minmax(src, tgt, vol.shape)
    am, aM = np.array([lo.min()], np.float32), np.array([hi.max()], np.float32)
    out = np.zeros((B, N), np.float32)
    emu_lib.call("ddrr_trilinear_forward", P(vol), *vol.shape, P(src), src.shape[1], P(tgt),
                 P(img), B, N, 0.5, 1e-8, 33, P(am), P(aM), 1, 1, 0, 0, 0, 1, 64, P(out), None)
    assert rel_err(out, g["out_f32"].reshape(B, N)) < FWD_TOL


# ---------------------------------------------------- pose classes, both Siddon walks

SLAB_POSES = [
    ("base", [0.0, 0.0, 0.0], [0.0, 850.0, 0.0]),
    ("rotZ", [0.3, 0.0, 0.0], [0.0, 850.0, 0.0]),
    ("tilt", [0.0, 0.4, 0.0], [0.0, 850.0, 0.0]),
    ("inplane", [0.0, 0.0, 0.5], [0.0, 850.0, 0.0]),
    ("oblique", [0.7, -0.5, 0.6], [20.0, 830.0, -15.0]),
    ("lateral", [1.5, 0.1, 0.2], [0.0, 850.0, 0.0]),
    ("axial", [0.0, 1.45, 0.0], [0.0, 850.0, 0.0]),
    ("diag45", [0.79, 0.0, 0.0], [0.0, 850.0, 0.0]),
    ("inside", [0.2, 0.1, 0.0], [3.0, 20.0, -4.0]),   # source inside the volume
    ("behind", [0.1, 0.0, 0.1], [0.0, -900.0, 0.0]),  # volume behind the source: whole line
]


@pytest.mark.parametrize("D,H,W,delx", [(64, 40, 40, 3.0), (48, 70, 33, 2.0)])
def test_generic_and_brick_walks_vs_oracle(emulated_ops, D, H, W, delx):
    """The per-ray walk (siddon_core.h) and the volume-stationary brick walk (brick_step.h)
    on every pose class (beam along x / y / z, oblique, source inside the volume, volume
    behind the source) and on a volume that is not a multiple of the brick edge: images
    against the fp32 oracle, backward records against the fp64 oracle."""
    import torch

    from diffdrr_amd import DRR, convert
    from diffdrr_amd.data import synthetic_subject

    ops = emulated_ops
    drr = DRR(synthetic_subject(D, kind="noise", seed=0), sdd=1020.0, height=H, width=W,
              delx=delx)
    rot = torch.tensor([p[1] for p in SLAB_POSES])
    xyz = torch.tensor([p[2] for p in SLAB_POSES])
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    V = drr.density
    gen, aux_gen, _ = ops.siddon_forward(V, s, t, L, want_aux=True)
    a32 = (V.numpy(), s.numpy(), t.numpy(), L.numpy())
    ref = oracle.siddon(*a32)["out"].reshape(gen.shape)
    # volume-stationary brick kernel: per-brick pieces of every ray, added up
    outb, auxb = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=True)
    outb0, none = ops.siddon_forward_bricks(V, s, t, L, (H, W))
    # (forward only walks with accumulated chord-relative alphas, brick_step.h step_walk_fwd; with
    # the record every alpha is the reference's quotient: two fp32 evaluations of one integral)
    assert none is None and rel_err(outb0.numpy(), outb.numpy()) < 1e-5
    for b, (name, _, _) in enumerate(SLAB_POSES):
        assert rel_err(outb0[b].numpy(), ref[b]) < 5e-5, name
    for b, (name, _, _) in enumerate(SLAB_POSES):
        assert rel_err(gen[b].numpy(), ref[b]) < 5e-5, name
        assert rel_err(outb[b].numpy(), ref[b]) < 5e-5, name
        # both walks evaluate every crossing as the reference's quotient (k - shift - s) / d
        # (the brick walk via the first plane ahead of each brick entry); the sums are grouped
        # differently (per brick), which is what is left of the difference
        assert rel_err(outb[b].numpy(), gen[b].numpy()) < 2e-5, name
        planes = ops.record_planes(auxb, *gen.shape)  # blocked record -> (5, B, N)
        assert rel_err(planes[0, b].numpy(), aux_gen[b, :, 0].numpy()) < 2e-5, name
    # the planar record gives the same ray gradients as the generic walk's record
    go = torch.rand(gen.shape, generator=torch.Generator().manual_seed(3))
    gsb, gtb, gib = ops.siddon_backward_rays(auxb, go, s, t, L)
    gsg, gtg, gig = ops.siddon_backward_rays(aux_gen, go, s, t, L)
    assert rel_err(gib.numpy(), gig.numpy()) < 2e-5
    # 16-bit block-quantised bricks (brick_step.h q16_*): the same walk over q = 0 .. 65535 read
    # as denormal floats, alphas scaled by 2^64; |V - (vmin + q step)| <= range / 131070 per voxel
    outq, auxq = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=True, storage="q16")
    outq0, _ = ops.siddon_forward_bricks(V, s, t, L, (H, W), storage="q16")
    assert rel_err(outq0.numpy(), outq.numpy()) < 1e-5
    for b, (name, _, _) in enumerate(SLAB_POSES):
        assert rel_err(outq0[b].numpy(), ref[b]) < 5e-5, name
    gsq, gtq, giq = ops.siddon_backward_rays(auxq, go, s, t, L)
    for b, (name, _, _) in enumerate(SLAB_POSES):
        assert rel_err(outq[b].numpy(), ref[b]) < 5e-5, name
        assert rel_err(outq[b].numpy(), outb[b].numpy()) < 1e-5, name
        # the record of the quantised volume: the same crossings, voxel values 1e-5 apart
        assert rel_err(giq[b].numpy(), gib[b].numpy()) < 1e-5, name
        same = (gtq[b] - gtb[b]).abs().amax(-1) <= 1e-3 * gtb[b].abs().max()
        assert same.float().mean().item() > 0.995, name
        assert rel_err(gtq[b].sum(0).numpy(), gtb[b].sum(0).numpy()) < 2e-3, name
    # yardstick: the fp64 oracle, allowance: what the reference's own fp32 arithmetic (the fp32
    # oracle) loses against it.  Single rays differ where a crossing pair ties in fp32 and is
    # attributed to different axes; the walks' alphas are within an ulp of the reference's
    # quotient, so they flip where the reference's fp32 flips.
    o32 = oracle.siddon(*a32, grad_out=go.numpy())["g_target"]
    o64 = oracle.siddon(*(x.astype(np.float64) for x in a32),
                        grad_out=go.numpy().astype(np.float64))["g_target"]
    for b, (name, _, _) in enumerate(SLAB_POSES):
        scale = np.abs(o64[b]).max()
        close = lambda g: float((np.abs(np.asarray(g[b]) - o64[b]).max(-1) <= 1e-3 * scale).mean())  # noqa
        for mine in (gtb, gtg):
            assert close(mine) > 0.9, name
            if name in ("base", "diag45"):
                continue  # source on a symmetry plane: exact ties everywhere, attribution arbitrary
            # (rotZ / tilt: the detector's central column looks down the volume's central edge,
            # x- and y-planes tie exactly there: ~1 % of the rays)
            assert close(mine) >= close(o32) - 0.02, name
            # per-pose sums (what the pose gradient is made of); where exact ties dominate
            # the sum the attribution is a convention the two walks share
            assert rel_err(mine[b].sum(0).numpy(), o64[b].sum(0)) < \
                2 * rel_err(o32[b].sum(0), o64[b].sum(0)) + 2e-3 or \
                rel_err(gtb[b].sum(0).numpy(), gtg[b].sum(0).numpy()) < 2e-3, name


def test_gliding_rays_vs_fp64(emulated_ops):
    """Rays that glide along a voxel plane (one direction component ~1e-5 of the others,
    position within 1e-5 voxel of the plane): a position error of 1e-5 voxel is an alpha
    error of any size there.  The walks evaluate every crossing as the reference's quotient
    (k - shift - s) / d from the integer plane index and pick entry cells by the order of
    those alphas, never from positions alone (found at 512^3: one pixel off by 4e-4 with
    alpha = fma(k, 1/d, c)).  Yardstick: the fp64 oracle -- the reference's OWN fp32
    arithmetic (the fp32 oracle) is ~3e-2 off on these rays, because it rounds segment
    midpoints to voxels."""
    import torch

    ops = emulated_ops
    g = torch.Generator().manual_seed(11)
    D, H, W = 70, 16, 64
    V = torch.rand(D, D, D, generator=g)
    n = H * W
    axis = torch.arange(n) % 3                                   # the gliding axis
    plane = torch.randint(1, D - 1, (n,), generator=g).float()   # glide along this plane
    s = torch.empty(n, 3).uniform_(-300.0, -150.0, generator=g)
    t = torch.empty(n, 3).uniform_(150.0, 300.0, generator=g)
    flip = torch.rand(n, 3, generator=g) < 0.5
    s, t = torch.where(flip, t, s), torch.where(flip, s, t)
    off = (torch.rand(n, generator=g) - 0.5) * 4e-5              # distance from the plane
    tilt = (torch.rand(n, generator=g) - 0.5) * 2e-2             # total drift along the ray
    idx = torch.arange(n)
    s[idx, axis] = plane - 0.5 + off - tilt / 2
    t[idx, axis] = plane - 0.5 + off + tilt / 2
    # every ray its own "pose" with a 2x2 detector of identical targets (the brick entry point
    # takes one source per pose and a detector grid)
    src = s.view(n, 1, 3).contiguous()
    tgt = t.view(n, 1, 3).expand(n, 4, 3).contiguous()
    L = torch.ones(n, 4)
    a64 = (V.numpy().astype(np.float64), src.numpy().astype(np.float64),
           tgt.numpy()[:, :1].astype(np.float64), np.ones((n, 1)))
    ref64 = oracle.siddon(*a64)["out"].reshape(n)
    assert np.abs(ref64).max() > 0
    one, _, _ = ops.siddon_forward(V, src, tgt, L)
    bricks, _ = ops.siddon_forward_bricks(V, src, tgt, L, (2, 2))  # 32^3 bricks: 27 of them
    for k in range(4):
        assert rel_err(one.numpy()[:, k], ref64) < 1e-5
        assert rel_err(bricks.numpy()[:, k], ref64) < 1e-5


def test_volume_gradient_bricks_equals_rewalk(emulated_ops):
    """Volume gradient through the volume-stationary brick kernel (LDS accumulation,
    stored once) against the per-ray re-walk with scattered adds, on every pose class of
    SLAB_POSES and a volume that is not a multiple of the brick size."""
    import torch

    from diffdrr_amd import DRR, convert
    from diffdrr_amd.data import make_subject

    ops = emulated_ops
    H, W = 30, 26
    g = torch.Generator().manual_seed(2)
    drr = DRR(make_subject(torch.rand(70, 40, 33, generator=g), spacing=(1.0, 1.0, 1.0)),
              sdd=1020.0, height=H, width=W, delx=3.0)
    rot = torch.tensor([p[1] for p in SLAB_POSES])
    xyz = torch.tensor([p[2] for p in SLAB_POSES])
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    go = torch.rand(len(SLAB_POSES), H * W, generator=g)
    ref = ops.siddon_backward_volume(drr.density, s, t, L, go)
    out = ops.siddon_backward_volume_bricks(drr.density.shape, s, t, L, go, (H, W))
    assert out.shape == drr.density.shape
    assert rel_err(out.numpy(), ref.numpy()) < 1e-5
    # adjointness: <g_volume, V> = <grad_out, render(V)>
    img = ops.siddon_forward_bricks(drr.density, s, t, L, (H, W))[0]
    lhs, rhs = (out * drr.density).sum().item(), (go * img).sum().item()
    assert abs(lhs - rhs) < 1e-4 * abs(rhs)


def test_trilinear_bricks_equal_per_ray_march(emulated_ops):
    """Volume-stationary trilinear kernels (csrc/tri_brick.h) against the per-ray marcher on
    every pose class of SLAB_POSES, with a volume that is not a multiple of the brick edge
    and a batch-global alpha range: same image, same volume gradient; the Trilinear module
    routes detector-grid calls through them."""
    import torch

    from diffdrr_amd import DRR, Trilinear, convert
    from diffdrr_amd.data import make_subject
    from diffdrr_amd.renderers import get_alpha_minmax

    ops = emulated_ops
    H, W, P = 22, 26, 90
    g = torch.Generator().manual_seed(4)
    drr = DRR(make_subject(torch.rand(45, 64, 33, generator=g), spacing=(1.0, 1.0, 1.0)),
              sdd=1020.0, height=H, width=W, delx=3.0)
    rot = torch.tensor([p[1] for p in SLAB_POSES])
    xyz = torch.tensor([p[2] for p in SLAB_POSES])
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
        lo, hi = get_alpha_minmax(s, t, torch.tensor(drr.density.shape), 0.5, 1e-8)
        amin, amax = lo.min().reshape(1).contiguous(), hi.max().reshape(1).contiguous()
    V = drr.density
    ref = ops.trilinear_forward(V, s, t, L, amin, amax, n_points=P)
    out = ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (H, W), n_points=P)
    for b, (name, _, _) in enumerate(SLAB_POSES):
        assert rel_err(out[b].numpy(), ref[b].numpy()) < 2e-6, name
    go = torch.rand(len(SLAB_POSES), H * W, generator=g)
    gref = ops.trilinear_backward(V, s, t, L, go, amin, amax, n_points=P, want_rays=False,
                                  want_img=False, want_alpha=False, want_volume=True)["g_volume"]
    gout = ops.trilinear_backward_volume_bricks(V.shape, s, t, L, go, amin, amax, (H, W),
                                                n_points=P)
    assert rel_err(gout.numpy(), gref.numpy()) < 1e-5
    # through the module: forward + volume and ray gradients
    res = {}
    for bricks in (True, False):
        m = Trilinear()
        m.detector_shape, m.use_bricks = (H, W), bricks
        Vg, sg, tg = V.clone().requires_grad_(), s.clone().requires_grad_(), t.clone().requires_grad_()
        o = m(Vg, sg, tg, L.unsqueeze(1), n_points=P)
        (o.squeeze(1) * go).sum().backward()
        res[bricks] = (o.detach(), Vg.grad, sg.grad, tg.grad)
    for x, y in zip(res[True], res[False]):
        assert rel_err(x.numpy(), y.numpy()) < 1e-5


def test_trilinear_channels(emu_lib):
    """Trilinear mask_to_channels (renderers.py:242-252) against the reference fixture."""
    g, vol, src, tgt, img, B, N = load("trilinear_mask")
    labels = np.ascontiguousarray(g["mask"].astype(np.uint8))
    C = int(labels.max()) + 1
    lo, hi = oracle.alpha_minmax(src, tgt, vol.shape)
    am, aM = np.array([lo.min()], np.float32), np.array([hi.max()], np.float32)
    out = np.full((B, C, N), np.nan, np.float32)
    emu_lib.call("ddrr_trilinear_forward_channels", P(vol), P(labels), *vol.shape, P(src),
                 src.shape[1], P(tgt), P(img), B, N, C, 0.5, 1e-8, 40, P(am), P(aM), 0, 0, 0, 1, 64,
                 P(out), None)
    assert out.shape == g["out_f32"].shape
    assert rel_err(out, g["out_f32"]) < FWD_TOL
    plain = np.zeros((B, N), np.float32)
    emu_lib.call("ddrr_trilinear_forward", P(vol), *vol.shape, P(src), src.shape[1], P(tgt),
                 P(img), B, N, 0.5, 1e-8, 40, P(am), P(aM), 0, 0, 0, 0, 0, 1, 64, P(plain), None)
    assert rel_err(out.sum(1), plain) < 1e-5  # channels add up to the DRR


def test_siddon_segments_equal_oracle_terms(emu_lib):
    """ddrr_siddon_segments against the oracle's restatement of the (B, N, M-1) tensor the
    reference holds before `reduce` (renderers.py:71): position by position."""
    g, vol, src, tgt, img, B, N = load("siddon_sum_oblique")
    M1 = sum(vol.shape) + 2
    terms = np.full((B, M1, N), np.nan, np.float32)
    emu_lib.call("ddrr_siddon_segments", P(vol), *vol.shape, P(src), src.shape[1], P(tgt), P(img),
                 B, N, 0.5, 1e-8, P(terms), None)
    ref, _ = oracle.siddon_segments(vol, src, tgt, img)
    assert ref.shape == (B, N, M1)
    assert rel_err(terms.transpose(0, 2, 1), ref) < 1e-5


def test_channel_backward_on_bricks_vs_oracle(emulated_ops):
    """The ray / img backward of the channel render on the volume-stationary bricks (host
    emulation of step_walk_weighted) against the fp64 oracle and the per-ray channel backward."""
    conftest.check_channel_backward_on_bricks(emulated_ops, "cpu")
