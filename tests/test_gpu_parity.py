"""Parity of the HIP kernels on a real MI355X, through the C ABI
(diffdrr_amd.ops -> libdiffdrr_hip.so):
  * against the reference goldens (tests/golden) and the C oracle on the same
    inputs, at sizes the oracle finishes in seconds;
  * at BASELINE.json's full sizes (512^3 volume, 256^2 detector) through
    size-independent properties: analytic chord lengths, linearity in the
    volume, adjointness of the volume gradient, tiling / batch invariance.
Tolerances: forward image-normalised error <= 1e-4 vs the reference fp32
(north star); gradients <= 1e-3."""
import numpy as np
import pytest
import torch

import oracle
import conftest
from conftest import golden, rel_err
from diffdrr_amd import DRR, Siddon, Trilinear, convert, ops
from diffdrr_amd.data import Image, Subject, make_subject, synthetic_subject

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 1e-4, 1e-3


def dev_inputs(g, gpu):
    vol = torch.from_numpy(g["volume"].astype(np.float32)).to(gpu)
    src = torch.from_numpy(g["source"].astype(np.float32)).to(gpu)
    tgt = torch.from_numpy(g["target"].astype(np.float32)).to(gpu)
    B, N, _ = tgt.shape
    img = torch.from_numpy(g["img_f32"].reshape(B, 1, N).astype(np.float32)).to(gpu)
    return vol, src, tgt, img


# ----------------------------------------------------------- golden fixtures

@pytest.mark.parametrize("name,kw", [
    ("siddon_sum", {}), ("siddon_sum_oblique", {}), ("siddon_max", {"reducefn": "max"}),
    ("siddon_per_ray_source", {}), ("siddon_shift0", {"voxel_shift": 0.0}),
    ("siddon_stopgrad", {"stop_gradients_through_grid_sample": True}),
])
def test_siddon_golden(gpu, name, kw):
    g = golden(name)
    vol, src, tgt, img = (t.requires_grad_() for t in dev_inputs(g, gpu))
    out = Siddon(**kw)(vol, src, tgt, img)
    assert out.shape == g["out_f32"].shape
    assert rel_err(out.detach().cpu().numpy(), g["out_f32"]) < FWD_TOL
    ref_err = rel_err(g["out_f32"], g["out_f64"])
    assert rel_err(out.detach().cpu().numpy(), g["out_f64"]) < 2 * ref_err + 5e-6
    go = torch.from_numpy(g["grad_out_f32"]).to(gpu)
    grads = torch.autograd.grad(out, (src, tgt, img, vol), go, allow_unused=True)
    assert rel_err(grads[0].cpu().numpy(), g["g_source_f64"]) < GRAD_TOL
    assert rel_err(grads[1].cpu().numpy(), g["g_target_f64"]) < GRAD_TOL
    if name == "siddon_stopgrad":
        assert grads[2] is None and grads[3] is None  # like the reference (renderers.py:63-65)
    else:
        assert rel_err(grads[2].cpu().numpy(), g["g_img_f64"]) < GRAD_TOL
        assert rel_err(grads[3].cpu().numpy(), g["g_volume_f64"]) < GRAD_TOL


def test_siddon_special_rays_golden(gpu):
    g = golden("siddon_special_rays")
    vol, src, tgt, img = dev_inputs(g, gpu)
    out = Siddon()(vol, src, tgt, img).cpu().numpy()
    assert rel_err(out, g["out_f64"]) < 1e-5
    assert out[5, 0, 0] == 0 and out[6, 0, 0] == 0


@pytest.mark.parametrize("name,ctor,call", [
    ("siddon_bilinear", {"mode": "bilinear"}, {}),
    ("siddon_align_corners", {}, {"align_corners": True}),
])
def test_siddon_generic_lookup_golden(gpu, name, ctor, call):
    g = golden(name)
    vol, src, tgt, img = dev_inputs(g, gpu)
    out = Siddon(**ctor)(vol, src, tgt, img, **call)
    assert rel_err(out.cpu().numpy(), g["out_f32"]) < FWD_TOL


def test_siddon_mask_golden(gpu):
    g = golden("siddon_mask")
    vol, src, tgt, img = dev_inputs(g, gpu)
    mask = torch.from_numpy(g["mask"]).to(gpu)
    out = Siddon()(vol, src, tgt, img, mask=mask)
    assert out.shape == g["out_f32"].shape
    assert rel_err(out.cpu().numpy(), g["out_f32"]) < FWD_TOL
    plain = Siddon()(vol, src, tgt, img)
    assert rel_err(out.sum(1, keepdim=True).cpu().numpy(), plain.cpu().numpy()) < 1e-5


def test_siddon_callable_reducefn_golden(gpu):
    """Callable reducefn over the materialised per-segment tensor (ddrr_siddon_segments /
    _backward) against the reference's autograd (top-6 sum, introduction.ipynb:506-529)."""
    def topk_sum(img):
        return img.sort(descending=True).values[..., :6].sum(dim=-1)

    g = golden("siddon_callable")
    vol, src, tgt, img = (t.requires_grad_() for t in dev_inputs(g, gpu))
    out = Siddon(reducefn=topk_sum)(vol, src, tgt, img)
    assert rel_err(out.detach().cpu().numpy(), g["out_f32"]) < FWD_TOL
    go = torch.from_numpy(g["grad_out_f32"]).to(gpu)
    grads = torch.autograd.grad(out, [src, tgt, img, vol], go)
    for name, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.cpu().numpy(), g[name + "_f64"]) < GRAD_TOL, name


def test_trilinear_callable_reducefn_golden(gpu):
    """Callable reducefn of the marcher (ddrr_trilinear_samples / _backward) against the
    reference's autograd (top-6 sum over the per-sample tensor)."""
    def topk_sum(img):
        return img.sort(descending=True).values[..., :6].sum(dim=-1)

    g = golden("trilinear_callable")
    vol, src, tgt, img = (t.requires_grad_() for t in dev_inputs(g, gpu))
    out = Trilinear(reducefn=topk_sum)(vol, src, tgt, img, n_points=40)
    assert rel_err(out.detach().cpu().numpy(), g["out_f32"]) < FWD_TOL
    go = torch.from_numpy(g["grad_out_f32"]).to(gpu)
    grads = torch.autograd.grad(out, [src, tgt, img, vol], go)
    for name, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.cpu().numpy(), g[name + "_f64"]) < GRAD_TOL, name


@pytest.mark.parametrize("name,ctor,call", [
    ("siddon_bilinear", {"mode": "bilinear"}, {}),
    ("siddon_align_corners", {}, {"align_corners": True}),
])
def test_siddon_midpoint_gradients_golden(gpu, name, ctor, call):
    """ddrr_siddon_backward_midpoint against the reference's autograd for the midpoint lookups."""
    g = golden(name)
    vol, src, tgt, img = (t.requires_grad_() for t in dev_inputs(g, gpu))
    out = Siddon(**ctor)(vol, src, tgt, img, **call)
    assert rel_err(out.detach().cpu().numpy(), g["out_f32"]) < FWD_TOL
    go = torch.from_numpy(g["grad_out_f32"]).to(gpu)
    grads = torch.autograd.grad(out, [src, tgt, img, vol], go)
    for k, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.cpu().numpy(), g[k + "_f64"]) < GRAD_TOL, k


def test_siddon_mask_gradients_golden(gpu):
    """mask_to_channels backward (ddrr_siddon_backward_channels) against the reference's
    autograd through its scatter_add (renderers.py:77-89): grad_out is (B, C, N)."""
    g = golden("siddon_mask")
    vol, src, tgt, img = (t.requires_grad_() for t in dev_inputs(g, gpu))
    mask = torch.from_numpy(g["mask"]).to(gpu)
    out = Siddon()(vol, src, tgt, img, mask=mask)
    go = torch.from_numpy(g["grad_out_f32"]).to(gpu)
    grads = torch.autograd.grad(out, [src, tgt, img, vol], go)
    for name, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.cpu().numpy(), g[name + "_f64"]) < GRAD_TOL, name


def test_empty_bricks_are_skipped_without_changing_anything(gpu):
    """A CT is mostly air, which the HU -> density transform maps to exactly 0 (reference
    data.py:214-227): the brick kernels do not look at the candidates of a brick whose staged
    voxels are all zero.  Volume with a dense blob, a slab of -0.0 and zeros elsewhere: images,
    backward record, channels and the marcher against the per-ray kernels, which know no such
    shortcut, and against the oracle."""
    from diffdrr_amd import ops as O

    D, H, W = (96, 128, 70), 40, 52
    g = torch.Generator().manual_seed(3)
    vol = torch.zeros(D)
    vol[50:90, 20:70, 30:66] = torch.rand(40, 50, 36, generator=g)
    vol[:, 100:104, :] = -0.0
    labels = torch.zeros(D, dtype=torch.uint8)
    labels[40:, :, :] = 3
    labels[60:80, 30:50, :] = 7
    drr = DRR(make_subject(vol, (1.0, 1.0, 1.0), "AP", None), sdd=600.0, height=H, width=W,
              delx=3.0).to(gpu)
    # (no pose on a symmetry plane: exact ties are attributed by convention, test_emu_vs_oracle)
    rot = torch.tensor([[0.3, 0.2, -0.1], [1.2, 0.1, 0.0], [-0.4, 0.05, 0.6]], device=gpu)
    xyz = torch.tensor([[5.0, 400.0, -3.0], [0.0, 380.0, 0.0], [-7.0, 420.0, 4.0]], device=gpu)
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    V, lab = drr.density, labels.to(gpu)
    out, aux = O.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=True)
    gen, aux_gen, _ = O.siddon_forward(V, s, t, L, want_aux=True)
    ref = oracle.siddon(V.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy(), L.cpu().numpy())["out"]
    assert rel_err(out.cpu().numpy(), ref.reshape(out.shape)) < FWD_TOL
    assert rel_err(out.cpu().numpy(), gen.cpu().numpy()) < 2e-5
    go = torch.rand(out.shape, device=gpu, generator=torch.Generator(gpu).manual_seed(1))
    gb = O.siddon_backward_rays(aux, go, s, t, L)
    gg = O.siddon_backward_rays(aux_gen, go, s, t, L)
    for a, b in zip(gb, gg):
        assert rel_err(a.sum(1).cpu().numpy(), b.sum(1).cpu().numpy()) < 2e-3
    ch = O.siddon_forward_channels_bricks(V, lab, 8, s, t, L, (H, W))
    chr_ = O.siddon_forward_channels(V, lab, 8, s, t, L)
    assert rel_err(ch.cpu().numpy(), chr_.cpu().numpy()) < 3e-5
    a0, a1 = O.trilinear_alpha_range(s, t, V.shape)
    tri = O.trilinear_forward_bricks(V, s, t, L, a0, a1, (H, W), n_points=200)
    trr = O.trilinear_forward(V, s, t, L, a0, a1, n_points=200)
    assert rel_err(tri.cpu().numpy(), trr.cpu().numpy()) < 2e-5
    assert out.abs().max() > 1.0  # (the scene is not empty)


def test_alpha_range_kernel(gpu):
    """ddrr_trilinear_alpha_range against the tensor ops it replaces (reference
    renderers.py:124-140, 220-223) on the device: oblique rays, per-ray sources, rays parallel
    to an axis, all rays missing (negative alphamax), 300k rays (many blocks)."""
    from diffdrr_amd.renderers import get_alpha_minmax

    g = torch.Generator().manual_seed(0)
    dims = (120, 100, 140)
    for trial, (B, N) in enumerate([(3, 50), (2, 777), (1, 300000), (4, 64), (2, 100)]):
        src = (torch.rand(B, 1 if trial % 2 else N, 3, generator=g) * 600 - 300).to(gpu)
        tgt = (torch.rand(B, N, 3, generator=g) * 400 - 100).to(gpu)
        if trial == 3:
            tgt[:, :7, 0] = src[:, :1, 0] if src.shape[1] == 1 else src[:, :7, 0]
        if trial == 4:
            tgt = tgt + 5000.0
        lo, hi = get_alpha_minmax(src, tgt, torch.tensor(dims, device=gpu).float(), 0.5, 1e-8)
        a0, a1 = ops.trilinear_alpha_range(src, tgt, dims)
        assert abs(a0.item() - lo.min().item()) <= 1e-7 * max(1.0, abs(lo.min().item())), trial
        assert abs(a1.item() - hi.max().item()) <= 1e-7 * max(1.0, abs(hi.max().item())), trial


@pytest.mark.parametrize("D,kind", [((96, 70, 133), "phantom"), ((64, 64, 61), "noise")])
def test_siddon_channels_on_bricks_vs_oracle(gpu, D, kind):
    """mask_to_channels on the volume-stationary kernel (ddrr_siddon_forward_channels_bricks:
    value | label packed in one staged LDS word) against the oracle's channel render
    (renderers.py:77-89), the per-ray channel kernel and the plain render, on volumes of several
    bricks -- one with Dz a multiple of nothing (the scalar staging path), 256 labels, labels
    at an odd address."""
    from diffdrr_amd import DRR, convert, ops
    from diffdrr_amd.data import synthetic_subject

    H, W = 50, 37
    drr = DRR(synthetic_subject(D, kind=kind, seed=5), sdd=700.0, height=H, width=W,
              delx=3.0).to(gpu)
    rng = np.random.default_rng(11)
    blocks = rng.integers(0, 256, size=tuple((d + 7) // 8 for d in D)).astype(np.uint8)
    labels = np.kron(blocks, np.ones((8, 8, 8), np.uint8))[:D[0], :D[1], :D[2]].copy()
    rot = torch.tensor([[0.3, 0.2, -0.1], [1.5, 0.1, 0.0], [0.0, 1.45, 0.2], [0.0, 0.0, 0.0]],
                       device=gpu)
    xyz = torch.tensor([[5.0, 480.0, -3.0], [0.0, 460.0, 0.0], [2.0, 440.0, 1.0],
                        [0.0, 450.0, 0.0]], device=gpu)
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    V = drr.density
    C = 256
    store = torch.zeros(labels.size + 1, dtype=torch.uint8, device=gpu)
    store[1:] = torch.from_numpy(labels).to(gpu).flatten()
    for lab in (torch.from_numpy(labels).to(gpu), store[1:].view(*D)):
        ch = ops.siddon_forward_channels_bricks(V, lab, C, s, t, L, (H, W)).cpu().numpy()
        per_ray = ops.siddon_forward_channels(V, lab, C, s, t, L).cpu().numpy()
        ref = oracle.siddon_channels(V.cpu().numpy(), labels.astype(np.float32), s.cpu().numpy(),
                                     t.cpu().numpy(), L.cpu().numpy(), n_channels=C)
        assert rel_err(ch, ref) < FWD_TOL
        assert rel_err(ch, per_ray) < 3e-5
        plain, _ = ops.siddon_forward_bricks(V, s, t, L, (H, W))
        assert rel_err(ch.sum(1), plain.cpu().numpy()) < 3e-5
        assert np.all(ch[per_ray == 0] == 0)
    # fewer channels than labels: the rest is dropped
    ch8 = ops.siddon_forward_channels_bricks(V, lab, 8, s, t, L, (H, W)).cpu().numpy()
    assert rel_err(ch8, ref[:, :8]) < FWD_TOL
    # the module route (DRR with a mask) takes this kernel and stays differentiable
    sub = synthetic_subject(D, kind=kind, seed=5, n_labels=7)
    drr2 = DRR(sub, sdd=700.0, height=H, width=W, delx=3.0).to(gpu)
    r = rot[:2].clone().requires_grad_()
    x = xyz[:2].clone().requires_grad_()
    chm = drr2(r, x, parameterization="euler_angles", convention="ZXY", mask_to_channels=True)
    one = drr2(r, x, parameterization="euler_angles", convention="ZXY")
    assert rel_err(chm.sum(1, keepdim=True).detach().cpu().numpy(), one.detach().cpu().numpy()) < 3e-5
    w = torch.rand(one.shape, device=gpu, generator=torch.Generator(gpu).manual_seed(2))
    ga = torch.autograd.grad((chm.sum(1, keepdim=True) * w).sum(), [r, x])
    gb = torch.autograd.grad((one * w).sum(), [r, x])
    for a, b in zip(ga, gb):
        # (noise volumes: fp32 tie attribution makes any two walks differ at the 1e-2..1e-1 level)
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < (1e-1 if kind == "noise" else 2e-3)


@pytest.mark.parametrize("name,npts,rng,shift", [
    ("trilinear_global_range", 41, None, 0.5), ("trilinear_explicit_range", 64, (0.31, 0.77), 0.5),
    ("trilinear_oblique", 50, None, 0.5), ("trilinear_shift0", 40, None, 0.0),
])
def test_trilinear_golden(gpu, name, npts, rng, shift):
    g = golden(name)
    vol, src, tgt, img = (t.requires_grad_() for t in dev_inputs(g, gpu))
    kw = {}
    leaves = [src, tgt, img, vol]
    if rng is not None:
        kw["alphamin"] = torch.tensor(rng[0], device=gpu, requires_grad=True)
        kw["alphamax"] = torch.tensor(rng[1], device=gpu, requires_grad=True)
        leaves += [kw["alphamin"], kw["alphamax"]]
    out = Trilinear(voxel_shift=shift)(vol, src, tgt, img, n_points=npts, **kw)
    assert rel_err(out.detach().cpu().numpy(), g["out_f32"]) < FWD_TOL
    go = torch.from_numpy(g["grad_out_f32"]).to(gpu)
    grads = torch.autograd.grad(out, leaves, go)
    # with the batch-global range the reference's gradients include the path through
    # alphamin/alphamax -> the arg-min/arg-max ray; ours do too (torch min/max routes it)
    assert rel_err(grads[0].cpu().numpy(), g["g_source_f64"]) < GRAD_TOL
    assert rel_err(grads[1].cpu().numpy(), g["g_target_f64"]) < GRAD_TOL
    assert rel_err(grads[2].cpu().numpy(), g["g_img_f64"]) < GRAD_TOL
    assert rel_err(grads[3].cpu().numpy(), g["g_volume_f64"]) < GRAD_TOL
    if rng is not None:
        assert abs(grads[4].item() / g["g_alphamin_f64"] - 1) < GRAD_TOL
        assert abs(grads[5].item() / g["g_alphamax_f64"] - 1) < GRAD_TOL


def test_trilinear_max_gradients_golden(gpu):
    """Trilinear(reducefn="max") backward (ddrr_trilinear_backward_max) against the reference."""
    g = golden("trilinear_max")
    vol, src, tgt, img = (t.requires_grad_() for t in dev_inputs(g, gpu))
    out = Trilinear(reducefn="max")(vol, src, tgt, img, n_points=37)
    assert rel_err(out.detach().cpu().numpy(), g["out_f32"]) < FWD_TOL
    go = torch.from_numpy(g["grad_out_f32"]).to(gpu)
    grads = torch.autograd.grad(out, [src, tgt, img, vol], go)
    for k, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.cpu().numpy(), g[k + "_f64"]) < GRAD_TOL, k


def test_trilinear_nearest_max_golden(gpu):
    g = golden("trilinear_nearest_max")
    vol, src, tgt, img = dev_inputs(g, gpu)
    out = Trilinear(mode="nearest", reducefn="max")(vol, src, tgt, img, n_points=33)
    assert rel_err(out.cpu().numpy(), g["out_f32"]) < FWD_TOL


# ------------------------------------------------------------- DRR module

def _subject_a(g):
    vol = torch.from_numpy(g["volume"])
    mask = Image(torch.from_numpy(g["mask"]).unsqueeze(0), g["affine"])
    return Subject(Image(vol.unsqueeze(0), g["affine"]), Image(vol.unsqueeze(0), g["affine"]),
                   torch.from_numpy(g["reorient"]), mask)


def _geo(g, prefix="geo_"):
    geo = {k[len(prefix):]: g[k].item() for k in g.files if k.startswith(prefix)}
    for k in ("height", "width"):
        if k in geo:
            geo[k] = int(geo[k])
    return geo


@pytest.mark.parametrize("renderer,kw", [("siddon", {}), ("trilinear", {"n_points": 60})])
def test_drr_module_golden(gpu, renderer, kw):
    g = golden("drr_module")
    drr = DRR(_subject_a(g), renderer=renderer, **_geo(g)).to(gpu)
    rot = torch.from_numpy(g["rot"]).to(gpu).requires_grad_()
    xyz = torch.from_numpy(g["xyz"]).to(gpu).requires_grad_()
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", **kw)
    assert img.shape == g[f"{renderer}_img_f32"].shape
    assert rel_err(img.detach().cpu().numpy(), g[f"{renderer}_img_f32"]) < FWD_TOL
    img.backward(torch.from_numpy(g[f"{renderer}_grad_out_f32"]).to(gpu))
    assert rel_err(rot.grad.cpu().numpy(), g[f"{renderer}_g_rot_f64"]) < GRAD_TOL
    assert rel_err(xyz.grad.cpu().numpy(), g[f"{renderer}_g_xyz_f64"]) < GRAD_TOL
    if renderer == "siddon":
        with torch.no_grad():
            ch = drr(rot, xyz, parameterization="euler_angles", convention="ZXY",
                     mask_to_channels=True)
        assert rel_err(ch.cpu().numpy(), g["siddon_channels_f32"]) < FWD_TOL


# ------------------------------------------- oracle at config-1/2-like sizes

def scene(D, det, delx, B, gpu, seed=1, renderer="siddon", kind="noise", **kw):
    """SURVEY.md section 8(d) common scene: sdd 1020, AP, base pose (0,0,0)/(0,850,0)
    perturbed by U(+-pi/4)^3 and U(+-30)^3 for every pose but the first."""
    subject = synthetic_subject(D, kind=kind, seed=0)
    drr = DRR(subject, sdd=1020.0, height=det, delx=delx, renderer=renderer, **kw).to(gpu)
    g = torch.Generator().manual_seed(seed)
    rot = (torch.rand(B, 3, generator=g) - 0.5) * (np.pi / 2)
    xyz = torch.tensor([0.0, 850.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 60
    rot[0] = 0.0
    xyz[0] = torch.tensor([0.0, 850.0, 0.0])
    return drr, rot.to(gpu), xyz.to(gpu)


def voxel_rays(drr, rot, xyz):
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1)
        return (drr.affine_inverse(source).contiguous(), drr.affine_inverse(target).contiguous(),
                L.contiguous())


def _grad_close(mine, ref32, ref64, what):
    """fp32 ray gradients are sums of +-(V_before - V_after) over hundreds of crossings;
    when two crossings are closer than fp32 resolves, their order -- hence the voxel in
    between -- is arbitrary, for the reference's fp32 autograd (torch.sort) just as for
    the kernel.  So the yardstick is the fp64 oracle, and the allowance is what the
    reference's own fp32 arithmetic (the fp32 oracle) loses against it, times two."""
    err_mine = rel_err(mine, ref64)
    err_ref = rel_err(ref32, ref64)
    assert err_mine < 2 * err_ref + GRAD_TOL, (what, err_mine, err_ref)


@pytest.mark.parametrize("D,det,delx,B,kind", [(64, 64, 1.5, 1, "noise"),
                                               (128, 96, 2.0, 3, "noise"),
                                               (128, 96, 2.0, 3, "phantom")])
def test_siddon_vs_oracle_medium(gpu, D, det, delx, B, kind):
    """Config-1-like (64^3 -> 64^2) and 128^3 batches with oblique poses: forward vs the
    fp32 oracle, gradients vs the fp64 oracle."""
    drr, rot, xyz = scene(D, det, delx, B, gpu, kind=kind)
    # The exact base pose is a measure-zero case for GRADIENTS: the source sits on the
    # volume's symmetry axis, so diagonal pixels cross x- and z-planes at identical alphas
    # and the reference's gradient depends on torch.sort's tie order.
    # Poses within a few hundredths of a radian of it still have many crossings closer
    # than fp32 resolves, where the kernel (alpha = fma(k, 1/d, c)) flips somewhat more
    # often than the reference (alpha = (plane - s)/d); see DESIGN.md "gradient noise".
    # Use a clearly generic pose; the forward image at the exact base pose is checked in
    # test_siddon_base_pose_forward.
    rot[0] += torch.tensor([0.13, -0.21, 0.17], device=gpu)
    xyz[0] += torch.tensor([3.7, 0.0, -2.3], device=gpu)
    rot.requires_grad_()
    xyz.requires_grad_()
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    s, t, L = voxel_rays(drr, rot.detach(), xyz.detach())
    vol = drr.density.cpu().numpy()
    go = torch.randn(img.shape, generator=torch.Generator().manual_seed(5)).to(gpu)
    args32 = (vol, s.cpu().numpy(), t.cpu().numpy(), L.cpu().numpy())
    go_np = go.cpu().numpy().reshape(B, -1)
    ref = oracle.siddon(*args32, grad_out=go_np, want_volume_grad=True)
    ref64 = oracle.siddon(*(a.astype(np.float64) for a in args32),
                          grad_out=go_np.astype(np.float64), want_volume_grad=True)
    assert rel_err(img.detach().cpu().numpy().reshape(ref["out"].shape), ref["out"]) < FWD_TOL
    # kernel-level gradients w.r.t. the voxel-space rays and the volume
    s.requires_grad_()
    t.requires_grad_()
    vol_t = drr.density.clone().requires_grad_()
    out = Siddon()(vol_t, s, t, L.unsqueeze(1))
    gs, gt, gv = torch.autograd.grad(out, (s, t, vol_t), go.reshape(out.shape))
    _grad_close(gs.cpu().numpy(), ref["g_source"], ref64["g_source"], "g_source")
    _grad_close(gt.cpu().numpy(), ref["g_target"], ref64["g_target"], "g_target")
    assert rel_err(gv.cpu().numpy(), ref64["g_volume"]) < GRAD_TOL
    # per ray: no more rays disagree with fp64 than for the reference's fp32 arithmetic
    scale = np.abs(ref64["g_target"]).max()
    off = lambda g: float((np.abs(g - ref64["g_target"]).max(-1) > 1e-3 * scale).mean())  # noqa
    assert off(gt.cpu().numpy()) <= 2 * off(ref["g_target"]) + 0.005
    img.backward(go)
    assert torch.isfinite(rot.grad).all() and xyz.grad.abs().max() > 0


def test_siddon_base_pose_forward(gpu):
    """BASELINE config 1: 64^3 volume, 64x64 detector, Siddon, one pose (the base pose)."""
    drr, rot, xyz = scene(64, 64, 1.5, 1, gpu)
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    s, t, L = voxel_rays(drr, rot, xyz)
    ref = oracle.siddon(drr.density.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy(),
                        L.cpu().numpy())["out"]
    assert img.shape == (1, 1, 64, 64)
    assert rel_err(img.cpu().numpy().reshape(ref.shape), ref) < FWD_TOL


def test_trilinear_vs_oracle_medium(gpu):
    drr, rot, xyz = scene(96, 64, 2.5, 2, gpu, renderer="trilinear")
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", n_points=200)
    s, t, L = voxel_rays(drr, rot, xyz)
    ref = oracle.trilinear(drr.density.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy(),
                           L.cpu().numpy(), n_points=200)
    assert rel_err(img.cpu().numpy().reshape(ref["out"].shape), ref["out"]) < FWD_TOL


def test_voxel_counts_match_oracle(gpu):
    drr, rot, xyz = scene(64, 48, 2.0, 2, gpu)
    s, t, L = voxel_rays(drr, rot, xyz)
    _, _, nvox = ops.siddon_forward(drr.density, s, t, L, count_voxels=True, det=(48, 48))
    ref = oracle.siddon(drr.density.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy(),
                        L.cpu().numpy(), count_voxels=True)["n_inside"]
    assert abs(int(nvox.sum()) - int(ref.sum())) <= 0.002 * ref.sum() + 2


# --------------------------------------- full BASELINE sizes: properties only

@pytest.fixture(scope="module")
def big(gpu):
    """512^3 volume, 256^2 detector, delx 2.4 (SURVEY.md section 8d config 4/5 geometry)."""
    drr, rot, xyz = scene(512, 256, 2.4, 4, gpu)
    s, t, L = voxel_rays(drr, rot, xyz)
    return drr, s, t, L


def chord_lengths(s, t, D, eps=1e-8):
    """Analytic alpha-extent of line n box [-0.5, D-0.5]^3 (fp64)."""
    s, t = s.double(), t.double()
    d = t - s + eps
    a0, a1 = (-0.5 - s) / d, (D - 0.5 - s) / d
    lo = torch.minimum(a0, a1).amax(-1)
    hi = torch.maximum(a0, a1).amin(-1)
    return (hi - lo).clamp_min(0)


def test_full_size_constant_volume_gives_chord_length(gpu, big):
    """V == 1  =>  DRR pixel = ||t - s|| * (alpha_exit - alpha_entry), exactly."""
    drr, s, t, L = big
    ones = torch.ones_like(drr.density)
    out, _, nvox = ops.siddon_forward(ones, s, t, L, count_voxels=True, det=(256, 256))
    ref = chord_lengths(s, t, 512) * L.double()
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    hit = ref > 0
    assert (nvox[hit] >= 1).all() and int(nvox.max()) <= 3 * 512
    # (the detector plane of this scene lies INSIDE the 512^3 volume: the marcher's
    # default [0, 1] range would stop at the target, Siddon integrates the whole line)
    tri = ops.trilinear_forward(ones, s, t, L, torch.tensor(0.0, device=gpu),
                                torch.tensor(1.5, device=gpu), n_points=3072, det=(256, 256))
    # Trilinear of a box ramps 0 -> 1 over one voxel at each face (zeros padding), which is
    # chord-preserving only for rays that cross faces transversally: check the central
    # 64x64 pixels of the base pose (pose 0), rectangular rule => a couple of steps.
    c = torch.arange(96, 160, device=gpu)
    centre = (c[:, None] * 256 + c[None, :]).reshape(-1)
    step_len = 1.5 * L[0, centre] / 3071
    assert ((tri[0, centre] - ref[0, centre].float()).abs() <= 2.5 * step_len + 1e-3).all()


def test_full_size_linearity_and_invariances(gpu, big):
    drr, s, t, L = big
    V1 = drr.density
    V2 = torch.rand(V1.shape, generator=torch.Generator().manual_seed(9)).to(gpu)
    r = lambda V, **k: ops.siddon_forward(V, s, t, L, det=(256, 256), **k)[0]  # noqa: E731
    a = r(V1)
    lin = r(0.75 * V1 + 0.5 * V2)
    assert rel_err(lin.cpu().numpy(), (0.75 * a + 0.5 * r(V2)).cpu().numpy()) < 1e-5
    # tiling / XCD mapping only permute lanes and workgroups: bit-identical images
    for tile in [(64, 1), (8, 8), (1, 64)]:
        assert torch.equal(r(V1, tile=tile), a)
    assert torch.equal(ops.siddon_forward(V1, s, t, L)[0], a)  # plain ray list
    # a pose rendered alone equals the same pose inside the batch (no cross-talk)
    alone = ops.siddon_forward(V1, s[2:3], t[2:3], L[2:3], det=(256, 256))[0]
    assert torch.equal(alone[0], a[2])
    # the aux-emitting variant returns the same image
    with_aux = ops.siddon_forward(V1, s, t, L, want_aux=True, det=(256, 256))[0]
    assert rel_err(with_aux.cpu().numpy(), a.cpu().numpy()) < 1e-6


def test_full_size_volume_gradient_is_the_adjoint(gpu, big):
    """<render(V), g> == <V, backward_volume(g)> for the linear map V -> DRR."""
    drr, s, t, L = big
    V = drr.density
    g = torch.randn(4, 256 * 256, generator=torch.Generator().manual_seed(3)).to(gpu)
    out = ops.siddon_forward(V, s, t, L, det=(256, 256))[0]
    gv = ops.siddon_backward_volume(V, s, t, L, g, det=(256, 256))
    lhs = (out.double() * g.double()).sum().item()
    rhs = (V.double() * gv.double()).sum().item()
    assert abs(lhs - rhs) / abs(lhs) < 1e-4
    # trilinear: same identity through its backward
    a0, a1 = torch.tensor(0.3, device=gpu), torch.tensor(0.7, device=gpu)
    tri = ops.trilinear_forward(V, s[:1], t[:1], L[:1], a0, a1, n_points=300, det=(256, 256))
    r = ops.trilinear_backward(V, s[:1], t[:1], L[:1], g[:1], a0, a1, n_points=300,
                               want_volume=True, det=(256, 256))
    lhs = (tri.double() * g[:1].double()).sum().item()
    rhs = (V.double() * r["g_volume"].double()).sum().item()
    assert abs(lhs - rhs) / abs(lhs) < 1e-4


def test_pose_gradient_matches_fp64_chain(gpu):
    """End-to-end autograd (convert -> Detector -> HIP Siddon -> image) at 256^3 on the
    smooth phantom against the same chain evaluated in fp64 with the fp64 oracle's ray
    gradients.  (Finite differences are not a usable yardstick here: with
    nearest-neighbour voxels the image is only piecewise smooth in the rotation, so secant
    slopes do not converge to the derivative at fp32-resolvable step sizes.)"""
    import copy

    drr, rot, xyz = scene(256, 128, 3.0, 2, gpu, kind="phantom")
    rot = rot + 0.05
    ii, jj = torch.meshgrid(torch.linspace(-1, 1, 128), torch.linspace(-1, 1, 128),
                            indexing="ij")
    W = (1 + 0.5 * ii - 0.3 * jj)[None, None].to(gpu)
    r = rot.clone().requires_grad_()
    x = xyz.clone().requires_grad_()
    (drr(r, x, parameterization="euler_angles", convention="ZXY") * W).sum().backward()

    drr64 = copy.deepcopy(drr).cpu().double()
    g = W.cpu().double().reshape(1, -1).expand(2, -1).numpy()

    def chain(ray_grads):
        r64 = rot.cpu().double().requires_grad_()
        x64 = xyz.cpu().double().requires_grad_()
        pose = convert(r64, x64, parameterization="euler_angles", convention="ZXY")
        source, target = drr64.detector(pose, None)
        L = (target - source).norm(dim=-1)
        s, t = drr64.affine_inverse(source), drr64.affine_inverse(target)
        o = ray_grads(s.detach().numpy(), t.detach().numpy(), L.detach().numpy())
        as64 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))  # noqa: E731
        surrogate = ((as64(o["g_source"]) * s).sum() + (as64(o["g_target"]) * t).sum()
                     + (as64(o["g_img"]).reshape(L.shape) * L).sum())
        surrogate.backward()
        return r64.grad.numpy(), x64.grad.numpy()

    vol64 = drr64.density.numpy()
    truth_r, truth_x = chain(lambda s, t, L: oracle.siddon(vol64, s, t, L, grad_out=g))
    # the reference's fp32 arithmetic on the same rays (fp32 oracle), chained exactly:
    # the rotation gradient is a small difference of large source/target terms and loses
    # ~2 digits to the fp32 rounding of the ray endpoints alone
    f32 = lambda a: a.astype(np.float32)  # noqa: E731
    ref_r, ref_x = chain(lambda s, t, L: oracle.siddon(f32(vol64), f32(s), f32(t), f32(L),
                                                       grad_out=f32(g)))
    assert rel_err(r.grad.cpu().numpy(), truth_r) < 2 * rel_err(ref_r, truth_r) + 2e-3
    assert rel_err(x.grad.cpu().numpy(), truth_x) < 2 * rel_err(ref_x, truth_x) + 2e-3


def test_generic_record_identities_full_size(gpu, big):
    """What does not depend on how ties are attributed: sum_a S0_a = 0 and sum_a S1_a = I per
    ray, for the per-ray walk's record at full size."""
    drr, s, t, L = big
    ref, aux_ref, _ = ops.siddon_forward(drr.density, s, t, L, want_aux=True, det=(256, 256))
    scale = aux_ref.abs().max()
    assert (aux_ref[..., 1:4].sum(-1).abs().max() <= 2e-5 * scale)
    assert ((aux_ref[..., 4:7].sum(-1) - aux_ref[..., 0]).abs().max() <= 2e-5 * scale)


def test_brick_kernel_equals_generic_walk(gpu, big):
    """Volume-stationary brick kernel (csrc/brick_core.h) at full size: same image as the
    per-crossing walk up to the order in which the per-brick pieces are added."""
    drr, s, t, L = big
    V = drr.density
    ref, aux_ref, _ = ops.siddon_forward(V, s, t, L, det=(256, 256), want_aux=True)
    out, none = ops.siddon_forward_bricks(V, s, t, L, (256, 256))
    assert none is None
    # (both walks evaluate every crossing as the reference's quotient (k - shift - s) / d, the
    # brick walk via the first plane ahead of each brick entry; the sums are grouped per brick.
    # Accuracy proper is checked against the fp64 oracle in test_gpu_baseline_sizes.py.)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 2e-5
    again, aux = ops.siddon_forward_bricks(V, s, t, L, (256, 256), want_aux=True)
    # (the forward-only launch accumulates chord-relative alphas, the launch with the record
    # keeps the plane counters: two fp32 evaluations, brick_step.h step_walk_fwd)
    assert rel_err(again.cpu().numpy(), out.cpu().numpy()) < 1e-5
    # blocked backward record (csrc/record_layout.h): same I, and the same ray gradients as the
    # generic record except on the ~1 % of rays holding a crossing pair that ties in fp32
    assert aux.shape == (4 * 256 * 256 // 16, 80)
    planes = ops.record_planes(aux, 4, 256 * 256)
    assert torch.allclose(planes[0], aux_ref[..., 0], rtol=1e-4, atol=1e-5)
    # S0x, S0z, S1x, S1z of the interleaved record {I, S0xyz, S1xyz}: the planes are the ones
    # the layout says they are (ties aside: compare where the rays agree)
    for k, col in ((1, 1), (2, 3), (3, 4), (4, 6)):
        scale_k = aux_ref[..., col].abs().max()
        same = (planes[k] - aux_ref[..., col]).abs() <= 1e-3 * scale_k
        assert same[1:].float().mean().item() > 0.97, k  # (pose 0: on a symmetry plane, exact ties)
    go = torch.rand(out.shape, device=gpu, generator=torch.Generator(gpu).manual_seed(0))
    gsb, gtb, gib = ops.siddon_backward_rays(aux, go, s, t, L)
    gsg, gtg, gig = ops.siddon_backward_rays(aux_ref, go, s, t, L)
    assert rel_err(gib.cpu().numpy(), gig.cpu().numpy()) < 1e-4
    close = (gtb - gtg).abs().amax(-1) <= 1e-3 * gtg.abs().max()
    assert close[1:].float().mean().item() > 0.97
    # (per-pose sums: the two walks round alphas beyond a brick's first plane differently by up
    # to an ulp, so they flip different near-ties on this NOISE volume; against the fp64 oracle
    # both are held to the reference's own fp32 error in test_gpu_baseline_sizes.py)
    for b in range(1, 4):
        assert rel_err(gtb[b].double().sum(0).cpu().numpy(),
                       gtg[b].double().sum(0).cpu().numpy()) < 5e-2
        assert rel_err(gsb[b].double().sum(0).cpu().numpy(),
                       gsg[b].double().sum(0).cpu().numpy()) < 5e-2


def test_brick_kernel_small_and_ragged_volumes(gpu):
    for dims, (H, W) in (((40, 70, 33), (24, 31)), ((32, 32, 32), (16, 16)), ((5, 9, 64), (8, 8))):
        subject = make_subject(torch.rand(*dims, generator=torch.Generator().manual_seed(1)),
                               spacing=(1.0, 1.5, 2.0))
        drr = DRR(subject, sdd=300.0, height=H, width=W, delx=2.0).to(gpu)
        # (not the exact base pose: with an odd detector its centre column runs exactly
        # inside a voxel plane, where fp32 cannot tell the two neighbouring voxels apart)
        rot = torch.tensor([[0.01, 0.02, -0.01], [0.6, -0.4, 0.9], [1.5, 0.2, 0.1],
                            [0.2, 0.1, 0.0]], device=gpu)
        xyz = torch.tensor([[0.3, 200.0, 0.2], [4.0, 180.0, -6.0], [1.0, 210.0, 2.0],
                            [2.0, 5.0, -3.0]], device=gpu)  # last: source inside the volume
        s, t, L = voxel_rays(drr, rot, xyz)
        ref = oracle.siddon(drr.density.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy(),
                            L.cpu().numpy())["out"].reshape(4, -1)
        out, _ = ops.siddon_forward_bricks(drr.density, s, t, L, (H, W))
        assert rel_err(out.cpu().numpy(), ref) < FWD_TOL, dims


@pytest.mark.parametrize("q16", ["q16", "q16p"])
def test_q16_bricks_small_ragged_sparse_and_many_poses(gpu, q16):
    """The 16-bit block-quantised bricks (32 x 32 x 64, ddrr_siddon_forward_bricks with
    DDRR_BRICKS_Q16, and DDRR_BRICKS_Q16_PACKED: the same bricks staged from the packed copy in
    the workspace, first call and cached call) where the module would not choose them -- small volumes -- for the edge cases
    of the brick grid: partial bricks on every axis, a volume smaller than one brick, empty (all
    zero) bricks next to full ones, a z extent that is not a multiple of 4 (quads staged from
    dword-aligned addresses), more poses than a pose-table chunk; against the oracle, forward and the
    record's ray gradients; and a NaN voxel sends its brick to the fp32 path: only rays through
    the voxel itself are NaN, as with the volume's own values."""
    cases = (((40, 72, 36), (24, 31), 4), ((32, 32, 64), (16, 16), 4), ((5, 9, 64), (8, 8), 4),
             ((40, 70, 33), (24, 31), 4), ((70, 40, 132), (20, 28), 75))
    for dims, (H, W), B in cases:
        g = torch.Generator().manual_seed(1)
        vol = torch.rand(*dims, generator=g)
        if dims[0] >= 40:
            vol[: dims[0] // 2, : dims[1] // 2] = 0.0  # air: bricks of zeros are skipped
        drr = DRR(make_subject(vol, spacing=(1.0, 1.5, 2.0)), sdd=300.0, height=H, width=W,
                  delx=2.0).to(gpu)
        rot = ((torch.rand(B, 3, generator=g) - 0.5) * 2.0).to(gpu)
        xyz = (torch.tensor([0.0, 200.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 30).to(gpu)
        s, t, L = voxel_rays(drr, rot, xyz)
        V = drr.density
        go = torch.rand(B, H * W, generator=g).to(gpu)
        o = oracle.siddon(V.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy(), L.cpu().numpy(),
                          grad_out=go.cpu().numpy())
        out, aux = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=True, storage=q16)
        plain, _ = ops.siddon_forward_bricks(V, s, t, L, (H, W), storage=q16)
        # (cached: the second call reused the workspace the first one built)
        assert ops.brick_workspace(V, q16)[1] == int(ops.brick_storage_applies(V))
        exact, aux_f = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=True, storage="f32")
        for img in (out, plain):
            assert rel_err(img.cpu().numpy(), o["out"].reshape(B, -1)) < FWD_TOL, dims
            assert rel_err(img.cpu().numpy(), exact.cpu().numpy()) < 2e-5, dims
        gq = ops.siddon_backward_rays(aux, go, s, t, L)
        gf = ops.siddon_backward_rays(aux_f, go, s, t, L)
        assert rel_err(gq[2].cpu().numpy(), gf[2].cpu().numpy()) < 2e-5, dims      # d / d img
        for a, b in zip(gq[:2], gf[:2]):                                             # per-pose sums
            assert rel_err(a.double().sum(1).cpu().numpy(), b.double().sum(1).cpu().numpy()) < 5e-3, dims
    # a NaN voxel: the rays through it are NaN, the others are untouched
    vol = torch.rand(64, 64, 128, generator=torch.Generator().manual_seed(2))
    drr = DRR(make_subject(vol, spacing=(1.0, 1.0, 1.0)), sdd=400.0, height=32, delx=3.0).to(gpu)
    rot = torch.tensor([[0.1, 0.2, -0.1]], device=gpu)
    xyz = torch.tensor([[1.0, 250.0, 2.0]], device=gpu)
    s, t, L = voxel_rays(drr, rot, xyz)
    clean, _ = ops.siddon_forward_bricks(drr.density, s, t, L, (32, 32), storage=q16)
    bad = drr.density.clone()
    # (a voxel the central pixel's ray crosses: the middle of its samples inside the volume)
    n_c = 16 * 32 + 16
    al = torch.linspace(0, 1, 4001, device=gpu)[:, None]
    pts = s[0, 0] + al * (t[0, n_c] - s[0, 0])
    ins = ((pts > -0.5) & (pts < torch.tensor(bad.shape, device=gpu) - 0.5)).all(-1)
    vx = (pts[ins][int(ins.sum()) // 2] + 0.5).floor().long().tolist()
    bad[vx[0], vx[1], vx[2]] = float("nan")
    dirty, _ = ops.siddon_forward_bricks(bad, s, t, L, (32, 32), storage=q16)
    same, _ = ops.siddon_forward_bricks(bad, s, t, L, (32, 32), storage="f32")
    # an in-place edit of a rendered volume is seen (the version counter invalidates the workspace)
    bad[vx[0], vx[1], vx[2]] = 0.5
    fixed, _ = ops.siddon_forward_bricks(bad, s, t, L, (32, 32), storage=q16)
    assert not torch.isnan(fixed).any()
    nan = torch.isnan(dirty)
    assert nan[0, n_c] and not nan.all() and torch.equal(nan, torch.isnan(same))
    assert torch.allclose(dirty[~nan], clean[~nan], rtol=1e-5, atol=1e-5)  # (atomics: not bit-stable)


@pytest.mark.parametrize("H,W", [(70, 45), (64, 64), (33, 130)])
def test_odd_detectors_vs_oracle(gpu, H, W):
    subject = synthetic_subject(48, kind="noise", seed=0)
    drr = DRR(subject, sdd=400.0, height=H, width=W, delx=1.1).to(gpu)
    # (not the exact base pose: with an odd detector its centre row glides inside a voxel
    # plane, where the reference's fp32 position rounding -- not geometry -- picks the voxel)
    rot = torch.tensor([[0.01, 0.02, -0.01], [0.5, -0.3, 0.8], [1.5, 0.1, 0.2], [0.1, 1.4, 0.0]],
                       device=gpu)
    xyz = torch.tensor([[0.3, 300.0, 0.2], [5.0, 280.0, -7.0], [0.0, 310.0, 3.0],
                        [2.0, 300.0, 1.0]], device=gpu)
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    s, t, L = voxel_rays(drr, rot, xyz)
    ref = oracle.siddon(drr.density.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy(),
                        L.cpu().numpy())["out"]
    assert rel_err(img.cpu().numpy().reshape(ref.shape), ref) < FWD_TOL
    drr.renderer.grid_path = "generic"
    img2 = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    assert rel_err(img2.cpu().numpy(), img.cpu().numpy()) < 2e-5


def test_deterministic_forward(gpu, big):
    drr, s, t, L = big
    a = ops.siddon_forward(drr.density, s, t, L, det=(256, 256))[0]
    b = ops.siddon_forward(drr.density, s, t, L, det=(256, 256))[0]
    assert torch.equal(a, b)


def test_empty_and_ragged_inputs(gpu):
    vol = torch.rand(5, 6, 7, device=gpu)
    src = torch.zeros(0, 1, 3, device=gpu)
    tgt = torch.zeros(0, 9, 3, device=gpu)
    assert Siddon()(vol, src, tgt, torch.zeros(0, 1, 9, device=gpu)).shape == (0, 1, 9)
    # N not a multiple of 64, single ray, 1-voxel-thick volume
    thin = torch.rand(1, 9, 1, device=gpu)
    s = torch.tensor([[[0.0, -20.0, 0.0]]], device=gpu)
    t = torch.tensor([[[0.0, 30.0, 0.0]]], device=gpu)
    out = Siddon()(thin, s, t, torch.full((1, 1, 1), 50.0, device=gpu))
    assert abs(out.item() - thin.sum().item()) < 1e-4  # 9 unit-length voxels * 50/50
    # a callable reducefn gets the materialised per-segment tensor (ddrr_siddon_segments)
    one = torch.ones(1, 1, 1, device=gpu)
    a = Siddon(reducefn=lambda x: x.sum(-1))(vol, s, t, one)
    assert torch.allclose(a, Siddon()(vol, s, t, one), rtol=1e-5, atol=1e-6)
    # ... with any lookup (the general path, csrc/general_core.h)
    b = Siddon(mode="bilinear", reducefn=lambda x: x.sum(-1))(vol, s, t, one)
    assert torch.allclose(b, Siddon(mode="bilinear")(vol, s, t, one), rtol=1e-5, atol=1e-6)
    # float64 inputs render in double (csrc/f64_rays.hip), like the reference module .to(float64)
    d = Siddon()(vol.double(), s.double(), t.double(), torch.ones(1, 1, 1, device=gpu).double())
    assert d.dtype == torch.float64
    assert abs(d.item() - Siddon()(vol, s, t, one).item()) < 1e-5
    with pytest.raises(NotImplementedError):  # mixed dtypes are not guessed at
        ops.siddon_forward(vol.double(), s, t, None)


@pytest.mark.parametrize("name", ["siddon_sum", "siddon_sum_oblique", "siddon_per_ray_source"])
def test_siddon_float64_golden(gpu, name):
    """fp64 kernels on the GPU against the reference's own fp64 outputs and autograd gradients."""
    g = golden(name)
    vol, src, tgt = (torch.from_numpy(g[k].astype(np.float64)).to(gpu).requires_grad_()
                     for k in ("volume", "source", "target"))
    B, N, _ = tgt.shape
    img = torch.from_numpy(g["img_f64"].reshape(B, 1, N)).to(gpu).requires_grad_()
    out = Siddon()(vol, src, tgt, img)
    assert out.dtype == torch.float64
    assert rel_err(out.detach().cpu().numpy(), g["out_f64"]) < 1e-12
    gs, gt, gi, gv = torch.autograd.grad(out, (src, tgt, img, vol),
                                         torch.from_numpy(g["grad_out_f64"]).to(gpu))
    for mine, key in ((gs, "g_source_f64"), (gt, "g_target_f64"), (gi, "g_img_f64"), (gv, "g_volume_f64")):
        assert rel_err(mine.cpu().numpy(), g[key]) < 1e-10, key


def test_trilinear_float64_golden_and_drr_module(gpu):
    g = golden("trilinear_global_range")
    vol, src, tgt = (torch.from_numpy(g[k].astype(np.float64)).to(gpu).requires_grad_()
                     for k in ("volume", "source", "target"))
    B, N, _ = tgt.shape
    img = torch.from_numpy(g["img_f64"].reshape(B, 1, N)).to(gpu).requires_grad_()
    out = Trilinear()(vol, src, tgt, img, n_points=41)
    assert rel_err(out.detach().cpu().numpy(), g["out_f64"]) < 1e-6  # (fp32 linspace table, see f64_core.h)
    gs, gt, gi, gv = torch.autograd.grad(out, (src, tgt, img, vol),
                                         torch.from_numpy(g["grad_out_f64"]).to(gpu))
    for mine, key in ((gs, "g_source_f64"), (gt, "g_target_f64"), (gi, "g_img_f64"), (gv, "g_volume_f64")):
        assert rel_err(mine.cpu().numpy(), g[key]) < 1e-5, key
    # the whole module in double, like the reference's `DRR(...).to(torch.float64)`
    gm = golden("drr_module")
    drr = DRR(_subject_a(gm), renderer="siddon", **_geo(gm)).to(gpu).to(torch.float64)
    rot = torch.from_numpy(gm["rot"]).double().to(gpu).requires_grad_()
    xyz = torch.from_numpy(gm["xyz"]).double().to(gpu).requires_grad_()
    im = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    assert im.dtype == torch.float64
    assert rel_err(im.detach().cpu().numpy(), gm["siddon_img_f32"]) < FWD_TOL
    im.backward(torch.from_numpy(gm["siddon_grad_out_f32"]).double().to(gpu))
    assert rel_err(rot.grad.cpu().numpy(), gm["siddon_g_rot_f64"]) < 1e-6
    assert rel_err(xyz.grad.cpu().numpy(), gm["siddon_g_xyz_f64"]) < 1e-6


@pytest.mark.parametrize("stop", [False, True])
def test_fused_ray_generation_equals_general_path(gpu, stop):
    """DRR.forward's fused entry (ddrr_raygen_forward + brick kernel +
    ddrr_siddon_backward_pose) against Detector + render in PyTorch on the same poses:
    bit-identical rays, same image, same pose and volume gradients."""
    drr = DRR(synthetic_subject(96, kind="noise", seed=0), sdd=600.0, height=80, width=72,
              delx=1.6, stop_gradients_through_grid_sample=stop).to(gpu)
    drr.density.requires_grad_(not stop)
    g = torch.Generator().manual_seed(8)
    rot0 = ((torch.rand(6, 3, generator=g) - 0.5) * 1.2).to(gpu)
    xyz0 = (torch.tensor([0.0, 420.0, 0.0]) + (torch.rand(6, 3, generator=g) - 0.5) * 30).to(gpu)
    go = torch.rand(6, 1, 80, 72, generator=g).to(gpu)
    res = {}
    for fused in (True, False):
        drr.fuse_ray_generation = fused
        rot, xyz = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
        drr.density.grad = None
        # (a RigidTransform goes in, so that both paths see bit-identical rays: with the
        # fused Euler pose kernel the rotation differs from torch's matmul chain in the last
        # bit, enough to flip fp32 ties in single rays' gradient records)
        img = drr(convert(rot, xyz, parameterization="euler_angles", convention="ZXY"))
        (img * go).sum().backward()
        res[fused] = (img.detach().cpu().numpy(), rot.grad.cpu().numpy(), xyz.grad.cpu().numpy(),
                      None if stop else drr.density.grad.cpu().numpy())
    a, b = res[True], res[False]
    assert rel_err(a[0], b[0]) < 1e-5
    assert rel_err(a[1], b[1]) < GRAD_TOL and rel_err(a[2], b[2]) < GRAD_TOL
    if not stop:
        assert rel_err(a[3], b[3]) < 1e-5
    # the rays themselves
    pose = convert(rot0, xyz0, parameterization="euler_angles", convention="ZXY")
    s, t = drr.detector(pose, None)
    L = (t - s).norm(dim=-1)
    P = drr.detector.calibration(drr.detector.target)[0]
    Mw = (pose.matrix @ drr.detector._reorient)[:, :3, :]
    s2, t2, L2 = ops.raygen_forward(Mw, drr._affine_inverse.reshape(-1, 4, 4)[0, :3, :], P)
    assert (drr.affine_inverse(t) - t2).abs().max().item() < 2e-4
    assert (drr.affine_inverse(s) - s2).abs().max().item() < 2e-4
    assert (L - L2).abs().max().item() < 2e-4


def test_volume_gradient_bricks_full_size(gpu, big):
    """Brick volume gradient (LDS accumulation, no global atomics) at 512^3: equals the
    re-walk with global atomics, is the adjoint of the forward, and is deterministic up to
    the LDS summation order."""
    drr, s, t, L = big
    V = drr.density
    go = torch.rand(s.shape[0], 256 * 256, device=gpu, generator=torch.Generator(gpu).manual_seed(1))
    ref = ops.siddon_backward_volume(V, s, t, L, go, det=(256, 256))
    out = ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (256, 256))
    # (a voxel's gradient is a sum of a few w * dalpha; at 512^3 dalpha ~ 7e-4 carries an
    # ulp(alpha) / dalpha ~ 1e-4 rounding error, and the two walks round differently)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 5e-4
    img = ops.siddon_forward_bricks(V, s, t, L, (256, 256))[0]
    lhs, rhs = (out.double() * V.double()).sum().item(), (go.double() * img.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * abs(rhs)
    again = ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (256, 256))
    assert rel_err(again.cpu().numpy(), out.cpu().numpy()) < 2e-6


def test_trilinear_bricks_full_size(gpu, big):
    """Volume-stationary trilinear kernels at 512^3 / 256^2 / 200 samples: same image and
    volume gradient as the per-ray marcher; the gradient is the adjoint of the forward."""
    from diffdrr_amd.renderers import get_alpha_minmax

    drr, s, t, L = big
    V = drr.density
    lo, hi = get_alpha_minmax(s, t, torch.tensor(V.shape, device=gpu), 0.5, 1e-8)
    amin, amax = lo.min().reshape(1).contiguous(), hi.max().reshape(1).contiguous()
    P = 200
    ref = ops.trilinear_forward(V, s, t, L, amin, amax, n_points=P, det=(256, 256))
    out = ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (256, 256), n_points=P)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 5e-6
    go = torch.rand(s.shape[0], 256 * 256, device=gpu, generator=torch.Generator(gpu).manual_seed(3))
    gref = ops.trilinear_backward(V, s, t, L, go, amin, amax, n_points=P, want_rays=False,
                                  want_img=False, want_alpha=False, want_volume=True,
                                  det=(256, 256))["g_volume"]
    gout = ops.trilinear_backward_volume_bricks(V.shape, s, t, L, go, amin, amax, (256, 256),
                                                n_points=P)
    assert rel_err(gout.cpu().numpy(), gref.cpu().numpy()) < 2e-5
    lhs = (gout.double() * V.double()).sum().item()
    rhs = (go.double() * out.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * abs(rhs)
    # forward + record: same image; ray / range gradients from the record equal the re-march
    out2, aux = ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (256, 256), n_points=P,
                                             want_aux=True)
    assert rel_err(out2.cpu().numpy(), ref.cpu().numpy()) < 5e-6
    rec = ops.trilinear_backward_rays(aux, go, s, t, L, amin, amax, n_points=P)
    rem = ops.trilinear_backward(V, s, t, L, go, amin, amax, n_points=P, det=(256, 256))
    for k in ("g_target", "g_source", "g_img"):
        assert rel_err(rec[k].cpu().numpy(), rem[k].cpu().numpy()) < 2e-5, k
    ga, gb = rec["g_alpha"].double().sum((0, 1)), rem["g_alpha"].double().sum((0, 1))
    assert ((ga - gb).abs() / gb.abs()).max().item() < 1e-4


def test_volume_gradient_bricks_fixed_point_and_float_paths(gpu):
    """The brick volume gradients accumulate in int32 fixed point when a bound on the
    voxel sums exists and in float otherwise (source inside / next to the volume): both
    paths against the per-ray kernels with global atomics, Siddon and trilinear, on a
    volume that is not a multiple of the brick edge; gradients of wildly different scale."""
    from diffdrr_amd.renderers import get_alpha_minmax

    H, W = 24, 31
    subject = make_subject(torch.rand(40, 70, 33, generator=torch.Generator().manual_seed(1)),
                           spacing=(1.0, 1.5, 2.0))
    drr = DRR(subject, sdd=300.0, height=H, width=W, delx=2.0).to(gpu)
    V = drr.density
    cases = {
        "far": (torch.tensor([[0.01, 0.02, -0.01], [0.6, -0.4, 0.9], [1.5, 0.2, 0.1]]),
                torch.tensor([[0.3, 200.0, 0.2], [4.0, 180.0, -6.0], [1.0, 210.0, 2.0]])),
        "inside": (torch.tensor([[0.2, 0.1, 0.0], [0.0, 0.3, 0.1]]),
                   torch.tensor([[2.0, 5.0, -3.0], [0.0, 40.0, 1.0]])),
    }
    for name, (rot, xyz) in cases.items():
        s, t, L = voxel_rays(drr, rot.to(gpu), xyz.to(gpu))
        for scale in (1.0, 1e-6, 1e5):
            go = scale * torch.randn(s.shape[0], H * W, device=gpu,
                                     generator=torch.Generator(gpu).manual_seed(2))
            ref = ops.siddon_backward_volume(V, s, t, L, go, det=(H, W))
            out = ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (H, W))
            assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 2e-5, (name, scale)
            lo, hi = get_alpha_minmax(s, t, torch.tensor(V.shape, device=gpu), 0.5, 1e-8)
            amin, amax = lo.min().reshape(1).contiguous(), hi.max().reshape(1).contiguous()
            tref = ops.trilinear_backward(V, s, t, L, go, amin, amax, n_points=70,
                                          want_rays=False, want_img=False, want_alpha=False,
                                          want_volume=True, det=(H, W))["g_volume"]
            tout = ops.trilinear_backward_volume_bricks(V.shape, s, t, L, go, amin, amax, (H, W),
                                                        n_points=70)
            assert rel_err(tout.cpu().numpy(), tref.cpu().numpy()) < 2e-5, (name, scale)
    # the fixed-point path is bit-reproducible
    s, t, L = voxel_rays(drr, cases["far"][0].to(gpu), cases["far"][1].to(gpu))
    go = torch.randn(3, H * W, device=gpu, generator=torch.Generator(gpu).manual_seed(5))
    a = ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (H, W))
    b = ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (H, W))
    assert torch.equal(a, b)


@pytest.mark.parametrize("renderer", ["siddon", "trilinear"])
def test_reference_rays_through_swapped_renderers(gpu, renderer):
    """INTEGRATION.md section 2 on the device: the world-space rays of an unmodified reference
    ``DRR`` (committed fixture) through ``diffdrr_amd.Siddon`` / ``Trilinear`` with an untrusted
    ``detector_shape``, image and ray gradients against the reference's own."""
    from conftest import check_reference_rays_through_swapped_renderer

    check_reference_rays_through_swapped_renderer(renderer, gpu, ops)


def test_metrics_match_reference_on_the_gpu(gpu):
    """ddrr_ncc_* and ddrr_sobel_* through diffdrr_amd.metrics against the reference's values and
    autograd gradients (tests/golden/metrics.npz: NCC whole image / patch / multiscale,
    gradient-NCC with and without the Gaussian)."""
    from conftest import check_metrics_against_reference

    check_metrics_against_reference(gpu)


def test_brick_kernels_many_poses_multi_chunk(gpu):
    """More poses than one pose-table chunk (32): the brick kernels walk the batch in chunks
    with leftovers carried across them; B = 75 -> 3 chunks, against the per-ray kernels."""
    H, W, B = 20, 28, 75
    drr = DRR(synthetic_subject(50, kind="noise", seed=0), sdd=400.0, height=H, width=W,
              delx=1.5).to(gpu)
    g = torch.Generator().manual_seed(12)
    rot = ((torch.rand(B, 3, generator=g) - 0.5) * 2.0).to(gpu)
    xyz = (torch.tensor([0.0, 280.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 40).to(gpu)
    s, t, L = voxel_rays(drr, rot, xyz)
    V = drr.density
    ref, aux_ref, _ = ops.siddon_forward(V, s, t, L, want_aux=True)
    out, aux = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=True)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    assert torch.allclose(ops.record_planes(aux, B, H * W)[0], aux_ref[..., 0], rtol=1e-4, atol=1e-5)
    go = torch.rand(B, H * W, device=gpu, generator=torch.Generator(gpu).manual_seed(1))
    gref = ops.siddon_backward_volume(V, s, t, L, go)
    gout = ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (H, W))
    assert rel_err(gout.cpu().numpy(), gref.cpu().numpy()) < 2e-5


def test_trilinear_mask_golden(gpu):
    """Trilinear mask_to_channels (renderers.py:242-252) against the reference fixture."""
    g = golden("trilinear_mask")
    vol, src, tgt, img = dev_inputs(g, gpu)
    mask = torch.from_numpy(g["mask"]).to(gpu)
    # The first sample of the ray that sets the batch-global alphamin lies exactly on the
    # volume's face, where the label lookup is discontinuous: the fixture pins the
    # reference's CPU evaluation of the range, so the range is evaluated on the CPU here
    # (on the GPU torch's division may differ in the last bit, as it would for the
    # reference itself).
    from diffdrr_amd.renderers import get_alpha_minmax

    lo, hi = get_alpha_minmax(src.cpu(), tgt.cpu(), torch.tensor(vol.shape), 0.5, 1e-8)
    rng = {"alphamin": lo.min().to(gpu), "alphamax": hi.max().to(gpu)}
    out = Trilinear()(vol, src, tgt, img, n_points=40, mask=mask, **rng)
    assert out.shape == g["out_f32"].shape
    assert rel_err(out.cpu().numpy(), g["out_f32"]) < FWD_TOL
    plain = Trilinear()(vol, src, tgt, img, n_points=40, **rng)
    assert rel_err(out.sum(1, keepdim=True).cpu().numpy(), plain.cpu().numpy()) < 1e-5
    # backward (ddrr_trilinear_backward_channels) against the reference's autograd; the
    # range stays a differentiable function of the rays (its arg-min / arg-max ray)
    src_c, tgt_c = src.cpu().requires_grad_(), tgt.cpu().requires_grad_()
    lo, hi = get_alpha_minmax(src_c, tgt_c, torch.tensor(vol.shape), 0.5, 1e-8)
    src_g, tgt_g = src_c.to(gpu), tgt_c.to(gpu)
    vol_g, img_g = vol.clone().requires_grad_(), img.clone().requires_grad_()
    out = Trilinear()(vol_g, src_g, tgt_g, img_g, n_points=40, mask=mask,
                      alphamin=lo.min().to(gpu), alphamax=hi.max().to(gpu))
    go = torch.from_numpy(g["grad_out_f32"]).to(gpu)
    grads = torch.autograd.grad(out, [src_c, tgt_c, img_g, vol_g], go)
    for name, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.cpu().numpy(), g[name + "_f32"]) < GRAD_TOL, name


def test_fused_ncc_kernels(gpu):
    """ddrr_ncc_forward / _backward (reference metrics.py:21-44) on the GPU against the
    PyTorch formula, on DRR-like images (large mean): values and gradients."""
    from diffdrr_amd import NormalizedCrossCorrelation2d

    g = torch.Generator().manual_seed(0)
    ncc = NormalizedCrossCorrelation2d()
    fixed = (350 + 60 * torch.rand(1, 1, 256, 256, generator=g)).to(gpu)
    x2 = (330 + 80 * torch.rand(6, 1, 256, 256, generator=g)).to(gpu)
    w = torch.rand(6, generator=g).to(gpu)

    def formula(a, b):
        return ((ncc.norm(a) * ncc.norm(b)).flatten(1).sum(1)) / (256 * 256)

    b1, b2 = x2.clone().requires_grad_(), x2.clone().requires_grad_()
    fused = ncc(fixed.expand(6, -1, -1, -1), b1)
    ref = formula(fixed.expand(6, -1, -1, -1).double(), b2.double())
    assert torch.allclose(fused.double(), ref, atol=5e-6)
    (fused * w).sum().backward()
    (ref * w.double()).sum().backward()
    assert rel_err(b1.grad.cpu().numpy(), b2.grad.cpu().numpy()) < 1e-4


def test_fused_euler_pose_on_gpu(gpu):
    """Raw Euler parameters -> DRR through the fused pose kernel (ddrr_pose_euler_forward)
    against the same call with a RigidTransform built by `convert` in PyTorch."""
    from diffdrr_amd.pose import RigidTransform, euler_world_pose

    drr = DRR(synthetic_subject(64, kind="phantom", seed=0), sdd=500.0, height=48, delx=2.0).to(gpu)
    g = torch.Generator().manual_seed(21)
    rot0 = ((torch.rand(9, 3, generator=g) - 0.5) * 3.0).to(gpu)
    xyz0 = (torch.tensor([0.0, 350.0, 0.0]) + (torch.rand(9, 3, generator=g) - 0.5) * 50).to(gpu)
    for conv in ("ZXY", "XYZ", "YXY"):
        r1, x1 = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
        r2, x2 = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
        Mw = euler_world_pose(r1, x1, conv, drr.detector._reorient)
        ref = RigidTransform(drr.detector._reorient).compose(
            convert(r2, x2, parameterization="euler_angles", convention=conv)).matrix[:, :3, :]
        assert torch.allclose(Mw, ref, rtol=1e-5, atol=2e-4), conv
        w = torch.rand(9, 3, 4, device=gpu, generator=torch.Generator(gpu).manual_seed(2))
        (Mw * w).sum().backward()
        (ref * w).sum().backward()
        assert rel_err(r1.grad.cpu().numpy(), r2.grad.cpu().numpy()) < 1e-4, conv
        assert rel_err(x1.grad.cpu().numpy(), x2.grad.cpu().numpy()) < 1e-5, conv
    a = drr(rot0, xyz0, parameterization="euler_angles", convention="ZXY")
    b = drr(convert(rot0, xyz0, parameterization="euler_angles", convention="ZXY"))
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


def test_packed_record_on_gpu(gpu, big):
    """Opt-in fixed-point record (csrc/record_pack.h) at 512^3 / 256^2: the same ray gradients
    as the fp32 record up to its resolution, and bit-identical from run to run (integer
    atomics), which the fp32 record is not."""
    drr, s, t, L = big
    V = drr.density
    vmax = ops.volume_absmax(V)
    out_f, aux_f = ops.siddon_forward_bricks(V, s, t, L, (256, 256), want_aux=True)
    out_p, aux_p = ops.siddon_forward_bricks(V, s, t, L, (256, 256), want_aux=True,
                                             record_vmax=vmax)
    assert aux_p.shape[0] == 7 and rel_err(out_p.cpu().numpy(), out_f.cpu().numpy()) < 1e-6
    go = torch.rand(out_f.shape, device=gpu, generator=torch.Generator(gpu).manual_seed(5))
    gf = ops.siddon_backward_rays(aux_f, go, s, t, L)
    gp = ops.siddon_backward_rays(aux_p, go, s, t, L)
    for a, b in zip(gp, gf):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 5e-5
    _, aux_p2 = ops.siddon_forward_bricks(V, s, t, L, (256, 256), want_aux=True, record_vmax=vmax)
    assert torch.equal(aux_p[:4].view(torch.int32), aux_p2[:4].view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("name,tag", conftest.general_case_ids())
def test_general_path_matches_the_reference(gpu, name, tag):
    """The materialising general path on the GPU (csrc/general_rays.hip; float32 and float64):
    every keyword combination outside the fused kernels against fixtures of the unmodified
    reference, outputs and autograd gradients."""
    conftest.check_general_case(name, tag, gpu)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,det,P", [((96, 70, 133), (50, 37), 300), ((64, 64, 61), (40, 40), 150)])
def test_trilinear_channels_on_bricks(gpu, dims, det, P):
    """The marcher's mask_to_channels on the volume-stationary bricks against the per-ray channel
    kernel (itself pinned to the reference's fixture), the plain march, and through the module."""
    conftest.check_trilinear_channels_on_bricks(gpu, dims, det, P)


def test_filter_intersections_outside_volume(gpu):
    """the flag the reference itself cannot run (TypeError,
    renderers.py:118 vs :124), against the fixture of its intended semantics."""
    conftest.check_filter_intersections_outside_volume(gpu)


def test_fused_ncc_step(gpu):
    """ddrr_pose_raygen_forward / ddrr_siddon_ncc_forward / ddrr_siddon_ncc_backward_pose through
    DRR.ncc against the launches they fuse."""
    conftest.check_fused_ncc_step(gpu)


@pytest.mark.gpu
def test_euler_inference_path(gpu):
    conftest.check_euler_inference_path(gpu)


@pytest.mark.gpu
def test_pose_adam_matches_torch_adam(gpu):
    conftest.check_pose_adam(gpu)


@pytest.mark.parametrize("name", sorted(conftest.SPARSE_CASES))
def test_subsample_and_patches_match_reference_on_the_bricks(gpu, monkeypatch, name):
    """``p_subsample`` / ``patch_size`` against the fixture of the unmodified
    reference, on the device, with the C-ABI entries the render went through recorded: brick
    kernels, no per-ray forward kernel."""
    calls = []
    launch = ops._launch
    monkeypatch.setattr(ops, "_launch", lambda n, d, *a: (calls.append(n), launch(n, d, *a))[1])
    conftest.check_sparse_lever(name, gpu, ops, calls)


@pytest.mark.parametrize("renderer", ["siddon", "trilinear"])
def test_subsample_list_and_patched_render_vs_oracle(gpu, renderer):
    """The same levers at a size where the brick grid matters (a CT-like 192 x 192 x 67 volume through
    transform_hu_to_density, 112 x 96 detector, 3 poses) against the ORACLE on exactly the rays the
    reference would render: a 10 % subsample list (the oracle gets the listed rays, the marcher its
    range over them) and a patched render (the oracle chunk by chunk, the marcher's range per
    chunk, 6 ragged chunks); images 1e-4 of the image scale vs the oracle's fp32 and fp64 (gradients:
    the reference's fixture above)."""
    import oracle
    from diffdrr_amd.data import ct_like_hu_volume, make_subject, transform_hu_to_density

    vol = transform_hu_to_density(ct_like_hu_volume((192, 192, 67), seed=4))
    H, W, B, P = 112, 96, 3, 150
    kw = dict(n_points=P) if renderer == "trilinear" else {}
    g = torch.Generator().manual_seed(9)
    rot = ((torch.rand(B, 3, generator=g) - 0.5) * 1.2).to(gpu)
    xyz = (torch.tensor([0.0, 500.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 30).to(gpu)
    v64 = vol.numpy().astype(np.float64)

    def oracle_render(drr, pix, dtype):
        """the oracle on the rays `pix` of every pose (range: over exactly those rays)"""
        with torch.no_grad():
            pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
            src, tgt = drr.detector(pose, None)
            L = (tgt - src).norm(dim=-1)
            s, t = drr.affine_inverse(src), drr.affine_inverse(tgt)
        a = [x.cpu().numpy().astype(dtype) for x in (s, t[:, pix] if pix is not None else t,
                                                      L[:, pix] if pix is not None else L)]
        fn = oracle.siddon if renderer == "siddon" else oracle.trilinear
        return fn(v64.astype(dtype), *a, **kw)["out"].reshape(B, -1)

    # (a) a subsample list
    torch.manual_seed(77)
    sub = DRR(make_subject(vol, spacing=(1.0, 1.0, 2.0)), sdd=1000.0, height=H, width=W, delx=2.2,
              renderer=renderer, p_subsample=0.1, reshape=False).to(gpu)
    n = int(H * W * 0.1)
    assert len(sub.detector.subsamples[-1]) == n
    with torch.no_grad():
        img = sub(rot, xyz, parameterization="euler_angles", convention="ZXY", **kw)
    assert img.shape == (B, 1, n)
    # (the detector's own `target` IS the subsample: the oracle renders what Detector.forward hands over)
    r32, r64 = oracle_render(sub, None, np.float32), oracle_render(sub, None, np.float64)
    mine = img.reshape(B, n).cpu().numpy()
    assert rel_err(mine, r64) < 1e-4 and rel_err(mine, r32) < 1e-4, (rel_err(mine, r64), rel_err(mine, r32))
    # reshape=True scatters the same values into zeros (drr.py:142-147)
    sub.reshape = True
    with torch.no_grad():  # (the same batch: the marcher's range is taken over the poses of a call)
        full = sub(rot, xyz, parameterization="euler_angles", convention="ZXY", **kw)
    idx = torch.tensor(sub.detector.subsamples[-1], device=gpu)
    assert full.shape == (B, 1, H, W) and int((full[0] != 0).sum()) <= n
    # (two launches: the float atomics of the partial integrals arrive in another order)
    assert torch.allclose(full.reshape(B, -1)[:, idx], img[:, 0], rtol=1e-5, atol=0)
    # (b) a patched render: 6 ragged chunks of ceil(10752 / 6) rays
    pat = DRR(make_subject(vol, spacing=(1.0, 1.0, 2.0)), sdd=1000.0, height=H, width=W, delx=2.2,
              renderer=renderer, patch_size=42).to(gpu)
    assert pat.n_patches == 6
    with torch.no_grad():
        pimg = pat(rot, xyz, parameterization="euler_angles", convention="ZXY", **kw).reshape(B, H * W)
    size = -(-H * W // 6)
    for dtype in (np.float32, np.float64):
        ref = np.concatenate([oracle_render(pat, np.arange(a, min(a + size, H * W)), dtype)
                              for a in range(0, H * W, size)], axis=1)
        assert rel_err(pimg.cpu().numpy(), ref) < 1e-4, (renderer, dtype, rel_err(pimg.cpu().numpy(), ref))
    if renderer == "trilinear":
        # (the chunks' ranges differ: unpatched, the image is another one)
        whole = DRR(make_subject(vol, spacing=(1.0, 1.0, 2.0)), sdd=1000.0, height=H, width=W, delx=2.2,
                    renderer=renderer).to(gpu)
        with torch.no_grad():
            wimg = whole(rot, xyz, parameterization="euler_angles", convention="ZXY", **kw).reshape(B, H * W)
        assert rel_err(wimg.cpu().numpy(), ref) > 1e-4


def test_patch_ncc_kernels_against_the_composition(gpu):
    """ddrr_ncc_patch_forward / _backward on the device: MultiscaleNormalizedCrossCorrelation2d's local
    scale without `to_patches` (VERDICT r05 missing 5)."""
    conftest.check_patch_ncc_against_composition(gpu)


def test_blur_sobel_kernels_against_the_composition(gpu):
    """ddrr_blur_sobel_forward / _backward on the device: the Gaussian blur + Sobel pair in front of
    GradientNormalizedCrossCorrelation2d (reference metrics.py:88-93) without the k x k convolution."""
    conftest.check_blur_sobel_against_composition(gpu)


def test_euler_differentiable_path(gpu, monkeypatch):
    """ddrr_siddon_backward_pose_euler and the three-launch forward behind `drr(rot, xyz,
    parameterization="euler_angles")` with a gradient wanted, against the composition they replace."""
    from diffdrr_amd import ops

    calls, launch = [], ops._launch
    monkeypatch.setattr(ops, "_launch", lambda n, d, *a: (calls.append(n), launch(n, d, *a))[1])
    conftest.check_euler_differentiable_path(gpu, ops, calls)
