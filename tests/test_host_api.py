"""The PyTorch-side mirror of the reference's API (pose, Detector, DRR,
Registration, NCC) against goldens produced by the unmodified reference.  DRR
rendering runs through the host emulation of the kernel cores (conftest
``emulated_ops``): this checks the Python / autograd wiring on CPU; the same
module is checked on the real kernels in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import conftest
from conftest import golden, rel_err
from diffdrr_amd import DRR, Detector, Registration, RigidTransform, convert
from diffdrr_amd import pose as P
from diffdrr_amd.data import Image, Subject, make_subject
from diffdrr_amd.metrics import NormalizedCrossCorrelation2d

T = torch.from_numpy


# ------------------------------------------------------------------- pose

POSE_CASES = ["axis_angle", "euler_angles", "euler_angles_deg", "quaternion", "rotation_6d",
              "rotation_9d", "rotation_10d", "quaternion_adjugate", "se3_log_map"]
POSE_KW = {"euler_angles": {"convention": "ZXY"},
           "euler_angles_deg": {"convention": "XYZ", "degrees": True}}


@pytest.mark.parametrize("name", POSE_CASES)
def test_convert_matches_reference(name):
    g = golden("pose")
    param = "euler_angles" if name.startswith("euler") else name
    kw = POSE_KW.get(name, {})
    Tm = convert(T(g[name + "_in"]), T(g["translation"]), parameterization=param, **kw)
    ref = g[name + "_matrix"]
    if name == "rotation_10d":
        # eigenvector sign is arbitrary, but q and -q are the same rotation
        assert rel_err(Tm.matrix.numpy(), ref) < 1e-4
    else:
        assert rel_err(Tm.matrix.numpy(), ref) < 1e-5
    # round trip: parameters recovered from the matrix rebuild the same matrix
    back_kw = {"convention": kw["convention"]} if param == "euler_angles" else {}
    rot, xyz = Tm.convert(param, **back_kw)
    T2 = convert(rot, xyz, parameterization=param, **back_kw)
    assert rel_err(T2.matrix.numpy(), ref) < 1e-4
    assert rel_err(xyz.numpy(), g[name + "_back_xyz"]) < 1e-4
    if param in ("euler_angles", "axis_angle", "rotation_6d", "se3_log_map"):
        assert rel_err(rot.numpy(), g[name + "_back_rot"]) < 1e-4


def test_rigid_transform_algebra():
    g = golden("pose")
    t = T(g["translation"])
    A = convert(T(g["euler_angles_in"]), t, parameterization="euler_angles", convention="ZXY")
    B = convert(T(g["axis_angle_in"]), -t, parameterization="axis_angle")
    assert rel_err(A(T(g["apply_pts"])).numpy(), g["apply_out"]) < 1e-5
    assert rel_err(A.compose(B).matrix.numpy(), g["compose_matrix"]) < 1e-5
    assert rel_err(A.inverse().matrix.numpy(), g["inverse_matrix"]) < 1e-5
    assert len(A) == 4 and A[1:3].matrix.shape == (2, 4, 4)
    assert RigidTransform(A) is A
    with pytest.raises(ValueError):
        convert(t, t, parameterization="euler_angles")
    with pytest.raises(ValueError):
        convert(t, t, parameterization="nope")


@pytest.mark.parametrize("convention", ["XYZ", "XZY", "YXZ", "YZX", "ZXY", "ZYX",
                                        "XYX", "XZX", "YXY", "YZY", "ZXZ", "ZYZ"])
def test_euler_round_trip_all_conventions(convention):
    g = torch.Generator().manual_seed(3)
    a = (torch.rand(16, 3, generator=g) - 0.5) * 2.0
    if convention[0] == convention[2]:
        a[:, 1] = a[:, 1].abs() + 0.1  # proper Euler: middle angle in (0, pi)
    R = P.euler_angles_to_matrix(a, convention)
    assert torch.allclose(P.matrix_to_euler_angles(R, convention), a, atol=1e-5)
    assert torch.allclose(R @ R.mT, torch.eye(3).expand(16, 3, 3), atol=1e-5)


# --------------------------------------------------------------- detector

def _subject_a(g, with_mask=True):
    vol = T(g["volume"])
    mask = Image(T(g["mask"]).unsqueeze(0), g["affine"]) if with_mask else None
    return Subject(Image(vol.unsqueeze(0), g["affine"]), Image(vol.unsqueeze(0), g["affine"]),
                   T(g["reorient"]), mask)


def _geo(g, prefix="geo_"):
    keys = [k for k in g.files if k.startswith(prefix)]
    geo = {k[len(prefix):]: g[k].item() for k in keys}
    for k in ("height", "width"):
        if k in geo:
            geo[k] = int(geo[k])
    return geo


def test_detector_rays_match_reference():
    g = golden("drr_module")
    geo = _geo(g)
    det = Detector(geo["sdd"], geo["height"], geo["width"], geo["delx"], geo["dely"], geo["x0"],
                   geo["y0"], T(g["reorient"]), reverse_x_axis=True)
    pose = convert(T(g["rot"]), T(g["xyz"]), parameterization="euler_angles", convention="ZXY")
    assert rel_err(pose.matrix.numpy(), g["pose_matrix_f32"]) < 1e-6
    source, target = det(pose, None)
    assert source.shape == (3, 1, 3) and target.shape == (3, geo["height"] * geo["width"], 3)
    assert rel_err(source.numpy(), g["det_source_f32"]) < 1e-6
    assert rel_err(target.numpy(), g["det_target_f32"]) < 1e-6
    assert abs(det.x0 + geo["x0"]) < 1e-6 and abs(det.sdd - geo["sdd"]) < 1e-4

    # odd-sized detector, reverse_x_axis=False, PA orientation
    geo2 = _geo(g, "b_geo_")
    det2 = Detector(geo2["sdd"], geo2["height"], geo2["width"], geo2["delx"], geo2["delx"], 0.0,
                    0.0, T(g["b_reorient"]), reverse_x_axis=False)
    pose2 = convert(T(g["b_rot"]), T(g["b_xyz"]), parameterization="euler_angles",
                    convention="ZXY")
    s2, t2 = det2(pose2, None)
    assert rel_err(s2.numpy(), g["b_det_source"]) < 1e-6
    assert rel_err(t2.numpy(), g["b_det_target"]) < 1e-6


# -------------------------------------------------------------------- DRR

@pytest.mark.parametrize("renderer,kw", [("siddon", {}), ("trilinear", {"n_points": 60})])
def test_drr_module_matches_reference(emulated_ops, renderer, kw):
    g = golden("drr_module")
    drr = DRR(_subject_a(g), renderer=renderer, **_geo(g))
    rot = T(g["rot"]).clone().requires_grad_()
    xyz = T(g["xyz"]).clone().requires_grad_()
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", **kw)
    assert img.shape == g[f"{renderer}_img_f32"].shape  # (B, 1, H, W)
    assert rel_err(img.detach().numpy(), g[f"{renderer}_img_f32"]) < 1e-4
    assert rel_err(img.detach().numpy(), g[f"{renderer}_img_f64"]) < 1e-4
    img.backward(T(g[f"{renderer}_grad_out_f32"]))
    assert rel_err(rot.grad.numpy(), g[f"{renderer}_g_rot_f64"]) < 1e-3
    assert rel_err(xyz.grad.numpy(), g[f"{renderer}_g_xyz_f64"]) < 1e-3


def test_drr_mask_to_channels_and_patches(emulated_ops):
    g = golden("drr_module")
    geo = _geo(g)
    drr = DRR(_subject_a(g), **geo)
    rot, xyz = T(g["rot"]), T(g["xyz"])
    with torch.no_grad():
        ch = drr(rot, xyz, parameterization="euler_angles", convention="ZXY",
                 mask_to_channels=True)
        plain = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    assert ch.shape == g["siddon_channels_f32"].shape
    assert rel_err(ch.numpy(), g["siddon_channels_f32"]) < 1e-4
    assert rel_err(ch.sum(1, keepdim=True).numpy(), plain.numpy()) < 1e-5
    drr_p = DRR(_subject_a(g), patch_size=2, **geo)
    with torch.no_grad():
        patched = drr_p(rot, xyz, parameterization="euler_angles", convention="ZXY")
    assert rel_err(patched.numpy(), g["siddon_patched_f32"]) < 1e-4
    # (patched renders take the general path, `plain` the fused pose -> rays kernels, whose
    # rotation matrix differs from torch's matmul chain in the last bit)
    assert rel_err(patched.numpy(), plain.numpy()) < 1e-5
    drr.fuse_ray_generation = False
    with torch.no_grad():
        general = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    # Siddon is per-ray independent: patches change nothing but the kernel that renders them
    # (ray lists: the generic walk; the whole detector: the brick walk, other alpha arithmetic)
    assert rel_err(patched.numpy(), general.numpy()) < 2e-5


def test_drr_odd_detector_pa(emulated_ops):
    g = golden("drr_module")
    vol = T(g["b_volume"])
    subject = Subject(Image(vol.unsqueeze(0), g["b_affine"]), Image(vol.unsqueeze(0), g["b_affine"]),
                      T(g["b_reorient"]))
    drr = DRR(subject, reverse_x_axis=False, **_geo(g, "b_geo_"))
    pose = convert(T(g["b_rot"]), T(g["b_xyz"]), parameterization="euler_angles",
                   convention="ZXY")
    with torch.no_grad():
        img = drr(pose)
    assert rel_err(img.numpy(), g["b_img"]) < 1e-4


def test_drr_api_surface(emulated_ops):
    subject = make_subject(torch.rand(8, 9, 10), spacing=(1.0, 2.0, 1.5))
    drr = DRR(subject, sdd=100.0, height=6, width=4, delx=2.0, p_subsample=0.5)
    assert drr.detector.n_subsample == 12 and drr.device.type == "cpu"
    assert drr.dtype == torch.float32 and drr.affine.matrix.shape == (1, 4, 4)
    rot, xyz = torch.zeros(2, 3), torch.tensor([[0.0, 60.0, 0.0]] * 2)
    with torch.no_grad():
        img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    assert img.shape == (2, 1, 6, 4) and (img != 0).sum() <= 24
    drr.set_intrinsics_(height=5, width=7, n_subsample=None)
    assert (drr.detector.height, drr.detector.width) == (5, 7)
    drr.rescale_detector_(2.0)
    assert drr.detector.height == 10 and abs(drr.detector.delx - 1.0) < 1e-6
    with pytest.raises(ValueError):
        DRR(subject, sdd=1.0, height=2, delx=1.0, renderer="nope")
    with pytest.raises(ValueError):
        DRR(subject, sdd=1.0, height=2, delx=1.0, reducefn="median")(
            rot, xyz, parameterization="euler_angles", convention="ZXY")
    # projection helpers are inverse of each other on the detector plane
    drr2 = DRR(subject, sdd=100.0, height=6, delx=2.0)
    pose = convert(rot[:1], xyz[:1], parameterization="euler_angles", convention="ZXY")
    px = torch.tensor([[[1.0, 2.0], [4.0, 3.0]]])
    world = drr2.inverse_projection(pose, px.clone())
    assert torch.allclose(drr2.perspective_projection(pose, world), px, atol=1e-3)


# ----------------------------------------------------------- registration

def test_ncc_matches_reference():
    g = golden("registration")
    a, b = T(g["ncc_a"]), T(g["ncc_b"])
    assert rel_err(NormalizedCrossCorrelation2d()(a, b).numpy(), g["ncc_ab"]) < 1e-5
    assert rel_err(NormalizedCrossCorrelation2d(patch_size=5)(a, b).numpy(),
                   g["ncc_ab_patch5"]) < 1e-5


def test_ncc_backward_reads_an_expanded_gradient_in_place(emulated_ops):
    """`.sum().backward()` hands the fused NCC an expanded scalar (stride 0): ddrr_ncc_backward
    reads it in place (g_stride = 0) and gives what a materialised vector of ones gives."""
    g = torch.Generator().manual_seed(5)
    fixed = torch.rand(1, 1, 12, 9, generator=g) + 2
    grads = []
    for weights in (None, torch.ones(4)):
        x = (torch.rand(4, 1, 12, 9, generator=torch.Generator().manual_seed(6)) + 1).requires_grad_()
        ncc = NormalizedCrossCorrelation2d()(fixed.expand(4, -1, -1, -1), x)
        (ncc.sum() if weights is None else (ncc * weights).sum()).backward()
        grads.append(x.grad.clone())
    assert torch.equal(grads[0], grads[1])
    x = (torch.rand(4, 1, 12, 9, generator=torch.Generator().manual_seed(6)) + 1).requires_grad_()
    NormalizedCrossCorrelation2d()(fixed.expand(4, -1, -1, -1), x).mean().backward()
    assert torch.allclose(x.grad * 4, grads[0], rtol=1e-6, atol=0)


def test_metrics_match_reference_values_and_gradients(emulated_ops):
    """NCC (whole image, patch-wise, multiscale) and gradient-NCC against the fixture made from the
    unmodified reference's metrics.py (host emulation of the Sobel / NCC kernels; GPU twin:
    tests/test_gpu_parity.py::test_metrics_match_reference_on_the_gpu)."""
    from conftest import check_metrics_against_reference

    check_metrics_against_reference(torch.device("cpu"))


@pytest.mark.parametrize("storage", ["f32", "q16", "q16p"])
@pytest.mark.parametrize("stop", [False, True])
def test_registration_trajectory_matches_reference(emulated_ops, stop, storage):
    """First SGD steps of the tutorial's registration loop: same losses and the
    same parameter trajectory as the reference (registration.ipynb:240-316).  The loop on this
    24^3 NOISE volume is chaotic from about step 5 (every fp32 path, the reference's included,
    then goes its own way): the exact fp32 bricks are pinned over the whole fixture, the 16-bit
    block-quantised bricks (a 1e-5 perturbation of the volume) over its first steps."""
    g = golden("registration")
    vol = T(g["volume"])
    subject = Subject(Image(vol.unsqueeze(0), g["affine"]), Image(vol.unsqueeze(0), g["affine"]),
                      T(g["reorient"]))
    geo = _geo(g)
    drr = DRR(subject, stop_gradients_through_grid_sample=stop, **geo)
    drr.renderer.brick_storage = storage
    gt = T(g["gt"])
    with torch.no_grad():
        mine = drr(T(g["true_rot"]), T(g["true_xyz"]), parameterization="euler_angles",
                   convention="ZXY")
    assert rel_err(mine.numpy(), g["gt"]) < 1e-4
    reg = Registration(drr, T(g["rot0"]).clone(), T(g["xyz0"]).clone(),
                       parameterization="euler_angles", convention="ZXY")
    crit = NormalizedCrossCorrelation2d()
    opt = torch.optim.SGD([{"params": [reg._rotation], "lr": 5e-2},
                           {"params": [reg._translation], "lr": 1e2}], maximize=True)
    tag = "stop" if stop else "full"
    for k in range(len(g[f"losses_{tag}"]) if storage == "f32" else 5):
        opt.zero_grad()
        loss = crit(gt, reg()).mean()
        loss.backward()
        assert abs(loss.item() - g[f"losses_{tag}"][k]) < 2e-3, (k, loss.item())
        assert rel_err(reg._rotation.detach().numpy(), g[f"rots_{tag}"][k]) < 2e-2
        assert rel_err(reg._translation.detach().numpy(), g[f"xyzs_{tag}"][k]) < 2e-3
        opt.step()
    assert g[f"losses_{tag}"][-1] > g[f"losses_{tag}"][0]  # the reference itself improves


@pytest.mark.parametrize("stop", [False, True])
@pytest.mark.parametrize("path", ["bricks", "generic"])
def test_fused_ray_generation_equals_general_path(emulated_ops, path, stop):
    """DRR.forward's fused entry (raygen kernel + renderer + pose-gradient kernel,
    csrc/raygen_core.h) against the general path (Detector + render in PyTorch, renderer
    on ray tensors): same image, same gradients w.r.t. the pose parameters and the volume."""
    from diffdrr_amd import DRR
    from diffdrr_amd.data import synthetic_subject

    drr = DRR(synthetic_subject(40, kind="noise", seed=0), sdd=400.0, height=22, width=30,
              delx=1.8, stop_gradients_through_grid_sample=stop)
    drr.renderer.grid_path = path
    drr.density.requires_grad_(not stop)
    rot0 = torch.tensor([[0.1, -0.2, 0.3], [0.6, 0.4, -0.5], [0.0, 0.01, 0.02]])
    xyz0 = torch.tensor([[3.0, 250.0, -2.0], [10.0, 230.0, 6.0], [0.5, 260.0, 0.3]])
    go = torch.rand(3, 1, 22, 30, generator=torch.Generator().manual_seed(4))
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
        res[fused] = (img.detach(), rot.grad, xyz.grad,
                      None if stop else drr.density.grad.clone())
    a, b = res[True], res[False]
    assert rel_err(a[0].numpy(), b[0].numpy()) < 1e-5
    assert rel_err(a[1].numpy(), b[1].numpy()) < 2e-3
    assert rel_err(a[2].numpy(), b[2].numpy()) < 2e-3
    if not stop:
        assert rel_err(a[3].numpy(), b[3].numpy()) < 1e-5


def test_fused_entry_is_skipped_when_a_general_feature_is_asked_for(emulated_ops):
    from diffdrr_amd import DRR
    from diffdrr_amd.data import synthetic_subject

    drr = DRR(synthetic_subject(24, kind="phantom", seed=0), sdd=300.0, height=12, delx=2.0)
    assert drr._fused_ok(False, {})
    assert not drr._fused_ok(True, {})                       # mask_to_channels without a mask
    assert not drr._fused_ok(False, {"align_corners": True})  # renderer kwargs
    drr.renderer.reducefn = "max"
    assert not drr._fused_ok(False, {})
    drr2 = DRR(synthetic_subject(24, kind="phantom", seed=0), sdd=300.0, height=12, delx=2.0,
               renderer="trilinear")
    assert drr2._fused_ok(False, {}) and drr2._fused_ok(False, {"n_points": 50})
    assert not drr2._fused_ok(False, {"n_points": 50, "align_corners": True})
    drr2.renderer.mode = "nearest"
    assert not drr2._fused_ok(False, {})
    # patch_size / p_subsample stay on the fused entries since round 6 (DRR._render_sparse) -- except
    # for callers that read the record of the whole grid (DRR.ncc) with a subsample
    drr3 = DRR(synthetic_subject(24, kind="phantom", seed=0), sdd=300.0, height=12, delx=2.0,
               patch_size=6)
    assert drr3._fused_ok(False, {}) and drr3._fused_ok(False, {}, None, dense_only=True)
    drr4 = DRR(synthetic_subject(24, kind="phantom", seed=0), sdd=300.0, height=12, delx=2.0,
               p_subsample=0.5)
    assert drr4._fused_ok(False, {}) and not drr4._fused_ok(False, {}, None, dense_only=True)
    drr5 = DRR(synthetic_subject(24, kind="phantom", seed=0), sdd=300.0, height=12, delx=2.0,
               patch_size=13)  # (more than the detector holds: the reference's chunk(0) raises)
    assert not drr5._fused_ok(False, {})


@pytest.mark.parametrize("stop", [False, True])
def test_fused_mask_to_channels_equals_the_general_path(emulated_ops, stop):
    """`mask_to_channels=True` through the fused entry (pose -> rays -> channel images in
    kernels, the ray-generation adjoint restated on tensors in the backward) against the
    general path (Detector.forward + render + autograd of the torch ops in between): images,
    pose gradients and the volume gradient."""
    from diffdrr_amd import DRR
    from diffdrr_amd.data import synthetic_subject

    res = {}
    for fused in (True, False):
        drr = DRR(synthetic_subject((24, 30, 20), kind="phantom", seed=3, n_labels=6), sdd=300.0,
                  height=14, width=11, delx=2.0, stop_gradients_through_grid_sample=stop)
        drr.fuse_ray_generation = fused
        assert drr._fused_ok(True, {}) == fused
        drr.density.requires_grad_()
        rot = torch.tensor([[0.2, -0.1, 0.3], [0.0, 0.4, -0.2]], requires_grad=True)
        xyz = torch.tensor([[3.0, 210.0, -2.0], [-4.0, 190.0, 5.0]], requires_grad=True)
        ch = drr(rot, xyz, parameterization="euler_angles", convention="ZXY",
                 mask_to_channels=True)
        w = torch.rand(ch.shape, generator=torch.Generator().manual_seed(4))
        (ch * w).sum().backward()
        res[fused] = (ch.detach(), rot.grad, xyz.grad,
                      None if stop else drr.density.grad.clone())
    a, b = res[True], res[False]
    assert a[0].shape == (2, 6, 14, 11)
    assert rel_err(a[0].numpy(), b[0].numpy()) < 1e-5
    assert rel_err(a[1].numpy(), b[1].numpy()) < 2e-3
    assert rel_err(a[2].numpy(), b[2].numpy()) < 2e-3
    if not stop:  # (the two paths' rays differ in the last bit: segment lengths to ~1e-5)
        assert rel_err(a[3].numpy(), b[3].numpy()) < 1e-4


@pytest.mark.parametrize("mask", [False, True])
def test_fused_trilinear_entry_equals_the_general_path(emulated_ops, mask):
    """The marcher through the fused entry (pose -> rays in kernels, the differentiable
    ray-generation op of renderers.py) against the general path: images, pose gradients --
    including the path through the batch-global marching range -- and the volume gradient."""
    from diffdrr_amd import DRR
    from diffdrr_amd.data import synthetic_subject

    res = {}
    for fused in (True, False):
        drr = DRR(synthetic_subject((24, 30, 20), kind="phantom", seed=3, n_labels=5), sdd=300.0,
                  height=14, width=11, delx=2.0, renderer="trilinear")
        drr.fuse_ray_generation = fused
        assert drr._fused_ok(mask, {"n_points": 70}) == fused
        drr.density.requires_grad_()
        rot = torch.tensor([[0.2, -0.1, 0.3], [0.0, 0.4, -0.2]], requires_grad=True)
        xyz = torch.tensor([[3.0, 210.0, -2.0], [-4.0, 190.0, 5.0]], requires_grad=True)
        img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", n_points=70,
                  mask_to_channels=mask)
        w = torch.rand(img.shape, generator=torch.Generator().manual_seed(4))
        (img * w).sum().backward()
        res[fused] = (img.detach(), rot.grad, xyz.grad, drr.density.grad.clone())
    a, b = res[True], res[False]
    assert a[0].shape == (2, 5 if mask else 1, 14, 11)
    assert rel_err(a[0].numpy(), b[0].numpy()) < 1e-5
    assert rel_err(a[1].numpy(), b[1].numpy()) < 2e-3
    assert rel_err(a[2].numpy(), b[2].numpy()) < 2e-3
    assert rel_err(a[3].numpy(), b[3].numpy()) < 1e-4


def test_alpha_range_kernel_equals_the_tensor_ops(emulated_ops):
    """ddrr_trilinear_alpha_range (one pass, taken when nothing differentiates through the
    marching range) against get_alpha_minmax + min / max (reference renderers.py:124-140,
    220-223): bit-equal -- same IEEE operations -- on oblique rays, per-ray sources, rays
    parallel to an axis, rays that miss the volume; and Trilinear.forward renders the same
    image either way (fixture of the unmodified reference)."""
    from diffdrr_amd import Trilinear
    from diffdrr_amd.renderers import get_alpha_minmax

    g = torch.Generator().manual_seed(0)
    dims = (12, 10, 14)
    for trial in range(6):
        B, N = 3, 50
        src = torch.rand(B, 1 if trial % 2 else N, 3, generator=g) * 60 - 30
        tgt = torch.rand(B, N, 3, generator=g) * 40 - 10
        if trial == 2:
            tgt[:, :7, 0] = src[:, :1, 0] if src.shape[1] == 1 else src[:, :7, 0]   # d_x = eps
        if trial == 3:
            tgt = tgt + 500.0                                                       # all miss
        lo, hi = get_alpha_minmax(src, tgt, torch.tensor(dims).float(), 0.5, 1e-8)
        a0, a1 = emulated_ops.trilinear_alpha_range(src, tgt, dims)
        assert a0.item() == lo.min().item() and a1.item() == hi.max().item(), trial
    gold = golden("trilinear_global_range")
    f32 = lambda k: torch.from_numpy(np.ascontiguousarray(gold[k], dtype=np.float32))  # noqa: E731
    vol, s, t, img = (f32(k) for k in ("volume", "source", "target", "img_f32"))
    calls = []
    real = emulated_ops.trilinear_alpha_range
    emulated_ops.trilinear_alpha_range = lambda *a, **k: calls.append(1) or real(*a, **k)
    try:
        with torch.no_grad():
            out = Trilinear()(vol, s, t, img, n_points=41)
        assert calls == [1]
        out_g = Trilinear()(vol, s, t.clone().requires_grad_(), img, n_points=41)  # tensor ops
        assert calls == [1]
    finally:
        emulated_ops.trilinear_alpha_range = real
    assert rel_err(out.numpy(), gold["out_f32"]) < 1e-4
    assert torch.equal(out, out_g.detach())


def test_fused_ncc_equals_pytorch_formula(emulated_ops):
    """The fused NCC kernels (ddrr_ncc_forward / _backward, reference metrics.py:21-44)
    against the module's PyTorch formula: values and gradients, paired and with a fixed
    image shared by the batch (expand)."""
    from diffdrr_amd import NormalizedCrossCorrelation2d

    g = torch.Generator().manual_seed(0)
    ncc = NormalizedCrossCorrelation2d()
    x1 = 300 + 40 * torch.rand(5, 1, 18, 22, generator=g)
    x2 = 280 + 55 * torch.rand(5, 1, 18, 22, generator=g)
    w = torch.rand(5, generator=g)

    def formula(a, b):
        return ((ncc.norm(a) * ncc.norm(b)).flatten(1).sum(1)) / (18 * 22)

    for shared in (False, True):
        a_in = (x1[:1].expand(5, -1, -1, -1) if shared else x1)
        a1, b1 = a_in.clone().requires_grad_(), x2.clone().requires_grad_()
        a2, b2 = a_in.clone().requires_grad_(), x2.clone().requires_grad_()
        fused = ncc(a1.expand_as(b1) if False else a1, b1)
        ref = formula(a2, b2)
        assert torch.allclose(fused, ref, atol=2e-6)
        (fused * w).sum().backward()
        (ref * w).sum().backward()
        assert rel_err(b1.grad.numpy(), b2.grad.numpy()) < 1e-4
        assert rel_err(a1.grad.numpy(), a2.grad.numpy()) < 1e-4
    # the stride-0 fixed image (what Registration / sweeps pass) takes the shared route
    fixed = x1[:1]
    b3 = x2.clone().requires_grad_()
    v = ncc(fixed.expand(5, -1, -1, -1), b3)
    assert torch.allclose(v, formula(fixed.expand(5, -1, -1, -1), x2), atol=2e-6)


@pytest.mark.parametrize("convention", ["ZXY", "XYZ", "YZX", "ZYX", "XZX", "YXY"])
def test_fused_euler_pose_equals_convert_chain(emulated_ops, convention):
    """ddrr_pose_euler_forward / _backward (one kernel each way) against
    reorient.compose(convert(rot, xyz, 'euler_angles', convention)) in PyTorch: the 3x4
    world matrix and its gradients w.r.t. angles and translation, incl. degrees."""
    from diffdrr_amd.data import reorient_matrix
    from diffdrr_amd.pose import RigidTransform, convert, euler_world_pose

    g = torch.Generator().manual_seed(3)
    reorient = reorient_matrix("AP")
    w = torch.rand(5, 3, 4, generator=g)
    for degrees in (False, True):
        rot0 = (torch.rand(5, 3, generator=g) - 0.5) * (300.0 if degrees else 5.0)
        xyz0 = (torch.rand(5, 3, generator=g) - 0.5) * 800
        r1, x1 = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
        r2, x2 = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
        fused = euler_world_pose(r1, x1, convention, reorient, degrees=degrees)
        pose = convert(r2, x2, parameterization="euler_angles", convention=convention,
                       degrees=degrees)
        ref = RigidTransform(reorient).compose(pose).matrix[:, :3, :]
        assert torch.allclose(fused, ref, rtol=1e-5, atol=2e-4)
        (fused * w).sum().backward()
        (ref * w).sum().backward()
        assert rel_err(r1.grad.numpy(), r2.grad.numpy()) < 1e-4
        assert rel_err(x1.grad.numpy(), x2.grad.numpy()) < 1e-5


@pytest.mark.parametrize("kind", ["siddon", "trilinear"])
def test_mask_to_channels_gradients(emulated_ops, kind):
    """mask_to_channels is differentiable like the reference's scatter_add (renderers.py:77-89,
    242-252): gradients w.r.t. ray endpoints, img and the volume for a (B, C, N) grad_out
    against the reference's autograd (fixtures generated from the unmodified reference)."""
    from diffdrr_amd import Siddon, Trilinear

    g = golden(kind + "_mask")
    f32 = lambda k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32))  # noqa: E731
    vol, src, tgt, img = (f32(k).requires_grad_() for k in ("volume", "source", "target", "img_f32"))
    mask = torch.from_numpy(g["mask"])
    if kind == "siddon":
        out = Siddon()(vol, src, tgt, img, mask=mask)
    else:
        out = Trilinear()(vol, src, tgt, img, n_points=40, mask=mask)
    assert rel_err(out.detach().numpy(), g["out_f32"]) < 1e-4
    grads = torch.autograd.grad(out, [src, tgt, img, vol], f32("grad_out_f32"))
    for name, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        # fp32 fixture: in fp64 the label of a sample lying exactly on a voxel face may differ
        assert rel_err(gr.numpy(), g[name + "_f32"]) < 1e-3, name
    if kind == "siddon":
        # stop_gradients_through_grid_sample (renderers.py:63-65): only the alpha path is left
        vol2, src2, tgt2, img2 = (t.detach().clone().requires_grad_() for t in (vol, src, tgt, img))
        out2 = Siddon(stop_gradients_through_grid_sample=True)(vol2, src2, tgt2, img2, mask=mask)
        gs2, gt2, gi2, gv2 = torch.autograd.grad(out2, [src2, tgt2, img2, vol2], f32("grad_out_f32"),
                                                 allow_unused=True)
        assert torch.equal(gt2, grads[1]) and torch.equal(gs2, grads[0])
        assert gi2 is None and gv2 is None


def test_drr_mask_to_channels_pose_gradients(emulated_ops):
    """Pose gradients through `mask_to_channels=True`: a loss that weights every channel
    equally has the gradient of the plain DRR; a loss on a subset of structures is what the
    mask is for (reference introduction.ipynb:230-286)."""
    g = golden("drr_module")
    drr = DRR(_subject_a(g), **_geo(g))
    w = torch.rand(g["siddon_img_f32"].shape, generator=torch.Generator().manual_seed(5))
    grads = {}
    for name in ("channels", "plain", "subset"):
        rot, xyz = T(g["rot"]).requires_grad_(), T(g["xyz"]).requires_grad_()
        kw = dict(parameterization="euler_angles", convention="ZXY")
        if name == "plain":
            img = drr(rot, xyz, **kw)
        else:
            ch = drr(rot, xyz, mask_to_channels=True, **kw)
            img = ch.sum(1, keepdim=True) if name == "channels" else ch[:, 1:3].sum(1, keepdim=True)
        (img * w).sum().backward()
        grads[name] = (rot.grad.clone(), xyz.grad.clone())
    for a, b in zip(grads["channels"], grads["plain"]):
        assert rel_err(a.numpy(), b.numpy()) < 1e-4
    assert rel_err(grads["subset"][0].numpy(), grads["plain"][0].numpy()) > 1e-2


def test_siddon_callable_reducefn(emulated_ops):
    """A callable ``reducefn`` over the materialised per-segment tensor (reference
    renderers.py:175-183, introduction.ipynb:506-529), values and gradients against the
    reference's autograd (fixture from the unmodified reference, top-6 sum)."""
    from diffdrr_amd import Siddon

    def topk_sum(img):
        return img.sort(descending=True).values[..., :6].sum(dim=-1)

    g = golden("siddon_callable")
    f32 = lambda k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32))  # noqa: E731
    vol, src, tgt, img = (f32(k).requires_grad_() for k in ("volume", "source", "target", "img_f32"))
    out = Siddon(reducefn=topk_sum)(vol, src, tgt, img)
    assert out.shape == g["out_f32"].shape
    assert rel_err(out.detach().numpy(), g["out_f32"]) < 1e-4
    grads = torch.autograd.grad(out, [src, tgt, img, vol], f32("grad_out_f32"))
    for name, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.numpy(), g[name + "_f64"]) < 1e-3, name
    # the tensor itself: same shape as the reference's, and summing it is the plain render
    terms = Siddon(reducefn=lambda t: t)(vol, src, tgt, img).squeeze(1)
    B, N, _ = tgt.shape
    assert terms.shape == (B, N, sum(vol.shape) + 2)
    plain = Siddon()(vol, src, tgt, img)
    assert rel_err(terms.sum(-1).detach().numpy(), plain.squeeze(1).detach().numpy()) < 1e-5


def test_trilinear_callable_reducefn(emulated_ops):
    """A callable ``reducefn`` of the marcher over the materialised per-sample tensor
    (reference renderers.py:226-240), values and gradients against the reference's autograd."""
    from diffdrr_amd import Trilinear

    def topk_sum(img):
        return img.sort(descending=True).values[..., :6].sum(dim=-1)

    g = golden("trilinear_callable")
    f32 = lambda k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32))  # noqa: E731
    vol, src, tgt, img = (f32(k).requires_grad_() for k in ("volume", "source", "target", "img_f32"))
    out = Trilinear(reducefn=topk_sum)(vol, src, tgt, img, n_points=40)
    assert out.shape == g["out_f32"].shape
    assert rel_err(out.detach().numpy(), g["out_f32"]) < 1e-4
    grads = torch.autograd.grad(out, [src, tgt, img, vol], f32("grad_out_f32"))
    for name, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.numpy(), g[name + "_f64"]) < 1e-3, name
    samples = Trilinear(reducefn=lambda t: t)(vol, src, tgt, img, n_points=40).squeeze(1)
    assert samples.shape == (tgt.shape[0], tgt.shape[1], 40)
    plain = Trilinear()(vol, src, tgt, img, n_points=40)
    assert rel_err(samples.sum(-1).detach().numpy(), plain.squeeze(1).detach().numpy()) < 1e-5


@pytest.mark.parametrize("name,ctor,call", [
    ("siddon_bilinear", {"mode": "bilinear"}, {}),
    ("siddon_align_corners", {}, {"align_corners": True}),
])
def test_siddon_midpoint_lookup_gradients(emulated_ops, name, ctor, call):
    """Siddon(mode="bilinear") and align_corners=True are differentiable like the reference
    (midpoint lookups, renderers.py:57-71, incl. the path through the midpoint positions):
    gradients against the reference's autograd."""
    from diffdrr_amd import Siddon

    g = golden(name)
    f32 = lambda k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32))  # noqa: E731
    vol, src, tgt, img = (f32(k).requires_grad_() for k in ("volume", "source", "target", "img_f32"))
    out = Siddon(**ctor)(vol, src, tgt, img, **call)
    assert rel_err(out.detach().numpy(), g["out_f32"]) < 1e-4
    grads = torch.autograd.grad(out, [src, tgt, img, vol], f32("grad_out_f32"))
    for k, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.numpy(), g[k + "_f64"]) < 1e-3, k


def test_trilinear_max_gradients(emulated_ops):
    """Trilinear(reducefn="max") is differentiable like the reference (the arg-max sample gets
    the gradient): against the reference's autograd."""
    from diffdrr_amd import Trilinear

    g = golden("trilinear_max")
    f32 = lambda k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32))  # noqa: E731
    vol, src, tgt, img = (f32(k).requires_grad_() for k in ("volume", "source", "target", "img_f32"))
    out = Trilinear(reducefn="max")(vol, src, tgt, img, n_points=37)
    assert rel_err(out.detach().numpy(), g["out_f32"]) < 1e-4
    grads = torch.autograd.grad(out, [src, tgt, img, vol], f32("grad_out_f32"))
    for k, gr in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        assert rel_err(gr.numpy(), g[k + "_f64"]) < 1e-3, k


def test_packed_record_gradients(emulated_ops):
    """Opt-in fixed-point backward record of the brick kernel (csrc/record_pack.h): same pose
    gradients as the fp32 record up to its resolution, also with the source inside the volume
    (alpha bound > 1), and the C-ABI consumers decode it."""
    g = golden("drr_module")
    res = {}
    for packed in (False, True):
        drr = DRR(_subject_a(g), **_geo(g))
        drr.renderer.packed_record = packed
        rot = torch.cat([T(g["rot"]), torch.tensor([[0.2, 0.1, 0.0]])]).requires_grad_()
        xyz = torch.cat([T(g["xyz"]), torch.tensor([[3.0, 20.0, -4.0]])]).requires_grad_()
        img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
        w = torch.rand(img.shape, generator=torch.Generator().manual_seed(3))
        (img * w).sum().backward()
        res[packed] = (img.detach(), rot.grad.clone(), xyz.grad.clone())
    assert rel_err(res[True][0].numpy(), res[False][0].numpy()) < 1e-6
    assert rel_err(res[True][1].numpy(), res[False][1].numpy()) < 1e-4
    assert rel_err(res[True][2].numpy(), res[False][2].numpy()) < 1e-4


def test_record_pack_roundtrip():
    """Two signed 32-bit fixed-point fields in one 64-bit sum: adding packed values equals
    adding the fields, for either sign and through borrows."""
    rng = np.random.default_rng(0)
    lo = rng.integers(-2**20, 2**20, size=(1000, 19))
    hi = rng.integers(-2**20, 2**20, size=(1000, 19))
    packed = (hi.astype(np.int64) * 2**32 + lo.astype(np.int64)).sum(1)
    lo_sum = ((packed & 0xFFFFFFFF).astype(np.uint32)).astype(np.int32).astype(np.int64)
    hi_sum = (packed - lo_sum) // 2**32
    assert np.array_equal(lo_sum, lo.sum(1)) and np.array_equal(hi_sum, hi.sum(1))


# ------------------------------------------------ regressions from the round-1 review

def _small_drr(renderer="siddon", **kw):
    from diffdrr_amd.data import synthetic_subject

    return DRR(synthetic_subject(40, kind="noise", seed=0, n_labels=5), sdd=400.0, height=22,
               width=30, delx=1.8, renderer=renderer, **kw)


def test_render_with_permuted_rays_is_not_taken_for_a_detector_grid(emulated_ops):
    """`DRR.render` is public (reconstruction.ipynb:122 calls it): rays that are NOT the
    row-major detector grid must not reach the volume-stationary kernels, which cull rays with
    an affine model of that grid -- permuted rays render to the same values, permuted."""
    drr = _small_drr()
    rot = torch.tensor([[0.3, -0.2, 0.4]])
    xyz = torch.tensor([[5.0, 250.0, -3.0]])
    pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
    source, target = drr.detector(pose, None)
    with torch.no_grad():
        ref = drr.render(drr.density, source, target)
        assert drr.renderer.detector_shape == (22, 30)      # a true grid: checked, accepted
        perm = torch.randperm(target.shape[1], generator=torch.Generator().manual_seed(0))
        out = drr.render(drr.density, source, target[:, perm])
        assert drr.renderer.detector_shape is None            # not a grid: per-ray kernels
    assert rel_err(out.numpy(), ref[..., perm].numpy()) < 2e-5
    assert not emulated_ops.rays_form_detector_grid(source, target[:, perm], 22, 30)
    assert emulated_ops.rays_form_detector_grid(source, target, 22, 30)


@pytest.mark.parametrize("renderer", ["siddon", "trilinear"])
def test_swapped_in_renderer_checks_the_detector_shape_it_is_given(emulated_ops, renderer):
    """A renderer whose `detector_shape` was set by someone else than `diffdrr_amd.DRR` (the
    swap into the reference's DRR) checks the promise per call: rays that are
    not that grid go to the per-ray kernels instead of being culled by the grid model."""
    import diffdrr_amd

    drr = _small_drr(renderer)
    pose = convert(torch.tensor([[0.3, -0.2, 0.4]]), torch.tensor([[5.0, 250.0, -3.0]]),
                   parameterization="euler_angles", convention="ZXY")
    source, target = drr.detector(pose, None)
    img = (target - source).norm(dim=-1).unsqueeze(1)
    s, t = drr.affine_inverse(source), drr.affine_inverse(target)
    mod = diffdrr_amd.Siddon() if renderer == "siddon" else diffdrr_amd.Trilinear()
    kw = {} if renderer == "siddon" else {"n_points": 60}
    calls = []
    real = emulated_ops.rays_form_detector_grid
    emulated_ops.rays_form_detector_grid = lambda *a, **k: calls.append(1) or real(*a, **k)
    try:
        with torch.no_grad():
            plain = mod(drr.density, s, t, img, **kw)
            assert not calls                                  # no shape promised: nothing to check
            mod.detector_shape = (22, 30)
            grid = mod(drr.density, s, t, img, **kw)
            assert len(calls) == 1
            perm = torch.randperm(t.shape[1], generator=torch.Generator().manual_seed(1))
            out = mod(drr.density, s, t[:, perm], img[..., perm], **kw)
            assert len(calls) == 2
            mod.trust_detector_shape = True                   # what diffdrr_amd.DRR does
            mod(drr.density, s, t, img, **kw)
            assert len(calls) == 2
    finally:
        emulated_ops.rays_form_detector_grid = real
    assert rel_err(grid.numpy(), plain.numpy()) < 3e-5
    assert rel_err(out.numpy(), plain[..., perm].numpy()) < 3e-5


@pytest.mark.parametrize("renderer", ["siddon", "trilinear"])
def test_detector_grid_trust_ends_with_the_drr_call(emulated_ops, renderer):
    """`DRR` vouches for the rays IT generated, for that call only: a later direct
    `drr.renderer(...)` call with permuted rays (N == H * W, not the grid; the trilinear
    tutorial calls the renderer directly) is checked again and rendered per ray, not culled
    with the grid model of the previous call."""
    drr = _small_drr(renderer)
    rot = torch.tensor([[0.3, -0.2, 0.4]])
    xyz = torch.tensor([[5.0, 250.0, -3.0]])
    kw = {} if renderer == "siddon" else {"n_points": 60}
    with torch.no_grad():
        ref = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", **kw)
        assert drr.renderer.trust_detector_shape is False
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        drr.render(drr.density, source, target, **kw)
        assert drr.renderer.trust_detector_shape is False
        assert drr.renderer.detector_shape == (22, 30)       # (left behind by the call above)
        img = (target - source).norm(dim=-1).unsqueeze(1)
        s, t = drr.affine_inverse(source), drr.affine_inverse(target)
        perm = torch.randperm(t.shape[1], generator=torch.Generator().manual_seed(2))
        out = drr.renderer(drr.density, s, t[:, perm], img[..., perm], **kw)
    assert rel_err(out.numpy(), ref.reshape(1, 1, -1)[..., perm].numpy()) < 3e-5
    # degenerate "grids" are not grids, an empty batch is not checked at all
    assert not emulated_ops.rays_form_detector_grid(s, t[:, :1].expand(-1, 660, -1).contiguous(), 22, 30)
    assert not emulated_ops.rays_form_detector_grid(s[:0], t[:0], 22, 30)


@pytest.mark.parametrize("renderer", ["siddon", "trilinear"])
def test_reference_rays_through_swapped_renderers_on_the_host(emulated_ops, renderer):
    """tests/golden/reference_drr_rays.npz through the product's modules on the host emulation
    (the GPU twin: tests/test_gpu_parity.py::test_reference_rays_through_swapped_renderers)."""
    from conftest import check_reference_rays_through_swapped_renderer

    check_reference_rays_through_swapped_renderer(renderer, torch.device("cpu"), emulated_ops)


def test_mask_label_cache_is_tied_to_the_mask_object(emulated_ops):
    """A new mask that lands at a freed mask's address must not be served the old labels."""
    from diffdrr_amd.renderers import _labels_u8

    seen = []
    for trial in range(20):
        n = 3 + trial % 5
        mask = torch.randint(0, n, (6, 6, 6), generator=torch.Generator().manual_seed(trial)).float()
        mask[0, 0, 0] = n - 1
        (lab, C, _), = _labels_u8(mask)
        assert C == n and torch.equal(lab.float(), mask)
        seen.append(C)
        del mask, lab
    a = torch.zeros(4, 4, 4)
    (lab0, C0, _), = _labels_u8(a)
    a[0, 0, 0] = 2                      # in-place change: the version moves
    (lab1, C1, _), = _labels_u8(a)
    assert (C0, C1) == (1, 3)


@pytest.mark.parametrize("renderer", ["siddon", "trilinear"])
def test_more_than_256_labels_render_in_chunks(emulated_ops, renderer):
    """The reference takes any ``mask.max()`` (renderers.py:81); the kernels take one byte per
    label, so a map with 300 labels is rendered 255 labels at a time: same channels as the
    plain render split by label, and differentiable."""
    from diffdrr_amd import Siddon, Trilinear

    g = torch.Generator().manual_seed(5)
    vol = torch.rand(12, 10, 14, generator=g)
    mask = torch.randint(0, 300, vol.shape, generator=g).float()
    mask[0, 0, 0] = 299
    src = torch.tensor([[[-30.0, 4.0, 6.0]]])
    tgt = torch.stack([torch.full((40,), 45.0), torch.linspace(1, 9, 40), torch.linspace(2, 12, 40)], -1)[None]
    img = (tgt - src).norm(dim=-1).unsqueeze(1)
    mod = (Siddon if renderer == "siddon" else Trilinear)()
    kw = {} if renderer == "siddon" else {"n_points": 50}
    vol_g = vol.clone().requires_grad_()
    out = mod(vol_g, src, tgt, img, mask=mask, **kw)
    assert out.shape == (1, 300, 40)
    plain = mod(vol, src, tgt, img, **kw)
    assert rel_err(out.sum(1, keepdim=True).detach().numpy(), plain.numpy()) < 1e-5
    # one channel against the plain render of the volume restricted to that label
    c = int(mask[6, 5, 7].item())
    if renderer == "siddon":
        only = mod(vol * (mask == c), src, tgt, img, **kw)
        assert rel_err(out[:, c:c + 1].detach().numpy(), only.numpy()) < 1e-5
    out.sum().backward()
    assert torch.isfinite(vol_g.grad).all() and vol_g.grad.abs().sum() > 0


@pytest.mark.parametrize("renderer", ["siddon", "trilinear"])
def test_non_contiguous_volume_views(emulated_ops, renderer):
    """A permuted view of a volume renders like its contiguous copy (the kernels take raw
    pointers: every entry point makes its inputs contiguous)."""
    from diffdrr_amd import Siddon, Trilinear

    drr = _small_drr(renderer)
    pose = convert(torch.tensor([[0.2, 0.1, -0.3]]), torch.tensor([[2.0, 240.0, 4.0]]),
                   parameterization="euler_angles", convention="ZXY")
    source, target = drr.detector(pose, None)
    img = (target - source).norm(dim=-1).unsqueeze(1)
    s, t = drr.affine_inverse(source), drr.affine_inverse(target)
    vol = drr.density
    view = vol.permute(2, 1, 0).contiguous().permute(2, 1, 0)   # same values, other strides
    assert not view.is_contiguous() and torch.equal(view, vol)
    mod = Siddon() if renderer == "siddon" else Trilinear()
    kw = {} if renderer == "siddon" else {"n_points": 40}
    with torch.no_grad():
        assert rel_err(mod(view, s, t, img, **kw).numpy(), mod(vol, s, t, img, **kw).numpy()) < 1e-6
    v = view.clone().requires_grad_()
    mod(v.permute(0, 1, 2), s, t, img, **kw).sum().backward()
    assert torch.isfinite(v.grad).all() and v.grad.abs().max() > 0


def test_calibration_that_requires_grad_takes_the_general_path(emulated_ops):
    """Gradients w.r.t. a user calibration flow through Detector.forward (reference
    detector.py:147-150): the fused pose entry must step aside for them."""
    from diffdrr_amd import RigidTransform

    drr = _small_drr()
    rot = torch.tensor([[0.1, 0.2, -0.1]])
    xyz = torch.tensor([[1.0, 250.0, 2.0]])
    M = drr.detector.calibration.matrix.detach().clone().requires_grad_()
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY",
              calibration=RigidTransform(M))
    img.sum().backward()
    assert M.grad is not None and M.grad.abs().max() > 0
    with torch.no_grad():
        plain = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    assert rel_err(img.detach().numpy(), plain.numpy()) < 2e-5


def test_hu_to_density_matches_reference():
    """Ingest: HU -> [0, 1] density (reference data.py:214-227), bit for bit, on the fixture
    made from the reference's own source (tests/golden/make_golden_ingest.py)."""
    from diffdrr_amd.data import transform_hu_to_density

    g = golden("hu_to_density")
    vol = T(g["volume"])
    for m in (1.0, 2.5):
        assert np.array_equal(transform_hu_to_density(vol, m).numpy(), g[f"density_{m}"])


# ------------------------------------------------------------------ double precision

@pytest.mark.parametrize("name,kw", [("siddon_sum", {}), ("siddon_sum_oblique", {}),
                                     ("siddon_per_ray_source", {}), ("siddon_max", {"reducefn": "max"})])
def test_siddon_float64_matches_reference_fp64(emulated_ops, name, kw):
    """A float64 volume renders through the fp64 kernels (reference: the module `.to(float64)`,
    drr.py:71-75): the reference's own fp64 outputs and autograd gradients to ~1e-12."""
    from diffdrr_amd import Siddon

    g = golden(name)
    vol, src, tgt = (T(g[k].astype(np.float64)).requires_grad_() for k in ("volume", "source", "target"))
    B, N, _ = tgt.shape
    img = T(g["img_f64"].reshape(B, 1, N)).requires_grad_()
    out = Siddon(**kw)(vol, src, tgt, img)
    assert out.dtype == torch.float64 and out.shape == g["out_f64"].shape
    assert rel_err(out.detach().numpy(), g["out_f64"]) < 1e-12
    if kw:
        return  # (max: forward only in fp64)
    gs, gt, gi, gv = torch.autograd.grad(out, (src, tgt, img, vol), T(g["grad_out_f64"]))
    for mine, key in ((gs, "g_source_f64"), (gt, "g_target_f64"), (gi, "g_img_f64"), (gv, "g_volume_f64")):
        assert rel_err(mine.numpy(), g[key]) < 1e-10, key


@pytest.mark.parametrize("name,npts,rng", [("trilinear_global_range", 41, None),
                                           ("trilinear_explicit_range", 64, (0.31, 0.77)),
                                           ("trilinear_oblique", 50, None)])
def test_trilinear_float64_matches_reference_fp64(emulated_ops, name, npts, rng):
    from diffdrr_amd import Trilinear

    g = golden(name)
    vol, src, tgt = (T(g[k].astype(np.float64)).requires_grad_() for k in ("volume", "source", "target"))
    B, N, _ = tgt.shape
    img = T(g["img_f64"].reshape(B, 1, N)).requires_grad_()
    kw = {} if rng is None else {"alphamin": rng[0], "alphamax": rng[1]}
    out = Trilinear()(vol, src, tgt, img, n_points=npts, **kw)
    assert out.dtype == torch.float64
    # (the reference's sample fractions are an fp32 torch.linspace table cast to fp64,
    # renderers.py:224; aten's vectorised kernel rounds a few entries one fp32 ulp away from
    # the scalar formula the kernels restate: ~1e-8 here, not 1e-15)
    assert rel_err(out.detach().numpy(), g["out_f64"]) < 1e-6
    gs, gt, gi, gv = torch.autograd.grad(out, (src, tgt, img, vol), T(g["grad_out_f64"]))
    for mine, key in ((gs, "g_source_f64"), (gt, "g_target_f64"), (gi, "g_img_f64"), (gv, "g_volume_f64")):
        assert rel_err(mine.numpy(), g[key]) < 1e-5, key


def test_drr_module_in_float64(emulated_ops):
    """`DRR(...).to(torch.float64)` end to end, like the reference module: fp64 image, gradients
    w.r.t. the pose parameters; agrees with the fp32 module to fp32 accuracy."""
    drr32 = _small_drr()
    drr64 = _small_drr().to(torch.float64)
    assert drr64.density.dtype == torch.float64
    rot = torch.tensor([[0.2, -0.1, 0.3]])
    xyz = torch.tensor([[4.0, 250.0, -2.0]])
    r64 = rot.double().requires_grad_()
    x64 = xyz.double().requires_grad_()
    img64 = drr64(r64, x64, parameterization="euler_angles", convention="ZXY")
    assert img64.dtype == torch.float64 and img64.shape == (1, 1, 22, 30)
    img64.sum().backward()
    assert torch.isfinite(r64.grad).all() and r64.grad.abs().max() > 0
    with torch.no_grad():
        img32 = drr32(rot, xyz, parameterization="euler_angles", convention="ZXY")
    assert rel_err(img32.numpy(), img64.detach().numpy()) < 1e-4


# ------------------------------------------------------------------ the general path

@pytest.mark.parametrize("name,tag", conftest.general_case_ids())
def test_general_path_matches_the_reference(emulated_ops, name, tag):
    """Every keyword combination outside the fused kernels (csrc/general_core.h: masks,
    callables, max, stop-gradients with the midpoint lookups; all of float64) against fixtures
    of the unmodified reference: outputs and autograd gradients."""
    conftest.check_general_case(name, tag, "cpu")


def test_brick_workspace_follows_the_volume(emulated_ops):
    """The per-volume workspace of the 16-bit bricks (ranges + packed copy) is rebuilt when the
    volume is edited in place -- into the same buffer -- and a volume that keeps changing goes
    back to fp32 bricks instead of paying the rebuild on every render."""
    from diffdrr_amd import ops
    from diffdrr_amd.renderers import _brick_storage

    vol = torch.rand(64, 64, 128)
    buf, valid = ops.brick_workspace(vol, "q16p")
    # handing the buffer out does not make it valid: only the launch that filled it does
    assert valid == 0 and ops.brick_workspace(vol, "q16p")[1] == 0
    ops.brick_workspace_commit(vol, "q16p")
    assert ops.brick_workspace(vol, "q16p")[1] == 1
    vol[0, 0, 0] = 2.0
    buf2, valid2 = ops.brick_workspace(vol, "q16p")
    assert valid2 == 0 and buf2.data_ptr() == buf.data_ptr()
    ops.brick_workspace_commit(vol, "q16p")
    assert ops.workspace_churn(vol, "q16p") == 1 and ops.workspace_churn(vol, "q16") == 0
    cfg = {"storage": "q16p"}
    assert _brick_storage(vol, cfg) == "q16p"
    for _ in range(2):
        vol[0, 0, 0] += 1.0
        ops.brick_workspace(vol, "q16p")
        ops.brick_workspace_commit(vol, "q16p")
    assert _brick_storage(vol, cfg) == "f32"
    # a temporary made contiguous per call never gets a workspace; any shape does (the
    # reference's example CT has 133 slices)
    assert _brick_storage(torch.rand(64, 64, 130)[:, :, :128], cfg) == "f32"
    assert _brick_storage(torch.rand(64, 64, 126), cfg) == "q16p"
    assert _brick_storage(torch.rand(64, 64, 133), cfg) == "q16p"
    assert _brick_storage(torch.rand(64, 64, 128, requires_grad=True), cfg) == "f32"
    # volumes with few double bricks per CU: the measured table (profiles/r06/storage_table.txt)
    from diffdrr_amd.renderers import _few_bricks_take_q16 as take
    assert [take(1.0, b) for b in (1, 5, 6, 8, 32)] == [True, True, False, False, False]        # 256^3
    assert all(take(2.25, b) for b in (1, 8, 32, 512))                                          # 384 x 384 x 256
    assert [take(3.0, b) for b in (1, 7, 8, 12, 13, 32)] == [False, False, True, True, False, False]  # the example CT's shape
    assert not take(3.0, None) and not take(1.0, None)


def test_trilinear_channels_on_bricks_on_the_host(emulated_ops):
    """ddrr_trilinear_forward_channels_bricks (host emulation of the same march) against the
    per-ray channel kernel, the plain march and through the module."""
    conftest.check_trilinear_channels_on_bricks("cpu", (40, 36, 45), (14, 11), 60)


@pytest.mark.parametrize("dims", [(70, 50, 133), (33, 34, 5), (20, 24, 3), (40, 36, 1), (3, 50, 1), (2, 3, 37)])
def test_any_depth_on_every_brick_storage_on_the_host(emulated_ops, dims):
    """The host twin of tests/test_gpu_brick_storage.py::test_any_depth_on_the_configurable_kernel:
    the Python layer hands volumes of any D.z to the 16-bit storages (ops.brick_storage_applies)
    and the emulation of the same entry points agrees with the per-ray walk."""
    import test_gpu_brick_storage as G

    G.test_any_depth_on_the_configurable_kernel("cpu", dims)


def test_channel_render_with_an_odd_label_address_on_the_host(emulated_ops):
    import test_gpu_brick_storage as G

    G.test_channel_render_stages_any_depth_and_label_alignment("cpu", (40, 36, 45), 5)


@pytest.mark.parametrize("dims", [(1, 1, 7), (1, 2, 5), (2, 2, 4), (3, 2, 6), (2, 3, 9), (1, 1, 4),
                                  (4, 4, 1), (2, 2, 1), (1, 4, 1), (5, 1, 1), (3, 3, 2), (2, 3, 3)])
def test_quads_are_staged_from_the_right_voxels_on_any_shape(emu_lib, dims):
    """The arithmetic of brick_shared.h quad_load / quad_fix (brick_core.h quad_clamped_at /
    quad_shift, compiled here for the host): a quad is one 16-byte load clamped to the volume's last
    four voxels and shifted where it is used -- right for every D.z >= 4.  With fewer slices quads of
    rows BEFORE the volume's last row are clamped too ((4, 4, 1) staged two voxels from
    the wrong address): `quads_serve` says so, and the launchers send such volumes to the general
    kernel's scalar staging (tests/test_gpu_brick_storage.py::test_any_depth_...: (40, 36, 1) ...)."""
    import ctypes

    fn = emu_lib.cdll.ddrr_emu_quad_stage
    fn.restype = ctypes.c_int
    dx, dy, dz = dims
    vol = np.arange(1, dx * dy * dz + 1, dtype=np.float32).reshape(dims)
    out = np.zeros(4, dtype=np.float32)
    args = lambda x, y, z: (vol.ctypes.data_as(ctypes.c_void_p), dx, dy, dz, x, y, z,  # noqa: E731
                            out.ctypes.data_as(ctypes.c_void_p))
    if dz < 4:
        assert fn(*args(0, 0, 0)) == -1  # refused: not this path's volume
        return
    for x in range(dx):
        for y in range(dy):
            for z in range(0, dz, 4):
                assert fn(*args(x, y, z)) == 0
                for i in range(4):
                    if z + i < dz:  # (what lies behind the row's end is the caller's to mask)
                        assert out[i] == vol[x, y, z + i], (dims, x, y, z, i)


def test_filter_intersections_outside_volume_on_the_host(emulated_ops):
    """Row a5 on the host emulation (the GPU twin: tests/test_gpu_parity.py)."""
    conftest.check_filter_intersections_outside_volume("cpu")


def test_channel_backward_on_bricks_smooth_volume_on_the_host(emulated_ops):
    conftest.check_channel_backward_on_bricks_smooth(emulated_ops, "cpu")


def test_fused_ncc_step_on_the_host(emulated_ops):
    """DRR.ncc through the host build of the entries it fuses (tests/emu): the Python / autograd
    wiring; the GPU twin checks the kernels (tests/test_gpu_parity.py)."""
    conftest.check_fused_ncc_step("cpu")


def test_euler_inference_path_on_the_host_twin(emulated_ops):
    conftest.check_euler_inference_path("cpu")


def test_euler_differentiable_path_on_the_host_twin(emulated_ops, monkeypatch):
    """(the Python / autograd wiring and the C-ABI entries taken; the device twin: tests/test_gpu_parity.py)"""
    calls, launch = [], emulated_ops._launch
    monkeypatch.setattr(emulated_ops, "_launch", lambda n, d, *a: (calls.append(n), launch(n, d, *a))[1])
    conftest.check_euler_differentiable_path("cpu", emulated_ops, calls)


def test_pose_adam_matches_torch_adam_on_the_host_twin(emulated_ops):
    conftest.check_pose_adam("cpu")


@pytest.mark.parametrize("name", sorted(conftest.SPARSE_CASES))
def test_subsample_and_patches_match_reference_on_the_bricks(emulated_ops, monkeypatch, name):
    """The reference's own speed levers through the host build of the brick
    entries: fixture of the unmodified reference, and no per-ray forward kernel in the call list."""
    calls = []
    launch = emulated_ops._launch
    monkeypatch.setattr(emulated_ops, "_launch", lambda n, d, *a: (calls.append(n), launch(n, d, *a))[1])
    conftest.check_sparse_lever(name, "cpu", emulated_ops, calls)


def test_patch_ncc_kernels_against_the_composition(emulated_ops):
    """(host build of ncc_patch_core.h; the device twin: tests/test_gpu_parity.py)"""
    conftest.check_patch_ncc_against_composition(torch.device("cpu"))


def test_blur_sobel_kernels_against_the_composition(emulated_ops):
    """(host build of blur_core.h; the device twin: tests/test_gpu_parity.py)"""
    conftest.check_blur_sobel_against_composition(torch.device("cpu"))


def test_channel_render_from_ready_packed_words_on_the_host(emulated_ops):
    conftest.check_channel_words("cpu", emulated_ops)
