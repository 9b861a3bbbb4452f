"""Shared test plumbing.

Markers: ``gpu`` = needs a real MI355X (run by the driver with ``-m gpu``);
everything else runs in the GPU-less build container.

Helpers here are TEST infrastructure: the golden-fixture loader, the host
emulation build of the kernel cores (tests/emu) and a monkeypatch that routes
``diffdrr_amd.ops`` launches to it so that the autograd wiring can be checked
without a GPU.  None of this is reachable from the product package.
"""
from __future__ import annotations

import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_SRC = os.path.join(ROOT, "tests", "emu", "ddrr_emu.cpp")
EMU_SO = os.path.join(ROOT, "tests", "emu", "_build", "libddrr_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    """max |a - b| / max |b|: the image-normalised error of SURVEY.md section 8(d)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def build_emu():
    csrc = os.path.join(ROOT, "diffdrr_amd", "csrc")
    deps = [EMU_SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] \
        + [os.path.join(ROOT, "include", "diffdrr_hip.h")]
    if os.path.exists(EMU_SO) and all(os.path.getmtime(d) <= os.path.getmtime(EMU_SO)
                                      for d in deps):
        return EMU_SO
    os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
    subprocess.run(
        ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-mfma", "-ffp-contract=off",
         "-Wno-unknown-pragmas", EMU_SRC, "-o", EMU_SO], check=True)
    return EMU_SO


@pytest.fixture(scope="session")
def emu_lib():
    from diffdrr_amd._lib import DdrrLibrary

    return DdrrLibrary(build_emu())


@pytest.fixture()
def emulated_ops(emu_lib, monkeypatch):
    """Route diffdrr_amd.ops launches to the host emulation of the kernel cores
    (same source as the HIP kernels, compiled for the CPU) so that the Python /
    autograd layer can be exercised on CPU tensors."""
    from diffdrr_amd import ops

    monkeypatch.setattr(ops, "_require_gpu", lambda volume: None)
    monkeypatch.setattr(ops, "on_device", lambda t: True)
    monkeypatch.setattr(ops, "_launch", lambda name, device, *a: emu_lib.call(name, *a, None))
    monkeypatch.setattr(ops, "_query", lambda name, *a: emu_lib.query(name, *a))
    return ops


def check_reference_rays_through_swapped_renderer(renderer, device, ops):
    """The renderer seam of the reference (diffdrr/drr.py:94-101, 209-224) with the rays of an
    UNMODIFIED reference ``DRR`` (tests/golden/reference_drr_rays.npz, made by
    tests/golden/make_golden.py from /root/reference): ``diffdrr_amd.Siddon`` / ``Trilinear`` get
    them exactly as a renderer swapped into ``diffdrr.drr.DRR`` would -- after the four lines of
    ``DRR.render`` that precede the renderer call -- with ``detector_shape`` promised but NOT
    trusted, so the grid check runs before the volume-stationary kernels take the rays.
    Image and ray gradients against what the reference's own renderer returned."""
    import torch

    import diffdrr_amd

    g = golden("reference_drr_rays")
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(device)  # noqa: E731
    vol, Ainv = T(g["volume"]), T(g["affine_inverse"])
    source, target = T(g["source"]).requires_grad_(), T(g["target"]).requires_grad_()
    H, W = int(g["geo_height"]), int(g["geo_width"])
    mod = (diffdrr_amd.Siddon if renderer == "siddon" else diffdrr_amd.Trilinear)(voxel_shift=0.5)
    mod.detector_shape = (H, W)  # the promise INTEGRATION.md asks the integrator to make
    kw = {} if renderer == "siddon" else {"n_points": 80}
    calls = []
    real = ops.rays_form_detector_grid
    ops.rays_form_detector_grid = lambda *a, **k: calls.append(real(*a, **k)) or calls[-1]
    try:
        # reference drr.py:201-205: img = ray length in world units, rays to voxel coordinates
        img = (target - source).norm(dim=-1).unsqueeze(1)
        apply = lambda x: x @ Ainv[:3, :3].T + Ainv[:3, 3]  # noqa: E731
        out = mod(vol, apply(source), apply(target), img, **kw)
    finally:
        ops.rays_form_detector_grid = real
    assert calls == [True]  # checked once, accepted: the brick kernels rendered this
    assert out.shape == g[f"{renderer}_img"].shape
    assert rel_err(out.detach().cpu().numpy(), g[f"{renderer}_img"]) < 1e-4
    (out * T(g["grad_out"])).sum().backward()
    gs, gt = source.grad.cpu().numpy(), target.grad.cpu().numpy()
    rs, rt = g[f"{renderer}_g_source"], g[f"{renderer}_g_target"]
    es, et = g[f"{renderer}_g_source_f64"], g[f"{renderer}_g_target_f64"]  # the reference in fp64
    # per pose sums (what a pose gradient is made of), against the exact gradient, allowance:
    # twice what the reference's own fp32 arithmetic loses (3e-4 Siddon, 6e-3 trilinear here)
    # (the marcher's source gradient is dominated by the path through the batch-global marching
    # range -- d/d alphamin summed over every sample of every ray, routed to one ray: a heavily
    # cancelling sum that the reference's fp32 autograd itself gets 3e-3 wrong here; the record
    # kernels' fp32 atomics add their own share on the device)
    slack = 1e-3 if renderer == "siddon" else 1.5e-2
    assert rel_err(gs, es) < 2 * rel_err(rs, es) + slack
    assert rel_err(gt.sum(1), et.sum(1)) < 2 * rel_err(rt.sum(1), et.sum(1)) + slack
    # per ray: as many rays within 1e-3 of the exact gradient as the reference's fp32 has
    close = lambda a: float((np.abs(a - et).max(-1) <= 1e-3 * np.abs(et).max()).mean())  # noqa: E731
    assert close(gt) >= close(rt) - 0.01 and close(gt) > 0.97


def check_metrics_against_reference(device):
    """diffdrr_amd.metrics against tests/golden/metrics.npz: values and autograd gradients of the
    UNMODIFIED reference's NormalizedCrossCorrelation2d (whole image / patch_size),
    MultiscaleNormalizedCrossCorrelation2d and GradientNormalizedCrossCorrelation2d
    (diffdrr/metrics.py:21-104).  Yardstick: the reference in float64; allowance: twice what the
    reference's own float32 evaluation loses, + 2e-6 (values) / 2e-5 of the largest gradient."""
    import torch

    from diffdrr_amd import metrics as M

    g = golden("metrics")
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(device)  # noqa: E731
    w = T(g["w"])
    cases = {
        "ncc": M.NormalizedCrossCorrelation2d(),
        "ncc_patch5": M.NormalizedCrossCorrelation2d(patch_size=5),
        "multiscale": M.MultiscaleNormalizedCrossCorrelation2d([None, 7], [0.5, 0.5]),
        "gncc_sigma0": M.GradientNormalizedCrossCorrelation2d(sigma=0.0),
        "gncc_sigma1": M.GradientNormalizedCrossCorrelation2d(sigma=1.0),
        "gncc_patch7_sigma0": M.GradientNormalizedCrossCorrelation2d(patch_size=7, sigma=0.0),
    }
    assert rel_err(M.Sobel(0.0)(T(g["a"])).cpu().numpy(), g["sobel_a"]) < 1e-6
    for name, crit in cases.items():
        x1, x2 = T(g["a"]).requires_grad_(), T(g["b"]).requires_grad_()
        val = crit(x1, x2)
        (val * w).sum().backward()
        v64 = g[f"{name}_f64"]
        assert np.abs(val.detach().cpu().numpy() - v64).max() <= \
            2 * np.abs(g[f"{name}_f32"] - v64).max() + 2e-6, name
        for mine, key in ((x1.grad, "g1"), (x2.grad, "g2")):
            e64, r32 = g[f"{name}_{key}_f64"], g[f"{name}_{key}_f32"]
            assert rel_err(mine.cpu().numpy(), e64) < 2 * rel_err(r32, e64) + 2e-5, (name, key)


def _topk_sum(img):
    """the callable reducefn of the fixtures (tests/golden/make_golden.py topk_sum)"""
    return img.sort(descending=True).values[..., :6].sum(dim=-1)


# The keyword combinations that go through the materialising general path
# (csrc/general_core.h): (fixture, renderer, ctor kwargs, call kwargs, dtypes it is general for).
# "f32" entries are the float32 cases no fused kernel takes; every "f64" entry is a float64
# module outside the default fused fp64 kernels.
GENERAL_CASES = [
    ("siddon_bilinear_mask", "Siddon", {"mode": "bilinear"}, {}, ("f32", "f64")),
    ("siddon_align_mask", "Siddon", {}, {"align_corners": True}, ("f32", "f64")),
    ("siddon_mask_max", "Siddon", {"reducefn": "max"}, {}, ("f32", "f64")),
    ("siddon_bilinear_callable", "Siddon", {"mode": "bilinear", "reducefn": _topk_sum}, {},
     ("f32", "f64")),
    ("siddon_align_max", "Siddon", {"reducefn": "max"}, {"align_corners": True}, ("f32", "f64")),
    ("siddon_bilinear_max", "Siddon", {"mode": "bilinear", "reducefn": "max"}, {}, ("f32", "f64")),
    ("siddon_bilinear_stopgrad", "Siddon",
     {"mode": "bilinear", "stop_gradients_through_grid_sample": True}, {}, ("f32", "f64")),
    ("siddon_max", "Siddon", {"reducefn": "max"}, {}, ("f64",)),
    ("siddon_stopgrad", "Siddon", {"stop_gradients_through_grid_sample": True}, {}, ("f64",)),
    ("siddon_bilinear", "Siddon", {"mode": "bilinear"}, {}, ("f64",)),
    ("siddon_align_corners", "Siddon", {}, {"align_corners": True}, ("f64",)),
    ("siddon_callable", "Siddon", {"reducefn": _topk_sum}, {}, ("f64",)),
    ("siddon_mask", "Siddon", {}, {}, ("f64",)),
    ("siddon_shift0", "Siddon", {"voxel_shift": 0.0}, {}, ("f64",)),
    ("trilinear_nearest_mask", "Trilinear", {"mode": "nearest"}, {"n_points": 40}, ("f32", "f64")),
    ("trilinear_mask_callable", "Trilinear", {"reducefn": _topk_sum}, {"n_points": 40},
     ("f32", "f64")),
    ("trilinear_align_corners", "Trilinear", {}, {"n_points": 45, "align_corners": True},
     ("f32", "f64")),
    ("trilinear_nearest", "Trilinear", {"mode": "nearest"}, {"n_points": 45}, ("f32", "f64")),
    ("trilinear_nearest_max", "Trilinear", {"mode": "nearest", "reducefn": "max"},
     {"n_points": 33}, ("f64",)),
    ("trilinear_mask", "Trilinear", {}, {"n_points": 40}, ("f64",)),
    ("trilinear_max", "Trilinear", {"reducefn": "max"}, {"n_points": 37}, ("f64",)),
    ("trilinear_callable", "Trilinear", {"reducefn": _topk_sum}, {"n_points": 40}, ("f64",)),
    ("trilinear_shift0", "Trilinear", {"voxel_shift": 0.0}, {"n_points": 40}, ("f64",)),
]


def general_case_ids():
    return [(name, tag) for name, _, _, _, tags in GENERAL_CASES for tag in tags]


def check_general_case(name, tag, device):
    """One fixture of the unmodified reference (output + autograd gradients in float32 and
    float64) through the product renderer in dtype `tag` on `device`."""
    import torch

    import diffdrr_amd

    _, cls, ctor, call, _ = next(c for c in GENERAL_CASES if c[0] == name)
    g = golden(name)
    dt = torch.float32 if tag == "f32" else torch.float64
    npdt = np.float32 if tag == "f32" else np.float64
    leaf = lambda k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=npdt)).to(device)  # noqa: E731
    vol, src, tgt = (leaf(k).requires_grad_() for k in ("volume", "source", "target"))
    img = leaf(f"img_{tag}").requires_grad_()
    kw = dict(call)
    if "mask" in g.files:
        kw["mask"] = leaf("mask")
    renderer = getattr(diffdrr_amd, cls)(**ctor)
    out = renderer(vol, src, tgt, img, **kw)
    ref = g[f"out_{tag}"]
    assert out.dtype == dt and tuple(out.shape) == ref.shape
    marcher = cls == "Trilinear"
    # float64: the marcher's sample fractions are an fp32 linspace table whose aten kernel
    # rounds a few entries one ulp away from the scalar formula (test_host_api.py): 1e-6
    tol_out = 1e-4 if tag == "f32" else (1e-6 if marcher else 1e-10)
    tol_grad = 2e-3 if tag == "f32" else (1e-5 if marcher else 1e-8)
    assert rel_err(out.detach().cpu().numpy(), ref) < tol_out
    if f"grad_out_{tag}" not in g.files:
        return
    grads = torch.autograd.grad(out, (src, tgt, img, vol), leaf(f"grad_out_{tag}"),
                                allow_unused=True)
    for key, mine in zip(("g_source", "g_target", "g_img", "g_volume"), grads):
        k = f"{key}_{tag}"
        if k not in g.files:  # the reference's autograd returned None (stop-gradients)
            assert mine is None or float(mine.abs().max()) == 0.0, key
            continue
        want = g[k]
        if mine is None:
            assert float(np.abs(want).max()) == 0.0, key
            continue
        assert rel_err(mine.cpu().numpy(), want) < tol_grad, (key, rel_err(mine.cpu().numpy(), want))


def check_trilinear_channels_on_bricks(device, dims, det, n_points):
    """The marcher's mask_to_channels on the volume-stationary bricks
    (ddrr_trilinear_forward_channels_bricks) against the per-ray channel kernel (pinned to the
    reference by the trilinear_mask fixture), the plain march, and through the module with
    gradients (the backward is the per-ray channel kernel's)."""
    import torch

    from diffdrr_amd import DRR, Trilinear, convert, ops
    from diffdrr_amd.data import synthetic_subject

    H, W = det
    sub = synthetic_subject(dims, kind="phantom", seed=5, n_labels=7)
    drr = DRR(sub, sdd=700.0, height=H, width=W, delx=3.0, renderer="trilinear").to(device)
    rng = np.random.default_rng(11)
    blocks = rng.integers(0, 256, size=tuple((d + 4) // 5 for d in dims)).astype(np.uint8)
    labels = torch.from_numpy(np.kron(blocks, np.ones((5, 5, 5), np.uint8))
                              [:dims[0], :dims[1], :dims[2]].copy()).to(device)
    rot = torch.tensor([[0.3, 0.2, -0.1], [1.5, 0.1, 0.0], [0.0, 0.0, 0.0]], device=device)
    xyz = torch.tensor([[5.0, 480.0, -3.0], [0.0, 460.0, 0.0], [0.0, 450.0, 0.0]], device=device)
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    V = drr.density
    a0, a1 = (x.reshape(1) for x in ops.trilinear_alpha_range(s, t, V.shape)) \
        if device != "cpu" else (torch.tensor([0.3]), torch.tensor([0.8]))
    C = 256
    ch = ops.trilinear_forward_channels_bricks(V, labels, C, s, t, L, a0, a1, (H, W),
                                               n_points=n_points).cpu().numpy()
    per_ray = ops.trilinear_forward_channels(V, labels, C, s, t, L, a0, a1,
                                             n_points=n_points).cpu().numpy()
    # (the staged values keep a 16-bit mantissa: 2^-17 relative per voxel)
    assert rel_err(ch, per_ray) < 3e-5
    assert np.all(ch[per_ray == 0] == 0)  # every sample in its own channel
    plain = ops.trilinear_forward(V, s, t, L, a0, a1, n_points=n_points)
    plain = (plain[0] if isinstance(plain, tuple) else plain).cpu().numpy()
    assert rel_err(ch.sum(1), plain) < 3e-5
    ch8 = ops.trilinear_forward_channels_bricks(V, labels, 8, s, t, L, a0, a1, (H, W),
                                                n_points=n_points).cpu().numpy()
    assert np.array_equal(ch8, ch[:, :8]) or rel_err(ch8, ch[:, :8]) < 1e-6
    # the module route takes this kernel for a detector grid and stays differentiable
    taken = []
    orig = ops.trilinear_forward_channels_bricks
    ops.trilinear_forward_channels_bricks = lambda *a, **k: (taken.append(1), orig(*a, **k))[1]
    try:
        r = rot[:2].clone().requires_grad_()
        x = xyz[:2].clone().requires_grad_()
        kw = dict(parameterization="euler_angles", convention="ZXY", n_points=n_points)
        chm = drr(r, x, mask_to_channels=True, **kw)
    finally:
        ops.trilinear_forward_channels_bricks = orig
    assert taken and chm.shape[1] == 7
    one = drr(r, x, **kw)
    assert rel_err(chm.sum(1, keepdim=True).detach().cpu().numpy(), one.detach().cpu().numpy()) < 3e-5
    w = torch.rand(one.shape, generator=torch.Generator().manual_seed(2)).to(device)
    ga = torch.autograd.grad((chm.sum(1, keepdim=True) * w).sum(), [r, x])
    gb = torch.autograd.grad((one * w).sum(), [r, x])
    for a, b in zip(ga, gb):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-3


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from diffdrr_amd import _lib

    _lib.get_lib()  # fail loudly if the HIP library was not built
    return torch.device("cuda:0")
