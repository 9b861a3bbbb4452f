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
    reference by the trilinear_mask fixture), the plain march, its ray backward
    (ddrr_trilinear_backward_channels_bricks) against the per-ray channel backward, and through
    the module with gradients."""
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
    # the ray backward on the bricks (the weighted record) against the per-ray channel backward
    # (pinned to the reference's autograd by the trilinear_mask fixture); labels >= 200 dropped
    Cb = 200
    B, N = L.shape
    go = torch.rand(B, Cb, N, generator=torch.Generator().manual_seed(4)).to(device)
    rb = ops.trilinear_backward_channels_bricks(V, labels, s, t, L, go, a0, a1, (H, W),
                                                n_points=n_points)
    rp = ops.trilinear_backward_channels(V, labels, s, t, L, go, a0, a1, n_points=n_points)
    for key, tol in (("g_img", 3e-5), ("g_target", 2e-4), ("g_source", 2e-4), ("g_alpha", 2e-4)):
        mine, want = rb[key].cpu().numpy(), rp[key].cpu().numpy()
        assert rel_err(mine, want) < tol, (key, rel_err(mine, want))
    # the volume gradient on the owner bricks (ddrr_trilinear_backward_channels_volume_bricks)
    # against the per-ray kernel's global atomics; labels >= Cb get no weight
    gvb = ops.trilinear_backward_channels_volume_bricks(labels, s, t, L, go, a0, a1, (H, W),
                                                        n_points=n_points).cpu().numpy()
    gvr = ops.trilinear_backward_channels(V, labels, s, t, L, go, a0, a1, n_points=n_points,
                                          want_rays=False, want_img=False, want_alpha=False,
                                          want_volume=True)["g_volume"].cpu().numpy()
    assert np.isfinite(gvb).all()
    assert rel_err(gvb, gvr) < 5e-5, rel_err(gvb, gvr)
    # the module route takes these kernels for a detector grid and stays differentiable
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


# ------------------------------------------------------------------ 16-bit brick storage guard
Q16_RANGE_OVER_LEVEL = 12.0  # csrc/brick_step.h kQ16RangeOverLevel


def brick_levels(vol, bdims):
    """numpy restatement of brick_range_kernel + q16_usable (csrc/bricks_fwd.hip,
    csrc/brick_step.h): per brick of the ``bdims`` grid its level -- the smallest mean |V| of any
    4^3 block, over the non-zero voxels when the brick's minimum is 0 -- and whether the brick
    takes the fp32 path.  -> (level volume: the brick's level at every voxel of a QUANTISED brick,
    0 elsewhere; fallback flags (nbx, nby, nbz))"""
    v = np.asarray(vol, dtype=np.float32)
    D = v.shape
    nb = [-(-D[a] // bdims[a]) for a in range(3)]
    level_vol = np.zeros(D, np.float64)
    flags = np.zeros(nb, bool)
    for bx in range(nb[0]):
        for by in range(nb[1]):
            for bz in range(nb[2]):
                sl = tuple(slice(i * b, min((i + 1) * b, d)) for i, b, d in zip((bx, by, bz), bdims, D))
                blk = v[sl]
                with np.errstate(invalid="ignore", over="ignore"):
                    lo, hi = np.float32(blk.min()), np.float32(blk.max())
                    rng = hi - lo
                    level = np.inf
                    for x in range(0, blk.shape[0], 4):
                        for y in range(0, blk.shape[1], 4):
                            for z in range(0, blk.shape[2], 4):
                                s = blk[x:x + 4, y:y + 4, z:z + 4].astype(np.float64)
                                n = int((s != 0).sum()) if lo == 0 else s.size
                                if n:
                                    level = min(level, float(np.abs(s).sum()) / n)
                    if not np.isfinite(blk).all():
                        ok = False
                    elif rng == 0:
                        ok = abs(lo) < 2.0 ** 40
                    else:
                        ok = bool(2.0 ** -60 <= rng < 2.0 ** 40 and abs(lo) < 2.0 ** 40
                                  and rng <= Q16_RANGE_OVER_LEVEL * level)
                flags[bx, by, bz] = not ok
                if ok:
                    level_vol[sl] = level if np.isfinite(level) else 0.0
    return level_vol, flags


def guard_scene(device, det=(40, 40), delx=2.0, dims=(64, 64, 128)):
    """Rays of a few poses through a (64, 64, 128) volume (2 x 2 x 2 double bricks, 2 x 2 x 4 of
    32^3) as ``DRR`` builds them: (s, t, L) voxel-space tensors on ``device`` and the module."""
    import torch

    from diffdrr_amd import DRR, convert
    from diffdrr_amd.data import make_subject

    drr = DRR(make_subject(torch.zeros(*dims)), sdd=600.0, height=det[0], width=det[1], delx=delx)
    rot = torch.tensor([[0.0, 0.0, 0.0], [0.5, -0.3, 0.2], [-0.7, 0.4, 1.1]])
    xyz = torch.tensor([[0.0, 400.0, 0.0], [5.0, 380.0, -8.0], [-12.0, 420.0, 6.0]])
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    return s.to(device), t.to(device), L.to(device)


def guard_volumes():
    """name -> (volume (64, 64, 128) float32, comment).  Everything the 16-bit block quantisation
    must NOT be trusted with, next to what it is meant for."""
    rng = np.random.default_rng(7)
    D = (64, 64, 128)
    body = (0.5 + 0.01 * (rng.random(D) - 0.5)).astype(np.float32)  # 0.5 +- 0.005
    vols = {"noise": rng.random(D).astype(np.float32)}
    for f in (2.0, 20.0, 200.0, 2000.0, 1e4, 1e6):
        v = body.copy()
        v[40, 37, 70] = 0.5 * f  # in view of the detector's centre
        vols[f"outlier_x{f:g}_in_view"] = v
        v = body.copy()
        v[1, 62, 2] = 0.5 * f    # a corner no ray of the small detector passes
        vols[f"outlier_x{f:g}_out_of_view"] = v
    # un-normalised HU: air -1000, soft tissue ~40 +- 30, a bone shell, a metal sphere at 30000
    x, y, z = np.meshgrid(*(np.linspace(-1, 1, d) for d in D), indexing="ij")
    r = np.sqrt((x / 0.8) ** 2 + (y / 0.7) ** 2 + (z / 0.9) ** 2)
    hu = np.full(D, -1000.0, np.float32)
    hu[r < 1.0] = (40.0 + 30.0 * rng.standard_normal(D)).astype(np.float32)[r < 1.0]
    hu[(r > 0.55) & (r < 0.62)] = (1100.0 + 400.0 * rng.random(D)).astype(np.float32)[(r > 0.55) & (r < 0.62)]
    hu[np.sqrt((x - 0.2) ** 2 + (y + 0.1) ** 2 + (z - 0.3) ** 2) < 0.06] = 30000.0
    vols["hu_with_metal"] = hu
    vols["negative"] = (-rng.random(D)).astype(np.float32)
    vols["mixed_sign"] = rng.standard_normal(D).astype(np.float32)
    const = np.full(D, 0.37, np.float32)
    const[:32] = 0.0  # bricks of zeros (skipped as empty space) next to bricks of one value
    vols["constant_bricks"] = const
    half = body.copy()
    half[:, :, ::2] *= 1e-3  # every brick half bright, half 1000x dimmer, finely interleaved
    vols["dim_and_bright_layers"] = half
    step = body.copy()
    step[:, 32:, :] *= 1e-3  # ... and in two slabs that rays can see separately
    vols["dim_and_bright_slabs"] = step
    return vols


def check_brick_storage_guard(ops, device, name, storage, bdims):
    """A 16-bit storage of ddrr_siddon_forward_bricks against the fp64 oracle on one of
    guard_volumes(): the plain image-normalised 1e-4, the stated per-pixel bound
    (|error| <= 9.2e-5 of the line integral of the brick levels + fp32 rounding), the number of
    bricks sent to the fp32 path, and the record."""
    import torch

    import oracle

    vol = guard_volumes()[name]
    s, t, L = guard_scene(device)
    B, N = L.shape
    V = torch.from_numpy(vol).to(device)
    a64 = (vol.astype(np.float64), s.cpu().numpy().astype(np.float64),
           t.cpu().numpy().astype(np.float64), L.cpu().numpy().astype(np.float64))
    ref = oracle.siddon(*a64)["out"].reshape(B, N)
    if "out_of_view" in name:
        delta = np.zeros_like(a64[0])
        delta[1, 62, 2] = 1.0
        assert not oracle.siddon(delta, *a64[1:])["out"].any()  # really out of view
    level_vol, flags = brick_levels(vol, bdims)
    # the quantisation's share of a pixel's error: what it adds to the fp32 bricks' own result
    # (same clip, same alphas, same walk: fp32 geometry error is common to both), allowance for
    # the different grouping of the sums: 3e-6 of the line integral of |V|
    finite = np.where(np.isfinite(a64[0]), np.abs(a64[0]), 0.0)
    bound = (Q16_RANGE_OVER_LEVEL / 131070.0) * oracle.siddon(level_vol, *a64[1:])["out"].reshape(B, N) \
        + 3e-6 * oracle.siddon(finite, *a64[1:])["out"].reshape(B, N)
    out, aux = ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage=storage, want_aux=True)
    out0, _ = ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage=storage)
    assert ops.brick_fallbacks(V, storage) == (int(flags.sum()), flags.size)
    ref32, aux32 = ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage="f32", want_aux=True)
    ref32_0, _ = ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage="f32")
    # (like against like: the forward-only launches walk with accumulated alphas, the launches
    # with a record with the plane counters -- the same for both storages)
    for img, f32 in ((out, ref32), (out0, ref32_0)):
        img, f32 = (x.cpu().numpy().astype(np.float64) for x in (img, f32))
        err = np.abs(img - ref)
        assert err.max() <= 1e-4 * np.abs(ref).max(), (name, err.max() / np.abs(ref).max())
        errq = np.abs(img - f32)
        assert (errq <= bound + 1e-30).all(), (name, float((errq / (bound + 1e-30)).max()))
    # the record: the same crossings over voxel values within the bound -> the ray gradients of
    # the fp32 bricks, per pose sum (single rays flip where crossings tie)
    go = torch.rand(B, N, generator=torch.Generator().manual_seed(5)).to(device)
    gq = ops.siddon_backward_rays(aux, go, s, t, L)
    gf = ops.siddon_backward_rays(aux32, go, s, t, L)
    assert rel_err(gq[2].cpu().numpy(), gf[2].cpu().numpy()) < 1e-4  # d/d img = g * I
    same = (gq[1] - gf[1]).abs().amax(-1) <= 2e-3 * gf[1].abs().max()
    assert same.float().mean().item() > 0.99, name
    return int(flags.sum()), flags.size


def check_channel_backward_on_bricks(ops, device):
    """ddrr_siddon_backward_channels_bricks (the record of the volume weighted by every voxel's own
    incoming gradient, then ddrr_siddon_backward_rays) and ddrr_siddon_backward_channels_volume_bricks
    (the volume gradient) against the fp64 oracle's autograd of the
    mask branch (reference renderers.py:77-89) and against the per-ray channel backward: several
    bricks, up to 200 labels of which the last 56 have no channel, forward image for scale."""
    import torch

    import oracle
    from diffdrr_amd import DRR, convert
    from diffdrr_amd.data import synthetic_subject

    D, H, W, C = (40, 70, 36), 24, 31, 144
    drr = DRR(synthetic_subject(D, kind="noise", seed=5), sdd=600.0, height=H, width=W, delx=3.0)
    rng = np.random.default_rng(11)
    blocks = rng.integers(0, 200, size=(5, 9, 5)).astype(np.uint8)  # 8^3-voxel label blocks
    labels = np.kron(blocks, np.ones((8, 8, 8), np.uint8))[:D[0], :D[1], :D[2]].copy()
    rot = torch.tensor([[0.3, 0.2, -0.1], [1.5, 0.1, 0.0], [0.0, 1.45, 0.2]])
    xyz = torch.tensor([[5.0, 420.0, -3.0], [0.0, 400.0, 0.0], [2.0, 380.0, 1.0]])
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    B, N = L.shape
    V = drr.density.to(device)
    lab = torch.from_numpy(labels).to(device)
    go = torch.rand(B, C, N, generator=torch.Generator().manual_seed(4))
    sd, td, Ld, god = s.to(device), t.to(device), L.to(device), go.to(device)
    gs, gt, gi = ops.siddon_backward_channels_bricks(V, lab, sd, td, Ld, god, (H, W))
    ps, pt, pi, _ = ops.siddon_backward_channels(V, lab, sd, td, Ld, god, det=(H, W))
    f64 = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
    # the oracle: autograd of the mask branch with C channels = labels >= C dropped
    lab_c = np.where(labels < C, labels, 255).astype(np.float64)
    go_full = np.zeros((B, 256, N))
    go_full[:, :C] = go.numpy()
    o = oracle.siddon_channels_grad(f64(drr.density.numpy()), lab_c, f64(s.numpy()), f64(t.numpy()),
                                    f64(L.numpy()), go_full)
    assert rel_err(gi.cpu().numpy(), o["g_img"].reshape(B, N)) < 1e-4
    assert rel_err(gi.cpu().numpy(), pi.cpu().numpy()) < 3e-5
    # Target gradients.  A ray holding a crossing pair that ties in fp32 books a voxel difference
    # on one axis or the other by its walk's own rounding -- here the difference of the WEIGHTED
    # volume, O(1) at the faces of the 8^3 label blocks (random weights, dropped labels): a handful
    # of rays (measured: 3 of 2232 for the bricks, 2 for the per-ray kernel, each right where the
    # other is not) are off by a few per cent of the largest gradient, everything else agrees with
    # fp64 to 1e-3; the per-pose sums inherit those few rays.
    og = o["g_target"].reshape(B, N, 3)
    close = lambda m: float((np.abs(m - og).max(-1) <= 1e-3 * np.abs(og).max()).mean())  # noqa: E731
    for mine in (gt, pt):
        m = mine.cpu().numpy()
        assert close(m) > 0.99, close(m)
        assert rel_err(m.sum(1), og.sum(1)) < 5e-2
    assert rel_err(gs.sum(1).cpu().numpy(), o["g_source"].reshape(B, -1, 3).sum(1)) < 5e-2
    same = (gt - pt).abs().amax(-1) <= 1e-3 * pt.abs().max()
    assert same.float().mean().item() > 0.99
    # the volume gradient with the brick in LDS as the accumulator (24-bit fixed point over the
    # label byte; ddrr_siddon_backward_channels_volume_bricks) against the fp64 oracle and the
    # per-ray kernel's global atomics: every voxel stored (no NaN left), labels >= C get nothing
    gv = ops.siddon_backward_channels_volume_bricks(lab, sd, td, Ld, god, (H, W)).cpu().numpy()
    pv = ops.siddon_backward_channels(V, lab, sd, td, Ld, god, det=(H, W), want_rays=False,
                                      want_img=False, want_volume=True)[3].cpu().numpy()
    assert np.isfinite(gv).all()
    assert rel_err(gv, o["g_volume"]) < 1e-4, rel_err(gv, o["g_volume"])  # (fp32 sums of ~50 terms)
    assert rel_err(gv, pv) < 5e-5
    assert np.all(gv[labels >= C] == 0)


def check_channel_backward_on_bricks_smooth(ops, device):
    """the channel ray backward on the bricks builds its record from packed words
    whose values keep a 16-bit mantissa (2^-17 relative per voxel), and a record is made of
    DIFFERENCES of neighbouring voxels.  On a noise volume tie flips hide what that costs; here a
    smooth volume, labels in large smooth regions and per-channel weights that change by 1 % from
    one label to the next: no O(1) jump anywhere for a tie to book on the wrong axis, so EVERY
    ray's gradient must agree with the fp64 oracle, and the truncation error is bounded by a test."""
    import torch

    import oracle
    from diffdrr_amd import DRR, convert
    from diffdrr_amd.data import make_subject

    D, H, W, C = (48, 70, 40), 24, 31, 12
    x, y, z = np.meshgrid(*[np.arange(d, dtype=np.float64) for d in D], indexing="ij")
    vol = (0.55 + 0.25 * np.sin(x / 9.0) * np.cos(y / 11.0) + 0.15 * np.sin(z / 7.0 + 0.3 * x / 9.0))
    vol = vol.astype(np.float32)
    labels = np.clip((x / 8.0 + y / 14.0 + z / 20.0).astype(np.int64), 0, C - 1).astype(np.uint8)
    drr = DRR(make_subject(torch.from_numpy(vol)), sdd=600.0, height=H, width=W, delx=3.0)
    rot = torch.tensor([[0.3, 0.2, -0.1], [1.1, 0.15, 0.05], [0.1, 1.3, 0.2]])
    xyz = torch.tensor([[5.0, 420.0, -3.0], [0.0, 400.0, 0.0], [2.0, 380.0, 1.0]])
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous()
        t = drr.affine_inverse(target).contiguous()
    B, N = L.shape
    V = drr.density.to(device)
    lab = torch.from_numpy(labels).to(device)
    g = torch.Generator().manual_seed(9)
    pix = 0.5 + torch.rand(B, 1, N, generator=g)
    go = (pix * (1.0 + 0.01 * torch.arange(C).view(1, C, 1))).contiguous()
    sd, td, Ld, god = s.to(device), t.to(device), L.to(device), go.to(device)
    gs, gt, gi = ops.siddon_backward_channels_bricks(V, lab, sd, td, Ld, god, (H, W))
    ps, pt, pi, _ = ops.siddon_backward_channels(V, lab, sd, td, Ld, god, det=(H, W))
    f64 = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
    go_full = np.zeros((B, 256, N))
    go_full[:, :C] = go.numpy()
    o = oracle.siddon_channels_grad(f64(vol), f64(labels), f64(s.numpy()), f64(t.numpy()), f64(L.numpy()),
                                    go_full)
    og, os_ = o["g_target"].reshape(B, N, 3), o["g_source"].reshape(B, -1, 3)
    gtn, ptn = gt.cpu().numpy(), pt.cpu().numpy()
    assert rel_err(gi.cpu().numpy(), o["g_img"].reshape(B, N)) < 1e-5
    # against fp64: all rays but the handful that hold a crossing pair tied in fp32 (the per-ray
    # kernel on exact fp32 values misses the same ones: measured on the host emulation 0.22 % of
    # 2232 rays for either, largest error 1.3e-2 for both) ...
    near64 = lambda m: float((np.abs(m - og).max(-1) <= 1e-3 * np.abs(og).max()).mean())  # noqa: E731
    assert near64(gtn) > 0.995, near64(gtn)
    assert abs(near64(gtn) - near64(ptn)) < 2e-3
    # ... and against the per-ray kernel, which walks the volume's own fp32 values: 1e-4 on (all but
    # the rays where the two walks' roundings flip a tie differently) every ray -- the bound on what
    # the 16-bit mantissas of the packed words cost a smooth volume's gradients
    same = float((np.abs(gtn - ptn).max(-1) <= 1e-4 * np.abs(ptn).max()).mean())
    assert same > 0.998, same
    assert rel_err(gtn.sum(1), og.sum(1)) < 1e-2
    assert rel_err(gs.sum(1).cpu().numpy(), os_.sum(1)) < 1e-2


def check_filter_intersections_outside_volume(device):
    """``Siddon(filter_intersections_outside_volume=True)``.  The
    reference's branch raises TypeError (renderers.py:118 vs :124); the fixture holds the branch
    with its call completed (tests/golden/make_golden_filter.py): the intended semantics.  The
    product's render with the flag equals the fixture AND its own default render bit for bit."""
    import torch

    import diffdrr_amd

    g = golden("siddon_filter_outside")
    assert bool(g["raises_type_error"])  # what the unmodified reference does with the flag
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(device)  # noqa: E731
    B, N, _ = g["target"].shape
    res = {}
    for flag in (True, False):
        vol, src, tgt = (T(g[k]).requires_grad_() for k in ("volume", "source", "target"))
        img = T(g["filtered_img_f32"].reshape(B, 1, N)).requires_grad_()
        out = diffdrr_amd.Siddon(filter_intersections_outside_volume=flag)(vol, src, tgt, img)
        grads = torch.autograd.grad(out, (src, tgt, img, vol), T(g["filtered_grad_out_f32"]))
        res[flag] = [out.detach().cpu().numpy()] + [x.cpu().numpy() for x in grads]
    names = ("out", "g_source", "g_target", "g_img", "g_volume")
    for k, a, b in zip(names, res[True], res[False]):
        # the flag changes nothing in the product (bit for bit, but for the volume gradient's fp32
        # atomics on the device, whose order differs from launch to launch)
        assert np.array_equal(a, b) or (k == "g_volume" and rel_err(a, b) < 1e-6), k
        assert a.shape == g[f"filtered_{k}_f32"].shape
        tol = 1e-4 if k == "out" else 1e-3
        assert rel_err(a, g[f"filtered_{k}_f64"]) < tol, k
        assert rel_err(a, g[f"filtered_{k}_f32"]) < tol, k
    # ... and the intended semantics are the default's (fp64: to the rounding of the shorter sums)
    for k in names:
        assert rel_err(g[f"filtered_{k}_f64"], g[f"default_{k}_f64"]) < 1e-12


def check_euler_inference_path(device):
    """``drr(rot, xyz, parameterization="euler_angles")`` with nothing to differentiate takes two
    launches (DRR._render_euler_inference: pose -> matrix -> rays + the clears, the brick kernel);
    with a gradient wanted it takes a differentiable path (check_euler_differentiable_path): the same
    kernels' arithmetic, so the images agree to the atomics' order of summation -- in radians and
    degrees, for every brick storage, and through ``convert`` + ``drr(pose)`` (reference
    drr.py:155-188)."""
    import torch

    from diffdrr_amd import DRR, convert
    from diffdrr_amd.data import synthetic_subject

    drr = DRR(synthetic_subject(40, kind="phantom", seed=3), sdd=600.0, height=30, width=26, delx=3.0).to(device)
    g = torch.Generator().manual_seed(9)
    rot = ((torch.rand(3, 3, generator=g) - 0.5) * 0.8).to(device)
    xyz = (torch.tensor([0.0, 400.0, 0.0]) + (torch.rand(3, 3, generator=g) - 0.5) * 20).to(device)
    calls = []
    orig = drr._render_euler_inference
    drr._render_euler_inference = lambda *a: calls.append(1) or orig(*a)
    for storage in ("f32", "q16", "q16p"):
        drr.renderer.brick_storage = storage
        with torch.no_grad():
            fast = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
            fast_deg = drr(rot * 180 / torch.pi, xyz, parameterization="euler_angles", convention="ZXY", degrees=True)
            via_pose = drr(convert(rot, xyz, parameterization="euler_angles", convention="ZXY"))
        n = len(calls)
        slow = drr(rot.clone().requires_grad_(), xyz, parameterization="euler_angles", convention="ZXY")
        assert len(calls) == n and slow.requires_grad and not fast.requires_grad
        scale = float(slow.detach().abs().max())
        assert fast.shape == slow.shape == (3, 1, 30, 26) and scale > 0
        assert float((fast - slow.detach()).abs().max()) <= 2e-6 * scale, storage
        assert float((fast_deg - fast).abs().max()) <= 2e-5 * scale
        assert float((via_pose - fast).abs().max()) <= 2e-5 * scale
    assert len(calls) == 6  # (both no-grad calls of every storage)


def check_euler_differentiable_path(device, ops=None, calls=None):
    """``drr(rot, xyz, parameterization="euler_angles")`` with pose parameters that take a gradient and the
    similarity computed outside the render (reference registration.py:32-42 with any criterion): the
    render's forward in three launches, its backward in ONE (DRR._render_euler_differentiable,
    renderers._EulerSiddonImageFn: ddrr_pose_raygen_forward with the clears, the brick kernel with its
    record, the image from the record | ddrr_siddon_backward_pose_euler) against the composition it
    replaces (euler_world_pose + render_poses: five and three launches; switched back on by
    FUSED_NCC_MAX_POSES = 0) -- images and the gradients of a random per-pixel weighting and of a criterion,
    in radians and degrees, two conventions, with and without stop_gradients_through_grid_sample, every
    brick storage.  The composition is itself pinned to the reference's fixtures (test_drr_module_golden,
    the registration trajectory)."""
    import torch

    from diffdrr_amd import DRR
    from diffdrr_amd.data import synthetic_subject
    from diffdrr_amd.metrics import MultiscaleNormalizedCrossCorrelation2d

    g = torch.Generator().manual_seed(21)
    for stop, storage, conv, deg, B in ((False, "f32", "ZXY", False, 3), (True, "q16p", "ZYX", True, 1),
                                        (False, "q16", "ZXY", False, 2)):
        drr = DRR(synthetic_subject(40, kind="phantom", seed=3), sdd=600.0, height=30, width=26, delx=3.0,
                  stop_gradients_through_grid_sample=stop).to(device)
        drr.renderer.brick_storage = storage
        rot0 = ((torch.rand(B, 3, generator=g) - 0.5) * 0.8)
        rot0 = (rot0 * 180 / torch.pi if deg else rot0).to(device)
        xyz0 = (torch.tensor([0.0, 400.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 20).to(device)
        w = torch.randn(B, 1, 30, 26, generator=g).to(device)
        with torch.no_grad():
            fixed = drr(rot0[:1] * 0, xyz0[:1], parameterization="euler_angles", convention=conv)
        crit = MultiscaleNormalizedCrossCorrelation2d([None, 7], [0.5, 0.5])
        res = []
        for cap in (DRR.FUSED_NCC_MAX_POSES, 0):
            drr.FUSED_NCC_MAX_POSES = cap
            n = len(calls) if calls is not None else 0
            rot, xyz = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
            img = drr(rot, xyz, parameterization="euler_angles", convention=conv, degrees=deg)
            loss = (img * w).sum() + 100.0 * crit(fixed.expand(B, -1, -1, -1), img).sum()
            loss.backward()
            res.append((img.detach(), rot.grad.clone(), xyz.grad.clone()))
            if calls is not None:
                took = [c for c in calls[n:] if c == "ddrr_siddon_backward_pose_euler"]
                assert len(took) == (1 if cap else 0), (cap, calls[n:])
        (i1, r1, x1), (i0, r0, x0) = res
        scale = float(i0.abs().max())
        assert i1.shape == i0.shape == (B, 1, 30, 26) and scale > 0
        assert float((i1 - i0).abs().max()) <= 2e-6 * scale, storage
        for a, b in ((r1, r0), (x1, x0)):
            assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-7, (storage, a, b)
    # what the route leaves alone: a volume that takes a gradient, more poses than the cap
    drr = DRR(synthetic_subject(24, kind="phantom", seed=3), sdd=600.0, height=16, delx=4.0).to(device)
    rot = torch.zeros(2, 3, device=device, requires_grad=True)
    xyz = torch.tensor([[0.0, 400.0, 0.0]] * 2, device=device)
    assert drr._render_euler_differentiable(rot, xyz, "ZXY", False) is not None
    drr.density.requires_grad_(True)
    assert drr._render_euler_differentiable(rot, xyz, "ZXY", False) is None
    drr.density.requires_grad_(False)
    with torch.no_grad():
        assert drr._render_euler_differentiable(rot, xyz, "ZXY", False) is None


def check_pose_adam(device):
    """``PoseAdam`` (ddrr_pose_adam_step: both pose parameter groups in one launch) against
    ``torch.optim.Adam`` with the same two groups -- the optimizer of the reference's registration
    loop (notebooks/tutorials/registration.ipynb:240-316) -- over a few steps of the same gradients:
    parameters and state, minimising and maximising, with a learning-rate change in between."""
    import torch

    from diffdrr_amd.registration import PoseAdam

    g = torch.Generator().manual_seed(11)
    for B, maximize in ((1, True), (5, False)):
        p0 = [torch.randn(B, 3, generator=g), torch.randn(B, 3, generator=g) * 50.0]
        mine = [torch.nn.Parameter(p.clone().to(device)) for p in p0]
        ref = [torch.nn.Parameter(p.clone()) for p in p0]
        opt = PoseAdam(mine[0], mine[1], 1e-1, 5e0, betas=(0.9, 0.999), eps=1e-8, maximize=maximize)
        ropt = torch.optim.Adam([{"params": [ref[0]], "lr": 1e-1}, {"params": [ref[1]], "lr": 5e0}],
                                maximize=maximize)
        for it in range(12):
            grads = [torch.randn(B, 3, generator=g) * (10.0 ** (it % 3 - 1)), torch.randn(B, 3, generator=g) * 1e-3]
            for p, r, gr in zip(mine, ref, grads):
                p.grad = gr.clone().to(device)
                r.grad = gr.clone()
            if it == 6:  # (a scheduler's doing)
                for o in (opt, ropt):
                    o.param_groups[0]["lr"] = 3e-2
            opt.step()
            ropt.step()
            # (a step moves a parameter by up to its learning rate: fp32 rounding of the quotient is
            # relative to THAT, and adds up over the steps)
            for p, r, lr in zip(mine, ref, (1e-1, 5e0)):
                assert torch.allclose(p.detach().cpu(), r.detach(), rtol=2e-5, atol=2e-5 * lr), (B, it)
        for p, r in zip(mine, ref):
            st, rst = opt.state[p], ropt.state[r]
            assert float(st["step"]) == float(rst["step"]) == 12.0
            assert torch.allclose(st["exp_avg"].cpu(), rst["exp_avg"], rtol=1e-4, atol=1e-9)
            assert torch.allclose(st["exp_avg_sq"].cpu(), rst["exp_avg_sq"], rtol=1e-4, atol=1e-12)
    # a parameter without a gradient is an error, not a silent no-op
    q = [torch.nn.Parameter(torch.zeros(1, 3, device=device)) for _ in range(2)]
    q[0].grad = torch.ones(1, 3, device=device)
    try:
        PoseAdam(q[0], q[1], 0.1, 1.0).step()
    except RuntimeError as e:
        assert "gradient" in str(e)
    else:
        raise AssertionError("PoseAdam.step() without a gradient did not raise")


def check_fused_ncc_step(device):
    """``DRR.ncc`` -- the registration step around the brick kernel as three fused launches
    (ddrr_pose_raygen_forward, ddrr_siddon_ncc_forward, ddrr_siddon_ncc_backward_pose; reference
    registration.py:32-42 + metrics.py:21-44) -- against the composition it replaces,
    ``NormalizedCrossCorrelation2d()(fixed, drr(rot, xyz, ...))``: values and the gradients of a
    weighted sum w.r.t. the pose parameters; a fixed image shared by the batch and one per pose;
    with and without the gradient path through the ray length; the workspace is left zero."""
    import torch

    from diffdrr_amd import DRR, NormalizedCrossCorrelation2d, ops
    from diffdrr_amd.data import synthetic_subject

    H, W = 28, 36
    for stop in (False, True):
        drr = DRR(synthetic_subject((40, 48, 36), kind="phantom", seed=3), sdd=500.0, height=H, width=W,
                  delx=2.5, stop_gradients_through_grid_sample=stop).to(device)
        rot0 = torch.tensor([[0.2, -0.1, 0.3], [0.0, 0.0, 0.0], [-0.4, 0.25, 0.1]], device=device)
        xyz0 = torch.tensor([[3.0, 300.0, -2.0], [0.0, 310.0, 0.0], [-4.0, 320.0, 5.0]], device=device)
        B = 3
        with torch.no_grad():
            base = drr(torch.zeros(1, 3, device=device), torch.tensor([[0.0, 305.0, 0.0]], device=device),
                       parameterization="euler_angles", convention="ZXY")
            per_pose = drr(rot0 + 0.05, xyz0 + 2.0, parameterization="euler_angles", convention="ZXY")
        w = torch.tensor([0.7, -1.3, 2.1], device=device)
        crit = NormalizedCrossCorrelation2d()
        v_base = None
        for fixed in (base, per_pose):
            res = []
            for fused in (True, False):
                r, x = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
                if fused:
                    val = drr.ncc(fixed, r, x, convention="ZXY")
                    assert type(val.grad_fn).__name__.startswith("_EulerSiddonNccFn")  # the fused route
                else:
                    val = crit(fixed.expand(B, -1, -1, -1),
                               drr(r, x, parameterization="euler_angles", convention="ZXY"))
                (val * w).sum().backward()
                res.append((val.detach().cpu().numpy(), r.grad.cpu().numpy(), x.grad.cpu().numpy()))
            (v1, gr1, gx1), (v0, gr0, gx0) = res
            assert v1.shape == (B,) and np.abs(v1 - v0).max() < 2e-6, (v1, v0)
            assert rel_err(gr1, gr0) < 2e-4 and rel_err(gx1, gx0) < 2e-4, (rel_err(gr1, gr0), rel_err(gx1, gx0))
            v_base = v0 if v_base is None else v_base
        if device != "cpu":  # (the host emulation does not touch the workspace)
            torch.cuda.synchronize()
            assert float(ops.siddon_ncc_workspace(B, device).abs().max()) == 0.0
        # reduction="sum": the batch's objective out of the epilogue's own launch (ABI 31 ncc_sum), its
        # gradient back in as one ready-made value -- the values and gradients of `.sum().backward()`
        ref_r, ref_x = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
        ref_v = drr.ncc(base, ref_r, ref_x, convention="ZXY")
        (ref_v.sum() * 1.7).backward()
        r, x = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
        tot = drr.ncc(base, r, x, convention="ZXY", reduction="sum")
        assert tot.dim() == 0 and type(tot.grad_fn).__name__.startswith("_EulerSiddonNccFn")
        assert abs(float(tot.detach()) - float(ref_v.detach().sum())) < 2e-6
        tot.backward(gradient=torch.tensor(1.7, device=device))
        assert rel_err(r.grad.cpu().numpy(), ref_r.grad.cpu().numpy()) < 2e-5
        assert rel_err(x.grad.cpu().numpy(), ref_x.grad.cpu().numpy()) < 2e-5
        # nothing to differentiate: the composition (forward-only kernel), same values
        with torch.no_grad():
            assert abs(float(drr.ncc(base, rot0, xyz0, reduction="sum")) - float(ref_v.detach().sum())) < 2e-6
            v = drr.ncc(base, rot0, xyz0)
        assert v.shape == (B,) and np.abs(v.cpu().numpy() - v_base).max() < 2e-6
        # the composed route with gradients: more poses than the fused step takes per call, and a
        # DRR(reshape=False), whose renders are (B, 1, N) (the fallback must hand the
        # criterion the detector's grid) -- same values, same gradients as the fused route
        flat = DRR(synthetic_subject((40, 48, 36), kind="phantom", seed=3), sdd=500.0, height=H, width=W,
                   delx=2.5, stop_gradients_through_grid_sample=stop, reshape=False).to(device)
        had = DRR.FUSED_NCC_MAX_POSES
        outs = []
        for d, cap in ((drr, had), (drr, 2), (flat, had), (flat, 2)):
            d.FUSED_NCC_MAX_POSES = cap
            r, x = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()
            val = d.ncc(base, r, x, convention="ZXY")
            assert type(val.grad_fn).__name__.startswith("_EulerSiddonNccFn") == (cap == had)
            (val * w).sum().backward()
            outs.append((val.detach().cpu().numpy(), r.grad.cpu().numpy(), x.grad.cpu().numpy()))
            d.FUSED_NCC_MAX_POSES = had
        for v_, gr_, gx_ in outs[1:]:
            assert np.abs(v_ - outs[0][0]).max() < 2e-6
            assert rel_err(gr_, outs[0][1]) < 2e-4 and rel_err(gx_, outs[0][2]) < 2e-4


SPARSE_CASES = {  # tests/golden/make_golden_sparse.py: name -> (renderer, DRR kwargs, call kwargs, poses)
    "siddon_sub": ("siddon", dict(p_subsample=0.3), {}, 1),
    "siddon_sub_flat": ("siddon", dict(p_subsample=0.3, reshape=False), {}, 2),
    "siddon_patch4": ("siddon", dict(patch_size=4), {}, 2),
    "siddon_patch4_channels": ("siddon", dict(patch_size=4), dict(mask_to_channels=True), 2),
    "trilinear_sub_flat": ("trilinear", dict(p_subsample=0.3, reshape=False), dict(n_points=40), 2),
    "trilinear_patch4": ("trilinear", dict(patch_size=4), dict(n_points=40), 2),
    "trilinear_patch5": ("trilinear", dict(patch_size=5), dict(n_points=40), 2),
    "trilinear_patch5_sub": ("trilinear", dict(patch_size=5, p_subsample=0.3), dict(n_points=40), 1),
    "trilinear_patch4_channels": ("trilinear", dict(patch_size=4), dict(n_points=40, mask_to_channels=True), 2),
    "trilinear_unpatched": ("trilinear", {}, dict(n_points=40), 2),
}


def check_sparse_lever(name, device, ops, calls=None):
    """``p_subsample`` / ``patch_size`` of ``DRR`` (reference drr.py:36-39, 142-147, 218-225) against
    the fixture of the UNMODIFIED reference (tests/golden/drr_sparse.npz): the same subsample drawn
    (``torch.manual_seed`` + ``randperm``), image and pose gradients -- and, asked for by VERDICT r05
    next 2, ON THE VOLUME-STATIONARY KERNELS: ``calls`` collects the C-ABI entries the render went
    through; the per-ray forward kernels must not be among them."""
    import torch

    from diffdrr_amd import DRR
    from diffdrr_amd.data import Image, Subject

    g = golden("drr_sparse")
    renderer, ctor, call, B = SPARSE_CASES[name]
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(device)  # noqa: E731
    vol = T(g["volume"])
    mask = Image(T(g["mask"]).unsqueeze(0), g["affine"])
    subject = Subject(Image(vol.unsqueeze(0), g["affine"]), Image(vol.unsqueeze(0), g["affine"]),
                      torch.from_numpy(g["reorient"]), mask)
    geo = {k[4:]: g[k].item() for k in g.files if k.startswith("geo_")}
    geo["height"], geo["width"] = int(geo["height"]), int(geo["width"])
    torch.manual_seed(int(g["seed"]))
    drr = DRR(subject, renderer=renderer, **geo, **ctor).to(device)
    if "p_subsample" in ctor:
        assert drr.detector.subsamples[-1] == g[f"{name}_subsample"].tolist()  # the reference's draw
    rot = T(g["rot"])[:B].clone().requires_grad_()
    xyz = T(g["xyz"])[:B].clone().requires_grad_()
    if calls is not None:
        calls.clear()
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", **call)
    assert img.shape == g[f"{name}_img_f32"].shape
    assert rel_err(img.detach().cpu().numpy(), g[f"{name}_img_f32"]) < 1e-4, name
    assert rel_err(img.detach().cpu().numpy(), g[f"{name}_img_f64"]) < 1e-4, name
    img.backward(T(g[f"{name}_grad_out_f32"]))
    # (the gradient gate of DESIGN section 4: no further from fp64 than 2 x the reference's own fp32
    # arithmetic + 1e-3 -- a noise volume, a dozen rays per chunk, ranges routed to arg-min rays)
    for mine, key in ((rot.grad, "g_rot"), (xyz.grad, "g_xyz")):
        own = rel_err(g[f"{name}_{key}_f32"], g[f"{name}_{key}_f64"])
        assert rel_err(mine.cpu().numpy(), g[f"{name}_{key}_f64"]) < 2 * own + 1e-3, (name, key, own)
    with torch.no_grad():  # the everyday call without gradients: the same image
        again = drr(rot.detach(), xyz.detach(), parameterization="euler_angles", convention="ZXY", **call)
    assert rel_err(again.cpu().numpy(), img.detach().cpu().numpy()) < 2e-6, name
    if calls is not None:
        per_ray = {"ddrr_siddon_forward", "ddrr_trilinear_forward", "ddrr_siddon_forward_channels",
                   "ddrr_trilinear_forward_channels", "ddrr_trilinear_backward", "ddrr_siddon_backward_channels"}
        assert not per_ray & set(calls), (name, sorted(per_ray & set(calls)))
        assert any("_bricks" in c for c in calls), (name, calls)
        if renderer == "siddon" and "p_subsample" in ctor:
            # the subsample goes through the kernels' pixel mask: the rays that were not drawn are
            # dropped after the candidate test, and what is left is the scattered image itself
            assert "ddrr_siddon_forward_bricks_masked" in calls, calls
            if ctor.get("reshape", True):
                drawn = torch.zeros(geo["height"] * geo["width"], dtype=torch.bool, device=device)
                drawn[torch.tensor(drr.detector.subsamples[-1], device=device)] = True
                assert float(img.detach().reshape(B, -1)[:, ~drawn].abs().max()) == 0.0
    return drr


def check_untracked_volume_edits(device, ops):
    """the default storage renders from a cache of 16-bit bricks
    keyed on what PyTorch tracks (the tensor's version counter, its storage's address).  Edits
    that bypass the counter (``volume.data.mul_(2)``) used to render the OLD bricks, silently.
    Now: the workspace carries a fingerprint of the volume it was built from and the launch
    compares it with the live volume -- a changed volume is rendered from its own fp32 values
    (correct, slower) until ``volume_changed()`` has the bricks rebuilt; ``volume.data = other``
    (a new storage) is seen by the host-side key.  The one thing left to the explicit call: an
    edit of a few voxels that misses all 1024 samples (documented here, not hidden)."""
    import torch

    from diffdrr_amd import DRR, Siddon
    from diffdrr_amd.data import make_subject

    vol = 0.2 + torch.rand(64, 64, 128, generator=torch.Generator().manual_seed(2))
    vol[:, :, :16] = 0.0  # (some air)
    drr = DRR(make_subject(vol, spacing=(1.0, 1.0, 1.0)), sdd=1000.0, height=40, delx=2.0).to(device)
    V = drr.density
    rot = torch.tensor([[0.2, -0.1, 0.3], [-0.4, 0.25, 0.1]], device=device)
    xyz = torch.tensor([[3.0, 500.0, -2.0], [-4.0, 520.0, 5.0]], device=device)
    from test_gpu_parity import voxel_rays
    s, t, L = voxel_rays(drr, rot, xyz)
    render = lambda: ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage="q16p")[0].clone()  # noqa: E731
    img0 = render()
    assert ops.brick_workspace(V, "q16p")[1] == 1 and ops.brick_workspace_stale(V, "q16p") == 0
    n_f32, n = ops.brick_fallbacks(V, "q16p")
    assert n_f32 < n  # (quantised bricks take part)
    # 1. an untracked in-place edit of the whole volume
    version = V._version
    V.data.mul_(2.0)
    assert V._version == version and ops.brick_workspace(V, "q16p")[1] == 1  # the host sees nothing ...
    img1 = render()
    assert rel_err(img1.cpu().numpy(), 2.0 * img0.cpu().numpy()) < 2e-5      # ... the launch does
    assert ops.brick_workspace_stale(V, "q16p") == 1
    render()
    assert ops.brick_workspace_stale(V, "q16p") == 2                          # (slow path until told)
    Siddon.volume_changed(V)
    assert ops.brick_workspace(V, "q16p")[1] == 0
    img2 = render()
    assert rel_err(img2.cpu().numpy(), 2.0 * img0.cpu().numpy()) < 2e-5
    assert ops.brick_workspace_stale(V, "q16p") == 0 and ops.brick_fallbacks(V, "q16p") == (n_f32, n)
    # through the module too
    with torch.no_grad():
        m0 = drr(rot, xyz, parameterization="euler_angles", convention="ZXY").clone()
        drr.density.data.mul_(0.5)
        m1 = drr(rot, xyz, parameterization="euler_angles", convention="ZXY").clone()
        drr.volume_changed()
        m2 = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    assert rel_err(m1.cpu().numpy(), 0.5 * m0.cpu().numpy()) < 2e-5
    assert rel_err(m2.cpu().numpy(), 0.5 * m0.cpu().numpy()) < 2e-5
    # 2. a new storage under the same tensor object: the host-side key
    img_before = render()
    V.data = (V.data * 3.0).clone()
    assert ops.brick_workspace(V, "q16p")[1] == 0
    assert rel_err(render().cpu().numpy(), 3.0 * img_before.cpu().numpy()) < 2e-5
    # 3. the documented limit: a few voxels between the samples (1024 samples, 524 288 voxels: one in
    # 512) -- an edit the fingerprint cannot see is rendered from the old bricks until volume_changed()
    img_before = render()
    flat = V.data.view(-1)
    n_vox = flat.numel()  # (csrc/brick_core.h fingerprint_index)
    cell = max(1, n_vox >> 10)
    mask = (1 << (cell.bit_length() - 1)) - 1
    sampled = {min(i * cell + ((((i * 2654435761 + 0x9e3779b9) & 0xffffffff) >> 4) & mask), n_vox - 1)
               for i in range(1024)}
    center = (32 * 64 + 32) * 128 + 64
    k = next(j for j in range(center, center + 200) if j not in sampled)
    flat[k] += 5.0
    unseen = render()
    assert ops.brick_workspace_stale(V, "q16p") == 0  # (not noticed: `unseen` shows the old bricks)
    if str(device) != "cpu":  # (the emulation stages from the live volume: it keeps no packed copy)
        assert (unseen - img_before).abs().max() < 1e-3
    Siddon.volume_changed(V)
    seen = render()
    assert (seen - img_before).abs().max() > 1e-3  # the brighter voxel shows after the explicit call
    assert ops.brick_workspace_stale(V, "q16p") == 0


def check_patch_ncc_against_composition(device):
    """ddrr_ncc_patch_forward / _backward (NormalizedCrossCorrelation2d(patch_size = p) without the
    reference's (B, windows, p, p) tensors; reference metrics.py:16-44) against the composition they
    replace -- `to_patches` + `norm` through autograd, itself pinned to the reference's fixture -- on
    what the fixture does not cover: non-square images, even / odd / extreme window sizes (1, the whole
    image), a fixed image shared by the batch (`expand`: read in place), two channels, a gradient
    w.r.t. the fixed image, `.sum()` (an expanded scalar gradient) and per-pose weights."""
    import copy

    import torch

    from diffdrr_amd import metrics as M
    from diffdrr_amd import ops

    calls = []
    fwd = ops.ncc_patch_forward
    ops.ncc_patch_forward = lambda *a, **k: (calls.append(1), fwd(*a, **k))[1]
    try:
        g = torch.Generator().manual_seed(4)
        # (p = 1 -- one-pixel windows: every z-score is 0 / sqrt(eps) -- has the value 0 and nothing to
        # differentiate but rounding residue times 1 / eps: value only)
        one = M.NormalizedCrossCorrelation2d(patch_size=1)(
            torch.rand(1, 1, 9, 12, generator=g).to(device), torch.rand(1, 1, 9, 12, generator=g).to(device))
        assert float(one.abs().max()) < 1e-6
        for (B, C, H, W), p in (((3, 1, 21, 34), 5), ((2, 1, 40, 33), 8), ((2, 2, 19, 23), 7), ((1, 1, 9, 12), 2),
                                ((2, 1, 12, 12), 12), ((2, 1, 50, 41), 33), ((1, 1, 70, 66), 64)):
            fixed = (torch.rand(1, C, H, W, generator=g) * 20 + 5).to(device)
            moving = (torch.rand(B, C, H, W, generator=g) * 20 + 5).to(device)
            w = (torch.rand(B, generator=g) + 0.5).to(device)
            crit = M.NormalizedCrossCorrelation2d(patch_size=p)
            ref = copy.deepcopy(crit)
            ref._no_patch_kernel = True
            res = []
            for c, grad_fixed in ((crit, True), (ref, True)):
                a = fixed.clone().requires_grad_(grad_fixed)
                x = moving.clone().requires_grad_()
                v = c(a.expand(B, -1, -1, -1), x)
                (v * w).sum().backward()
                res.append((v.detach().cpu().numpy(), x.grad.cpu().numpy(), a.grad.cpu().numpy()))
            (v1, g1, ga1), (v0, g0, ga0) = res
            assert np.abs(v1 - v0).max() < 5e-6, (p, v1, v0)
            assert rel_err(g1, g0) < 5e-5 and rel_err(ga1, ga0) < 5e-5, (p, rel_err(g1, g0), rel_err(ga1, ga0))
            # `.sum()`: the gradient arrives as an expanded scalar
            x = moving.clone().requires_grad_()
            crit(fixed.expand(B, -1, -1, -1), x).sum().backward()
            x0 = moving.clone().requires_grad_()
            ref(fixed.expand(B, -1, -1, -1), x0).sum().backward()
            assert rel_err(x.grad.cpu().numpy(), x0.grad.cpu().numpy()) < 5e-5
        assert len(calls) >= 14  # (the kernels really ran)
    finally:
        ops.ncc_patch_forward = fwd


def check_blur_sobel_against_composition(device):
    """ddrr_blur_sobel_forward / _backward (Sobel(sigma > 0): torchvision's gaussian_blur + the 3 x 3 pair in
    one launch each way; reference metrics.py:66, 88-93) against the composition they replace -- reflect pad,
    k x k depthwise conv2d, Sobel kernel, through autograd; itself pinned to the reference's fixture
    (`gncc_sigma1`) -- on what the fixture does not cover: image sizes that are not multiples of the kernels'
    32 x 32 tile, images barely larger than the padding (k // 2 = min(H, W) - 1: every pixel is folded),
    3 ... 31 taps, a fixed image shared by the batch (`expand`: read in place).  33 taps take the composition."""
    import copy

    import torch

    from diffdrr_amd import metrics as M
    from diffdrr_amd import ops

    calls = []
    fwd = ops.blur_sobel_forward
    ops.blur_sobel_forward = lambda *a, **k: (calls.append(1), fwd(*a, **k))[1]
    try:
        g = torch.Generator().manual_seed(5)
        for (B, H, W), sigma in (((2, 45, 70), 1.0), ((3, 32, 64), 0.4), ((1, 4, 5), 1.0), ((2, 16, 33), 5.0),
                                 ((2, 100, 37), 2.3), ((2, 70, 70), 5.3)):
            img = (torch.rand(B, 1, H, W, generator=g) * 30).to(device)
            w = torch.randn(B, 2, H, W, generator=g).to(device)
            sob = M.Sobel(sigma)
            ref = copy.deepcopy(sob)
            ref._no_blur_kernel = True
            n = len(calls)
            res = []
            for m in (sob, ref):
                x = img.clone().requires_grad_()
                out = m(x)
                (out * w).sum().backward()
                res.append((out.detach().cpu().numpy(), x.grad.cpu().numpy()))
            assert (len(calls) > n) == ((int(6 * sigma + 1) | 1) <= 31), sigma
            assert rel_err(res[0][0], res[1][0]) < 2e-6, (sigma, rel_err(res[0][0], res[1][0]))
            assert rel_err(res[0][1], res[1][1]) < 2e-6, (sigma, rel_err(res[0][1], res[1][1]))
            # one image for the whole batch
            one = img[:1].clone().requires_grad_()
            out = sob(one.expand(3, -1, -1, -1))
            (out * w[:1]).sum().backward()
            assert rel_err(out[2].detach().cpu().numpy(), res[1][0][0]) < 2e-6
            assert rel_err(one.grad.cpu().numpy()[0], 3 * res[1][1][0]) < 1e-5
        # the criterion end to end
        a = (torch.rand(2, 1, 40, 52, generator=g) * 30).to(device)
        b = (torch.rand(2, 1, 40, 52, generator=g) * 30).to(device)
        for patch in (None, 9):
            crit = M.GradientNormalizedCrossCorrelation2d(patch_size=patch, sigma=1.0)
            ref = copy.deepcopy(crit)
            ref.sobel._no_blur_kernel = True
            res = []
            for c in (crit, ref):
                x = b.clone().requires_grad_()
                v = c(a, x)
                v.sum().backward()
                res.append((v.detach().cpu().numpy(), x.grad.cpu().numpy()))
            assert np.abs(res[0][0] - res[1][0]).max() < 2e-6
            assert rel_err(res[0][1], res[1][1]) < 2e-5, rel_err(res[0][1], res[1][1])
    finally:
        ops.blur_sobel_forward = fwd


def check_channel_words(device, ops):
    """mask_to_channels from the volume's ready-packed words (ops.channel_words, ddrr_channel_words +
    ddrr_siddon_forward_channels_bricks_words; reference renderers.py:77-89): the first render of a
    (volume, label map) pair stages from both, the second packs the words, later ones stage them as
    they are -- same images --, a tracked edit forces a repack, an UNTRACKED one (`density.data.mul_`)
    is found by the call's own fingerprint comparison on the device and repacked too: never old
    words."""
    import torch

    from diffdrr_amd import DRR
    from diffdrr_amd.data import synthetic_subject
    from diffdrr_amd.renderers import _labels_u8

    drr = DRR(synthetic_subject((40, 70, 36), kind="noise", seed=5, n_labels=9), sdd=600.0, height=24, width=31,
              delx=3.0).to(device)
    rot = torch.tensor([[0.3, 0.2, -0.1], [1.5, 0.1, 0.0]], device=device)
    xyz = torch.tensor([[5.0, 420.0, -3.0], [0.0, 400.0, 0.0]], device=device)
    calls = []
    launch = ops._launch
    ops._launch = lambda n, d, *a: (calls.append(n), launch(n, d, *a))[1]
    try:
        def render():
            calls.clear()
            with torch.no_grad():
                return drr(rot, xyz, parameterization="euler_angles", convention="ZXY", mask_to_channels=True).clone()
        labels, C, _ = _labels_u8(drr.mask)[0]
        first = render()
        assert "ddrr_siddon_forward_channels_bricks" in calls and "ddrr_channel_words" not in calls
        second = render()
        assert "ddrr_channel_words" in calls and "ddrr_siddon_forward_channels_bricks_words" in calls
        assert ops.channel_words_repacks(drr.density, labels, C) == 1
        third = render()
        assert ops.channel_words_repacks(drr.density, labels, C) == 1           # (compared, not packed again)
        scale = float(first.abs().max())
        assert float((second - first).abs().max()) <= 2e-6 * scale and float((third - first).abs().max()) <= 2e-6 * scale
        with torch.no_grad():
            drr.density[3, 4, 5] += 0.5                                          # tracked: the version counter
        fourth = render()
        assert ops.channel_words_repacks(drr.density, labels, C) == 2
        drr.renderer.channel_words = False
        assert float((fourth - render()).abs().max()) <= 2e-6 * scale            # = staged from volume + labels
        drr.renderer.channel_words = True
        drr.density.data.mul_(2.0)                                               # untracked
        fifth = render()
        assert ops.channel_words_repacks(drr.density, labels, C) == 3            # found on the device
        assert rel_err(fifth.cpu().numpy(), 2.0 * fourth.cpu().numpy()) < 2e-5
    finally:
        ops._launch = launch
