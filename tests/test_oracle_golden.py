"""The C oracle (oracle/drr_oracle.c) against the golden vectors produced by the
unmodified reference (tests/golden/make_golden.py).  This is what pins the
oracle; the fp64 build must agree to rounding, the fp32 build to fp32 rounding."""
import numpy as np
import pytest

import oracle
from conftest import golden, rel_err

SIDDON_CASES = [
    ("siddon_sum", {}),
    ("siddon_sum_oblique", {}),
    ("siddon_max", {"reducefn": "max"}),
    ("siddon_per_ray_source", {}),
    ("siddon_shift0", {"voxel_shift": 0.0}),
    ("siddon_stopgrad", {}),
]


def _inputs(g, tag):
    dt = np.float32 if tag == "f32" else np.float64
    vol, src, tgt = (g[k].astype(dt) for k in ("volume", "source", "target"))
    B, N, _ = tgt.shape
    return vol, src, tgt, g["img_" + tag].reshape(B, N).astype(dt)


@pytest.mark.parametrize("name,kw", SIDDON_CASES)
@pytest.mark.parametrize("tag,tol,gtol", [("f64", 1e-12, 1e-10), ("f32", 5e-6, 2e-4)])
def test_siddon_forward_backward(name, kw, tag, tol, gtol):
    g = golden(name)
    vol, src, tgt, img = _inputs(g, tag)
    res = oracle.siddon(vol, src, tgt, img, grad_out=g["grad_out_" + tag],
                        want_volume_grad=True, **kw)
    assert rel_err(res["out"], g["out_" + tag]) < tol
    assert rel_err(res["g_source"], g["g_source_" + tag]) < gtol
    assert rel_err(res["g_target"], g["g_target_" + tag]) < gtol
    if name != "siddon_stopgrad":  # the flag drops these two gradients (renderers.py:63-65)
        assert rel_err(res["g_img"], g["g_img_" + tag]) < gtol
        assert rel_err(res["g_volume"], g["g_volume_" + tag]) < gtol
    else:
        assert "g_img_" + tag not in g.files and "g_volume_" + tag not in g.files


def test_siddon_special_rays():
    """Axis-parallel rays (eps path), misses, source/target inside the volume.
    The reference's own fp32 result is unreliable here (eps = 1e-8 vanishes or
    explodes in fp32), so the pin is the fp64 run."""
    g = golden("siddon_special_rays")
    vol, src, tgt, img = _inputs(g, "f64")
    out = oracle.siddon(vol, src, tgt, img)["out"]
    assert rel_err(out, g["out_f64"]) < 1e-12
    ref = g["out_f64"].reshape(-1)
    assert ref[5] == 0 and ref[6] == 0  # the two misses
    assert (ref[[0, 1, 2, 3, 4, 7, 8, 9, 10]] > 0).all()


@pytest.mark.parametrize("name,kw", [
    ("siddon_bilinear", {"mode": "bilinear"}),
    ("siddon_align_corners", {"align_corners": True}),
])
def test_siddon_generic_lookup(name, kw):
    g = golden(name)
    vol, src, tgt, img = _inputs(g, "f64")
    assert rel_err(oracle.siddon(vol, src, tgt, img, **kw)["out"], g["out_f64"]) < 1e-12
    vol, src, tgt, img = _inputs(g, "f32")
    assert rel_err(oracle.siddon(vol, src, tgt, img, **kw)["out"], g["out_f32"]) < 5e-6


def test_siddon_mask_channels():
    g = golden("siddon_mask")
    vol, src, tgt, img = _inputs(g, "f64")
    ch = oracle.siddon_channels(vol, g["mask"].astype(np.float64), src, tgt, img)
    assert ch.shape == g["out_f64"].shape
    assert rel_err(ch, g["out_f64"]) < 1e-12
    # "summing over the channel dimension recapitulates the original DRR"
    # (reference introduction.ipynb:286)
    assert rel_err(ch.sum(1, keepdims=True), oracle.siddon(vol, src, tgt, img)["out"]) < 1e-12


def test_siddon_mask_channel_gradients():
    """The oracle's channel-wise gradients against the reference's autograd through its
    scatter_add (fixture generated from the unmodified reference)."""
    g = golden("siddon_mask")
    vol, src, tgt, img = _inputs(g, "f64")
    r = oracle.siddon_channels_grad(vol, g["mask"], src, tgt, img, g["grad_out_f64"])
    for k in ("g_source", "g_target", "g_img", "g_volume"):
        assert rel_err(r[k], g[k + "_f64"]) < 1e-10, k


TRI_CASES = [
    ("trilinear_global_range", dict(n_points=41), None),
    ("trilinear_explicit_range", dict(n_points=64), (0.31, 0.77)),
    ("trilinear_oblique", dict(n_points=50), None),
    ("trilinear_shift0", dict(n_points=40, voxel_shift=0.0), None),
]


@pytest.mark.parametrize("name,kw,rng", TRI_CASES)
@pytest.mark.parametrize("tag,tol,gtol", [("f64", 1e-12, 1e-9), ("f32", 5e-6, 5e-4)])
def test_trilinear_forward_backward(name, kw, rng, tag, tol, gtol):
    g = golden(name)
    vol, src, tgt, img = _inputs(g, tag)
    extra = {} if rng is None else {"alphamin": rng[0], "alphamax": rng[1]}
    res = oracle.trilinear(vol, src, tgt, img, grad_out=g["grad_out_" + tag],
                           want_volume_grad=True, **kw, **extra)
    assert rel_err(res["out"], g["out_" + tag]) < tol
    assert rel_err(res["g_img"], g["g_img_" + tag]) < gtol
    assert rel_err(res["g_volume"], g["g_volume_" + tag]) < gtol
    if rng is not None:
        # alphamin / alphamax were leaves: every gradient is directly comparable
        assert rel_err(res["g_source"], g["g_source_" + tag]) < gtol
        assert rel_err(res["g_target"], g["g_target_" + tag]) < gtol
        assert abs(res["g_alphamin"] / g["g_alphamin_" + tag] - 1) < gtol
        assert abs(res["g_alphamax"] / g["g_alphamax_" + tag] - 1) < gtol


def test_trilinear_nearest_max_and_alpha_range():
    g = golden("trilinear_nearest_max")
    vol, src, tgt, img = _inputs(g, "f64")
    res = oracle.trilinear(vol, src, tgt, img, n_points=33, mode="nearest", reducefn="max")
    assert rel_err(res["out"], g["out_f64"]) < 1e-12


def test_trilinear_mask_channel_sum():
    """Trilinear with a mask: channels must add up to the plain render."""
    g = golden("trilinear_mask")
    vol, src, tgt, img = _inputs(g, "f64")
    plain = oracle.trilinear(vol, src, tgt, img, n_points=40)["out"]
    assert rel_err(g["out_f64"].sum(1, keepdims=True), plain) < 1e-12


def test_voxel_count_matches_segments():
    g = golden("siddon_sum")
    vol, src, tgt, img = _inputs(g, "f64")
    res = oracle.siddon(vol, src, tgt, img, count_voxels=True)
    terms, _ = oracle.siddon_segments(vol, src, tgt, img)
    assert (res["n_inside"] >= (terms != 0).sum(-1)).all()
    assert res["n_inside"].max() <= sum(vol.shape)
