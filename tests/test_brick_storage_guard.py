"""The 16-bit block-quantised brick storages of ddrr_siddon_forward_bricks ("q16" / "q16p", the
default of diffdrr_amd.Siddon) must stay within the 1e-4 of the reference on ANY volume: the
reference gathers the volume's own fp32 values (diffdrr/renderers.py:159-164).  A brick whose
range is large against its level -- a bright voxel among dim ones, un-normalised HU with metal,
inf / NaN -- is rendered from its fp32 values in the same launch (csrc/brick_step.h q16_usable,
csrc/bricks_fwd.hip MIXED).  Checked against the fp64 oracle on the host emulation here and on
the device in tests/test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import conftest

NAMES = sorted(conftest.guard_volumes())


@pytest.mark.parametrize("name", NAMES)
def test_q16_storage_keeps_the_tolerance_on_any_volume(emulated_ops, name):
    """(the emulation walks 32^3 bricks; same q16_* arithmetic and the same guard)"""
    n_f32, n = conftest.check_brick_storage_guard(emulated_ops, "cpu", name, "q16", (32, 32, 32))
    if name == "noise":
        assert n_f32 == 0  # what the storage is meant for stays on the fast path
    if "x200" in name or "x1e+06" in name or name == "hu_with_metal":
        assert n_f32 > 0


def test_guard_restatement_against_the_judges_table():
    """0.5 +- 0.005 body, one voxel raised: from 20x on the brick holding it
    must leave the quantised path (2x stays)."""
    vols = conftest.guard_volumes()
    for f, want in ((2, 0), (20, 1), (200, 1), (2000, 1), (1e4, 1), (1e6, 1)):
        _, flags = conftest.brick_levels(vols[f"outlier_x{f:g}_in_view"], (32, 32, 64))
        assert int(flags.sum()) == want, f


def test_workspace_is_valid_only_after_a_launch_that_built_it(emulated_ops, monkeypatch):
    """an empty batch returns before anything fills the workspace, and so does a
    launch that fails -- neither may leave it marked as built (the emulation, like the product,
    trusts a workspace handed over as valid)."""
    import oracle

    ops = emulated_ops
    vol = conftest.guard_volumes()["noise"]
    s, t, L = conftest.guard_scene("cpu")
    V = torch.from_numpy(vol)
    ref = oracle.siddon(vol, s.numpy(), t.numpy(), L.numpy())["out"].reshape(L.shape)
    # empty batch first
    out, _ = ops.siddon_forward_bricks(V, s[:0], t[:0], L[:0], (40, 40), storage="q16")
    assert out.shape == (0, 1600) and ops.brick_workspace(V, "q16")[1] == 0
    assert ops.brick_fallbacks(V, "q16") is None
    # a launch that fails before it does anything
    real = ops._launch

    def failing(name, *a):
        raise RuntimeError("12 * B * N >= 2^32: split the pose batch")

    monkeypatch.setattr(ops, "_launch", failing)
    with pytest.raises(RuntimeError):
        ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage="q16")
    monkeypatch.setattr(ops, "_launch", real)
    assert ops.brick_workspace(V, "q16")[1] == 0
    # garbage in the unbuilt workspace must not matter: the retry builds it
    ops.brick_workspace(V, "q16")[0].fill_(float("nan"))
    out, _ = ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage="q16")
    assert conftest.rel_err(out.numpy(), ref) < 5e-5
    assert ops.brick_workspace(V, "q16")[1] == 1 and ops.brick_fallbacks(V, "q16") == (0, 16)
    # ... and once built it is what the next launch renders from
    out2, _ = ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage="q16")
    assert np.array_equal(out2.numpy(), out.numpy())


def test_non_finite_voxels_take_the_fp32_path(emulated_ops):
    ops = emulated_ops
    vol = conftest.guard_volumes()["noise"].copy()
    vol[10, 10, 10] = np.inf
    vol[50, 50, 100] = np.nan
    s, t, L = conftest.guard_scene("cpu")
    V = torch.from_numpy(vol)
    q, _ = ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage="q16")
    f, _ = ops.siddon_forward_bricks(V, s, t, L, (40, 40), storage="f32")
    assert ops.brick_fallbacks(V, "q16") == (2, 16)
    q, f = q.numpy(), f.numpy()
    assert np.array_equal(np.isnan(q), np.isnan(f)) and np.array_equal(np.isinf(q), np.isinf(f))
    ok = np.isfinite(f)
    assert np.abs(q[ok] - f[ok]).max() <= 1e-5 * np.abs(f[ok]).max()


def _noisy_ct_like(seed, lung_sigma, tissue_sigma, dims=(256, 256, 67)):
    from diffdrr_amd.data import ct_like_hu_volume, transform_hu_to_density

    hu = ct_like_hu_volume(dims, seed=seed)
    g = torch.Generator().manual_seed(100 + seed)
    lung = (hu > -1000.0) & (hu < -600.0)
    tissue = (hu > -200.0) & (hu < 200.0)
    hu = torch.where(lung, -850.0 + (hu + 850.0) * lung_sigma, hu)
    hu = torch.where(tissue, hu + 15.0 * (tissue_sigma - 1.0) * torch.randn(hu.shape, generator=g), hu)
    return transform_hu_to_density(hu)


@pytest.mark.parametrize("seed,lung_sigma,tissue_sigma", [(0, 1.0, 1.0), (1, 3.0, 1.0), (2, 1.0, 4.0), (3, 5.0, 6.0),
                                                          (11, 0.5, 0.5), (16, 6.0, 6.0)])
def test_quantisation_error_per_pixel_on_noisy_ct_like_volumes(seed, lung_sigma, tissue_sigma):
    """the guard admits a brick to the 16-bit path when its range is <= 12x its LEVEL, the
    smallest MEAN |V| of a 4^3 block -- a statement about block means, while a ray crosses voxels:
    "a ray through the dim voxels of a noisy block can exceed it".  It can, and this test says by
    how much.  The quantisation's own share of a pixel's error, isolated from all fp32 arithmetic:
    the fp64 oracle on the volume and on the volume as the 16-bit bricks hold it (numpy
    restatement of brick_range_kernel / q16_usable / q16_encode: per 32 x 32 x 64 brick
    v' = min + rint((v - min) / step) step, flagged bricks untouched), CT-like volumes through
    transform_hu_to_density with the lung texture and the soft-tissue noise scaled 0.5x ... 6x,
    eight oblique poses each (a 20-volume sweep of the same kind chose the two worst seeds):
      * what the guard GUARANTEES holds at every pixel: |error| <= 12 / 131070 of the line integral
        of the brick LEVELS along the ray (worst ratio seen 0.94);
      * the image-normalised error -- the gate SURVEY 8(d) / the north star state -- is <= 3e-6;
      * relative to the PIXEL'S OWN value the guarantee is weaker than 1e-4 wherever a ray crosses
        mostly air and a sliver of skin inside tissue-level bricks: measured <= 1.6e-4 at pixels
        above 1e-3 of the image's maximum (the fp32 reference's own arithmetic is off by as much
        there: printed), up to 5e-3 at pixels above 1e-5 of it.  Held here to 3e-4 / 1e-2 so that
        a change of the guard's constant or statistic that worsens it is seen."""
    import oracle
    from diffdrr_amd import DRR, convert
    from diffdrr_amd.data import make_subject

    vol = _noisy_ct_like(seed, lung_sigma, tissue_sigma)
    v = vol.numpy()
    level_vol, flags = conftest.brick_levels(v, (32, 32, 64))
    vq = v.astype(np.float64).copy()
    for bx, by, bz in zip(*np.nonzero(~flags)):
        sl = (slice(bx * 32, bx * 32 + 32), slice(by * 32, by * 32 + 32), slice(bz * 64, bz * 64 + 64))
        blk = v[sl]
        lo, hi = np.float32(blk.min()), np.float32(blk.max())
        if hi > lo:
            step = np.float32(hi - lo) / np.float32(65535.0)
            q = np.rint((blk - lo).astype(np.float32) * (np.float32(1.0) / step))
            vq[sl] = np.float64(lo) + q.astype(np.float64) * np.float64(step)
    assert 0 < int(flags.sum()) < flags.size                  # both paths take part
    assert np.abs(vq - v).max() > 0 and (vq[v == 0] == 0).all()  # air stays exact
    drr = DRR(make_subject(vol, spacing=(1.0, 1.0, 1.0)), sdd=1200.0, height=96, width=96, delx=3.2)
    g = torch.Generator().manual_seed(seed)
    rot = (torch.rand(8, 3, generator=g) - 0.5) * 1.6
    xyz = torch.tensor([0.0, 600.0, 0.0]) + (torch.rand(8, 3, generator=g) - 0.5) * 40
    with torch.no_grad():
        src, tgt = drr.detector(convert(rot, xyz, parameterization="euler_angles", convention="ZXY"), None)
        L = (tgt - src).norm(dim=-1).numpy().astype(np.float64)
        s, t = (drr.affine_inverse(x).numpy().astype(np.float64) for x in (src, tgt))
    exact = oracle.siddon(v.astype(np.float64), s, t, L)["out"].reshape(-1)
    stored = oracle.siddon(vq, s, t, L)["out"].reshape(-1)
    ref32 = oracle.siddon(v, s.astype(np.float32), t.astype(np.float32), L.astype(np.float32))["out"].reshape(-1)
    bound = (conftest.Q16_RANGE_OVER_LEVEL / 131070.0) * oracle.siddon(level_vol, s, t, L)["out"].reshape(-1)
    err = np.abs(stored - exact)
    assert (err <= bound + 1e-30).all(), float((err / (bound + 1e-30)).max())  # the guarantee, as stated
    assert err.max() <= 3e-6 * exact.max()                                     # the gate, as stated
    figures = []
    for floor, cap in ((1e-3, 3e-4), (1e-5, 1e-2)):
        lit = exact > floor * exact.max()
        assert lit.mean() > 0.3
        per_pixel = float((err[lit] / exact[lit]).max())
        ref_pp = float((np.abs(ref32 - exact)[lit] / exact[lit]).max())
        figures.append(f"pixels > {floor:g} of the maximum: {per_pixel:.2e} (fp32 reference: {ref_pp:.2e})")
        assert per_pixel <= cap, (floor, per_pixel)
    print(f"[guard, seed {seed}, lung x{lung_sigma}, tissue x{tissue_sigma}] {int(flags.sum())} of {flags.size} "
          f"bricks on the fp32 path; quantisation error: image-normalised {err.max() / exact.max():.2e}, "
          f"at most {float((err / (bound + 1e-30)).max()):.2f} of the guaranteed bound; relative to the pixel's own "
          f"value, " + "; ".join(figures))


def test_untracked_volume_edits_are_rendered_from_the_live_values(emulated_ops):
    """(host emulation; the device twin: tests/test_gpu_brick_storage.py)"""
    conftest.check_untracked_volume_edits("cpu", emulated_ops)
