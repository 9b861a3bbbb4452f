"""Fixture for ``Siddon(filter_intersections_outside_volume=True)``.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_filter.py

The UNMODIFIED reference cannot execute this branch: ``_filter_intersections_outside_volume``
(diffdrr/renderers.py:116-121) calls ``_get_alpha_minmax(source, target, dims, eps)`` while the
function takes ``(source, target, dims, voxel_shift, eps)`` (:124) -- a ``TypeError`` on every
call.  This script records that fact (``raises_type_error``) and then pins the branch's INTENDED
semantics: the same reference with that one call given its missing argument (nothing else is
touched -- the rest of the branch, the sort, the lookup and autograd are the reference's own).
The filter drops the sorted-crossing columns that lie outside [alphamin, alphamax] for EVERY ray;
segments outside the volume add nothing (zero padding), so with both ray endpoints outside the
volume the image and every gradient equal the default render's -- which is what the product
implements (the keyword is accepted and the whole line is integrated) and what
tests/test_host_api.py / tests/test_gpu_parity.py check against this fixture.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import ref_loader  # noqa: E402
import make_golden as MG  # noqa: E402  (ray sets and the runner; importing it writes nothing)

ref = ref_loader.load()
R = ref.renderers


def main():
    dims = (12, 10, 14)
    g = torch.Generator().manual_seed(60)
    volume = torch.rand(*dims, generator=g)
    src_a, tgt_a = MG.random_rays(g, dims, 2, 40)
    src_b, tgt_b = MG.oblique_rays(g, dims, 2, 40)
    src, tgt = torch.cat([src_a, src_b]), torch.cat([tgt_a, tgt_b])

    # 1. the unmodified reference raises
    raised = False
    try:
        img = (tgt - src).norm(dim=-1).unsqueeze(1)
        R.Siddon(filter_intersections_outside_volume=True)(volume, src, tgt, img)
    except TypeError:
        raised = True
    assert raised, "the reference's filter branch ran: regenerate this fixture's rationale"

    # 2. the branch with its call completed (voxel_shift: the module's default 0.5)
    original = R._filter_intersections_outside_volume

    def completed(alphas, source, target, dims_, eps):
        alphamin, alphamax = R._get_alpha_minmax(source, target, dims_, 0.5, eps)
        good_idxs = torch.logical_and(alphamin <= alphas, alphas <= alphamax)
        return alphas[..., good_idxs.any(dim=[0, 1])]

    arrays = {"volume": MG.npy(volume), "source": MG.npy(src), "target": MG.npy(tgt),
              "raises_type_error": np.asarray(raised)}
    R._filter_intersections_outside_volume = completed
    try:
        for flag, key in ((True, "filtered"), (False, "default")):
            for dtype, tag in ((MG.F32, "f32"), (MG.F64, "f64")):
                gg = torch.Generator().manual_seed(61)
                res = MG.run_renderer(
                    lambda: R.Siddon(filter_intersections_outside_volume=flag), {}, volume, src, tgt,
                    dtype, gg)
                for k, v in res.items():
                    if v is not None:
                        arrays[f"{key}_{k}_{tag}"] = v
    finally:
        R._filter_intersections_outside_volume = original
    # the intended semantics leave the render unchanged (fp64: to rounding of the shorter sums)
    for k in ("out", "g_source", "g_target", "g_img", "g_volume"):
        a, b = arrays[f"filtered_{k}_f64"], arrays[f"default_{k}_f64"]
        err = np.abs(a - b).max() / np.abs(b).max()
        print(f"filtered vs default, {k}: {err:.2e}")
        assert err < 1e-12, (k, err)
    MG.save("siddon_filter_outside", **arrays)


if __name__ == "__main__":
    main()
