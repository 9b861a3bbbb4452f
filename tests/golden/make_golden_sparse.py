"""Golden fixture for the reference's own speed levers -- ``p_subsample`` (drr.py:36-39, 142-147;
detector.py:134-137) and ``patch_size`` (drr.py:218-225) -- from the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_sparse.py        ->  tests/golden/drr_sparse.npz

What it pins (fp32 and fp64 module, images and autograd gradients w.r.t. the Euler pose
parameters for a stored ``grad_out``):
  * a subsample is drawn once per detector by ``torch.randperm(H W)[:n]`` and the renderer's
    output is in THAT order (``subsample``: the indices; ``reshape=True`` scatters them into zeros);
  * ``patch_size`` cuts the rendered rays into ``target.chunk(n_patches, dim=1)`` -- consecutive
    rays, not 2-D patches, ragged when ``n_patches`` does not divide them -- and the marcher takes
    its marching range (renderers.py:220-223) over the rays of EACH chunk, so a patched trilinear
    render differs from the unpatched one; Siddon's does not;
  * both together; ``mask_to_channels`` under ``patch_size``.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import F32, F64, npy, ref, save, synthetic_subject  # noqa: E402

SEED = 1234  # torch.manual_seed before every DRR construction: the subsample's draw


def main():
    dims, spacing = (24, 20, 36), (1.5, 2.0, 1.25)
    subject, vol, affine = synthetic_subject(dims, spacing, 50, "AP", with_mask=True)
    geo = dict(sdd=300.0, height=12, width=10, delx=2.4, dely=2.1, x0=2.0, y0=-1.0)
    rot = torch.tensor([[0.1, -0.2, 0.15], [-0.5, 0.4, 0.25]])
    xyz = torch.tensor([[4.0, 190.0, -3.0], [-8.0, 230.0, 6.0]])
    arrays = {"volume": npy(vol), "affine": affine, "reorient": npy(subject.reorient),
              "mask": npy(subject.mask.data.squeeze()), "rot": npy(rot), "xyz": npy(xyz),
              "seed": np.asarray(SEED), **{"geo_" + k: np.asarray(v) for k, v in geo.items()}}
    # name -> (renderer, DRR kwargs, call kwargs, poses used)
    cases = {
        "siddon_sub": ("siddon", dict(p_subsample=0.3), {}, 1),
        "siddon_sub_flat": ("siddon", dict(p_subsample=0.3, reshape=False), {}, 2),
        "siddon_patch4": ("siddon", dict(patch_size=4), {}, 2),
        "siddon_patch4_channels": ("siddon", dict(patch_size=4), dict(mask_to_channels=True), 2),
        "trilinear_sub_flat": ("trilinear", dict(p_subsample=0.3, reshape=False), dict(n_points=40), 2),
        "trilinear_patch4": ("trilinear", dict(patch_size=4), dict(n_points=40), 2),       # 7 ragged chunks
        "trilinear_patch5": ("trilinear", dict(patch_size=5), dict(n_points=40), 2),       # 4 chunks of 3 rows
        "trilinear_patch5_sub": ("trilinear", dict(patch_size=5, p_subsample=0.3), dict(n_points=40), 1),
        "trilinear_patch4_channels": ("trilinear", dict(patch_size=4), dict(n_points=40, mask_to_channels=True), 2),
        "trilinear_unpatched": ("trilinear", {}, dict(n_points=40), 2),                    # (differs from patch4 / patch5)
    }
    for name, (renderer, ctor, call, B) in cases.items():
        for dtype, tag in ((F32, "f32"), (F64, "f64")):
            torch.manual_seed(SEED)
            drr = ref.DRR(subject, renderer=renderer, **geo, **ctor).to(dtype)
            r = rot[:B].to(dtype).clone().requires_grad_()
            x = xyz[:B].to(dtype).clone().requires_grad_()
            img = drr(r, x, parameterization="euler_angles", convention="ZXY", **call)
            go = torch.randn(img.shape, generator=torch.Generator().manual_seed(61)).to(dtype)
            gr, gx = torch.autograd.grad(img, (r, x), go)
            arrays.update({f"{name}_img_{tag}": npy(img), f"{name}_grad_out_{tag}": npy(go),
                           f"{name}_g_rot_{tag}": npy(gr), f"{name}_g_xyz_{tag}": npy(gx)})
            if "p_subsample" in ctor:
                arrays[f"{name}_subsample"] = np.asarray(drr.detector.subsamples[-1], np.int64)
                arrays[f"{name}_n_patches"] = np.asarray(drr.n_patches if "patch_size" in ctor else 0)
    # the chunks really see different marching ranges
    assert np.abs(arrays["trilinear_patch4_img_f64"] - arrays["trilinear_unpatched_img_f64"]).max() > 1e-6
    save("drr_sparse", **arrays)


if __name__ == "__main__":
    main()
