"""Generate the golden fixtures in this directory from the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every fixture stores the exact inputs handed to the reference and what the
reference returned for them on CPU, in fp32 and in fp64 (module ``.to(float64)``),
including torch-autograd gradients for a stored ``grad_out``.  The reference's
own tests hold no golden vectors for the rendering path, so
these files are the pin for ``oracle/drr_oracle.c`` and, through it, for the HIP
kernels.  The fixtures are small (a few hundred kB in total) by construction.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402

ref = ref_loader.load()
F32, F64 = torch.float32, torch.float64


def npy(t):
    return None if t is None else t.detach().cpu().numpy()


def save(name, **arrays):
    arrays = {k: v for k, v in arrays.items() if v is not None}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:34s} {os.path.getsize(path) / 1024:8.1f} kB")


# ----------------------------------------------------------------- ray sets

def random_rays(g, dims, B, N, per_ray_source=False):
    """Rays that mostly cross the volume: sources ~40 voxels in front of it."""
    c = torch.tensor([d / 2 for d in dims])
    n_src = N if per_ray_source else 1
    src = c + torch.tensor([0.0, -45.0, 0.0]) + 3 * torch.randn(B, n_src, 3, generator=g)
    tgt = c + torch.tensor([0.0, 50.0, 0.0]) + 6 * torch.randn(B, N, 3, generator=g)
    return src, tgt


def oblique_rays(g, dims, B, N):
    """Rays from random directions (all three axes take turns being dominant)."""
    c = torch.tensor([d / 2 for d in dims])
    dirs = torch.randn(B, 1, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    src = c + 60 * dirs
    tgt = c - 50 * dirs + 7 * torch.randn(B, N, 3, generator=g)
    return src, tgt


def special_rays(dims):
    """Hand-made rays: axis-parallel (exercise eps), misses, source inside the
    volume, target inside the volume (whole-line semantics), grazing a face."""
    Dx, Dy, Dz = dims
    s, t = [], []
    s.append([3.2, -20.0, 4.7]); t.append([3.2, 40.0, 4.7])          # parallel to y
    s.append([-30.0, 2.3, 5.1]); t.append([50.0, 2.3, 5.1])          # parallel to x
    s.append([4.4, 3.3, -25.0]); t.append([4.4, 3.3, 45.0])          # parallel to z
    s.append([4.4, 3.3, 45.0]); t.append([4.4, 3.3, -25.0])          # parallel to -z
    s.append([3.0, -20.0, 4.0]); t.append([3.0, 40.0, 4.0])          # through voxel centres
    s.append([Dx + 5.0, -20.0, 4.0]); t.append([Dx + 5.0, 40.0, 4.0])  # parallel miss
    s.append([-40.0, -40.0, -3.0]); t.append([60.0, 50.0, -2.0])     # oblique miss (below)
    s.append([Dx / 2, Dy / 2, Dz / 2]); t.append([Dx + 30.0, Dy + 10.0, Dz / 3])  # source inside
    s.append([-25.0, 1.0, 2.0]); t.append([Dx / 2, Dy / 2, Dz / 2])  # target inside
    s.append([-30.0, 4.0, 3.0]); t.append([-10.0, 4.5, 3.5])         # volume beyond target
    s.append([-10.0, -9.0, -8.0]); t.append([Dx + 9.0, Dy + 10.0, Dz + 11.0])  # body diagonal-ish
    s.append([2.5, -30.0, 1.5]); t.append([2.5, 30.0, 6.5])          # exactly on an x-plane
    s.append([0.2, -0.45, Dz + 20.0]); t.append([Dx - 1.2, Dy - 0.55, -20.0])  # along -z, oblique
    src = torch.tensor(s).unsqueeze(0).transpose(0, 1)  # (R,1,3): one "pose" per ray
    tgt = torch.tensor(t).unsqueeze(0).transpose(0, 1)  # (R,1,3)
    return src.contiguous(), tgt.contiguous()


def run_renderer(make, call_kw, volume, src, tgt, dtype, g, mask=None, want_grads=True,
                 extra_inputs=None):
    """Run a reference renderer in `dtype`, return out + autograd grads."""
    vol = volume.to(dtype).clone().requires_grad_(want_grads)
    s = src.to(dtype).clone().requires_grad_(want_grads)
    t = tgt.to(dtype).clone().requires_grad_(want_grads)
    img = (t - s).norm(dim=-1)
    if img.shape[1] != t.shape[1]:
        img = img.expand(-1, t.shape[1])
    img = img.unsqueeze(1).detach().clone().requires_grad_(want_grads)
    renderer = make().to(dtype)
    kw = dict(call_kw)
    extras = {}
    if extra_inputs:
        for k, v in extra_inputs.items():
            extras[k] = torch.tensor(v, dtype=dtype, requires_grad=want_grads)
            kw[k] = extras[k]
    if mask is not None:
        kw["mask"] = mask.to(dtype)
    out = renderer(vol, s, t, img, **kw)
    res = {"out": npy(out), "img": npy(img)}
    if want_grads:
        go = torch.randn(out.shape, generator=g).to(dtype)
        leaves = [s, t, img, vol] + list(extras.values())
        grads = torch.autograd.grad(out, leaves, go, allow_unused=True)
        names = ["g_source", "g_target", "g_img", "g_volume"] + ["g_" + k for k in extras]
        res["grad_out"] = npy(go)
        for n, gr in zip(names, grads):
            res[n] = npy(gr)
    return res


def renderer_fixture(name, make, call_kw, dims, rays, seed, mask=None, want_grads=True,
                     extra_inputs=None, meta=None):
    g = torch.Generator().manual_seed(seed)
    volume = torch.rand(*dims, generator=g)
    src, tgt = rays(g) if callable(rays) else rays
    arrays = {"volume": npy(volume), "source": npy(src), "target": npy(tgt)}
    if mask is not None:
        arrays["mask"] = npy(mask)
    for dtype, tag in ((F32, "f32"), (F64, "f64")):
        gg = torch.Generator().manual_seed(seed + 1)
        res = run_renderer(make, call_kw, volume, src, tgt, dtype, gg, mask, want_grads,
                           extra_inputs)
        for k, v in res.items():
            if v is not None:
                arrays[f"{k}_{tag}"] = v
    for k, v in (meta or {}).items():
        arrays["meta_" + k] = np.asarray(v)
    save(name, **arrays)


def topk_sum(img):
    """The reference tutorial's custom reduction: sum of the k largest per-segment terms."""
    return img.sort(descending=True).values[..., :6].sum(dim=-1)


def make_renderer_fixtures():
    dims = (12, 10, 14)  # deliberately non-cubic: catches axis transposes
    R = ref
    renderer_fixture("siddon_sum", lambda: R.Siddon(), {}, dims,
                     lambda g: random_rays(g, dims, 2, 48), 10)
    renderer_fixture("siddon_sum_oblique", lambda: R.Siddon(), {}, (9, 16, 11),
                     lambda g: oblique_rays(g, (9, 16, 11), 6, 24), 11)
    renderer_fixture("siddon_max", lambda: R.Siddon(reducefn="max"), {}, dims,
                     lambda g: random_rays(g, dims, 2, 48), 12)
    renderer_fixture("siddon_per_ray_source", lambda: R.Siddon(), {}, dims,
                     lambda g: random_rays(g, dims, 2, 20, per_ray_source=True), 13)
    renderer_fixture("siddon_special_rays", lambda: R.Siddon(), {}, (7, 8, 9),
                     special_rays((7, 8, 9)), 14, want_grads=False)
    renderer_fixture("siddon_shift0", lambda: R.Siddon(voxel_shift=0.0), {}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 15,
                     meta={"voxel_shift": 0.0})
    renderer_fixture("siddon_stopgrad",
                     lambda: R.Siddon(stop_gradients_through_grid_sample=True), {}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 16)
    renderer_fixture("siddon_bilinear", lambda: R.Siddon(mode="bilinear"), {}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 17)
    renderer_fixture("siddon_align_corners", lambda: R.Siddon(), {"align_corners": True}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 18)
    # a callable reducefn over the per-segment tensor (introduction.ipynb:506-529: top-k sum)
    renderer_fixture("siddon_callable", lambda: R.Siddon(reducefn=topk_sum), {}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 26)
    g = torch.Generator().manual_seed(19)
    mask = torch.randint(0, 5, dims, generator=g).to(F32)
    renderer_fixture("siddon_mask", lambda: R.Siddon(), {}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 19, mask=mask)

    renderer_fixture("trilinear_global_range", lambda: R.Trilinear(), {"n_points": 41}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 20)
    renderer_fixture("trilinear_explicit_range", lambda: R.Trilinear(), {"n_points": 64}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 21,
                     extra_inputs={"alphamin": 0.31, "alphamax": 0.77})
    renderer_fixture("trilinear_oblique", lambda: R.Trilinear(), {"n_points": 50}, (9, 16, 11),
                     lambda g: oblique_rays(g, (9, 16, 11), 6, 24), 22)
    renderer_fixture("trilinear_nearest_max",
                     lambda: R.Trilinear(mode="nearest", reducefn="max"), {"n_points": 33}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 23, want_grads=False)
    renderer_fixture("trilinear_mask", lambda: R.Trilinear(), {"n_points": 40}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 24, mask=mask)
    renderer_fixture("trilinear_max", lambda: R.Trilinear(reducefn="max"), {"n_points": 37}, dims,
                     lambda g: random_rays(g, dims, 2, 32), 28)
    renderer_fixture("trilinear_callable", lambda: R.Trilinear(reducefn=topk_sum),
                     {"n_points": 40}, dims, lambda g: random_rays(g, dims, 2, 32), 27)
    renderer_fixture("trilinear_shift0", lambda: R.Trilinear(voxel_shift=0.0), {"n_points": 40},
                     dims, lambda g: random_rays(g, dims, 2, 32), 25, meta={"voxel_shift": 0.0})


# ---------------------------------------------------------------- DRR level

def synthetic_subject(dims, spacing, seed, orientation="AP", with_mask=False):
    """A torchio-like Subject around a seeded noise volume, centred like
    reference data.py:187-211 (`canonicalize`) and reoriented like data.py:87-120."""
    g = torch.Generator().manual_seed(seed)
    vol = torch.rand(*dims, generator=g)
    affine = np.diag([spacing[0], spacing[1], spacing[2], 1.0])
    affine[:3, 3] = [-(d - 1) / 2 * s for d, s in zip(dims, spacing)]
    reorient = {
        "AP": [[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]],
        "PA": [[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]],
        None: [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
    }[orientation]
    mask = None
    if with_mask:
        labels = torch.randint(0, 4, dims, generator=g)
        mask = ref.LabelMap(labels.unsqueeze(0), affine)
    subject = ref.Subject(
        volume=ref.ScalarImage(vol.unsqueeze(0), affine),
        density=ref.ScalarImage(vol.unsqueeze(0), affine),
        mask=mask,
        reorient=torch.tensor(reorient, dtype=F32),
        fiducials=None,
    )
    return subject, vol, affine


def make_drr_fixtures():
    dims, spacing = (16, 14, 18), (1.5, 2.0, 1.25)
    subject, vol, affine = synthetic_subject(dims, spacing, 30, "AP", with_mask=True)
    geo = dict(sdd=300.0, height=10, width=12, delx=2.2, dely=1.9, x0=3.0, y0=-2.0)
    rot = torch.tensor([[0.0, 0.0, 0.0], [0.3, -0.2, 0.1], [-0.5, 0.4, 0.25]])
    xyz = torch.tensor([[0.0, 200.0, 0.0], [5.0, 180.0, -4.0], [-8.0, 230.0, 6.0]])
    arrays = {"volume": npy(vol), "affine": affine, "reorient": npy(subject.reorient),
              "mask": npy(subject.mask.data.squeeze()), "rot": npy(rot), "xyz": npy(xyz),
              **{"geo_" + k: np.asarray(v) for k, v in geo.items()}}
    for renderer, kw in (("siddon", {}), ("trilinear", {"n_points": 60})):
        for dtype, tag in ((F32, "f32"), (F64, "f64")):
            drr = ref.DRR(subject, renderer=renderer, **geo).to(dtype)
            r = rot.to(dtype).clone().requires_grad_()
            x = xyz.to(dtype).clone().requires_grad_()
            pose = ref.convert(r, x, parameterization="euler_angles", convention="ZXY")
            source, target = drr.detector(pose, None)
            img = drr(r, x, parameterization="euler_angles", convention="ZXY", **kw)
            g = torch.Generator().manual_seed(31)
            go = torch.randn(img.shape, generator=g).to(dtype)
            gr, gx = torch.autograd.grad(img, (r, x), go)
            arrays.update({
                f"{renderer}_img_{tag}": npy(img), f"{renderer}_grad_out_{tag}": npy(go),
                f"{renderer}_g_rot_{tag}": npy(gr), f"{renderer}_g_xyz_{tag}": npy(gx),
            })
            if renderer == "siddon":
                arrays[f"pose_matrix_{tag}"] = npy(pose.matrix)
                arrays[f"det_source_{tag}"] = npy(source)
                arrays[f"det_target_{tag}"] = npy(target)
                ch = drr(r.detach(), x.detach(), parameterization="euler_angles",
                         convention="ZXY", mask_to_channels=True)
                arrays[f"siddon_channels_{tag}"] = npy(ch)
                # patch_size path (drr.py:217-225)
                drr_p = ref.DRR(subject, renderer="siddon", patch_size=2, **geo).to(dtype)
                arrays[f"siddon_patched_{tag}"] = npy(
                    drr_p(r.detach(), x.detach(), parameterization="euler_angles",
                          convention="ZXY"))
    # odd-sized detector, reverse_x_axis=False, PA orientation, no mask
    subject2, vol2, affine2 = synthetic_subject((11, 13, 9), (2.0, 1.0, 1.5), 32, "PA")
    geo2 = dict(sdd=250.0, height=7, width=5, delx=3.0)
    drr2 = ref.DRR(subject2, reverse_x_axis=False, **geo2)
    rot2 = torch.tensor([[0.2, 0.1, -0.3]])
    xyz2 = torch.tensor([[3.0, 150.0, -2.0]])
    pose2 = ref.convert(rot2, xyz2, parameterization="euler_angles", convention="ZXY")
    s2, t2 = drr2.detector(pose2, None)
    arrays.update({"b_volume": npy(vol2), "b_affine": affine2, "b_reorient": npy(subject2.reorient),
                   "b_rot": npy(rot2), "b_xyz": npy(xyz2), "b_det_source": npy(s2),
                   "b_det_target": npy(t2), "b_img": npy(drr2(pose2)),
                   **{"b_geo_" + k: np.asarray(v) for k, v in geo2.items()}})
    save("drr_module", **arrays)


def make_pose_fixtures():
    g = torch.Generator().manual_seed(40)
    B = 4
    arrays = {}
    t = 50 * torch.randn(B, 3, generator=g)
    arrays["translation"] = npy(t)
    cases = {
        "axis_angle": (torch.randn(B, 3, generator=g), {}),
        "euler_angles": (torch.randn(B, 3, generator=g), {"convention": "ZXY"}),
        "euler_angles_deg": (60 * torch.randn(B, 3, generator=g),
                             {"convention": "XYZ", "degrees": True}),
        "quaternion": (torch.randn(B, 4, generator=g), {}),
        "rotation_6d": (torch.randn(B, 6, generator=g), {}),
        "rotation_9d": (torch.randn(B, 9, generator=g), {}),
        "rotation_10d": (torch.randn(B, 10, generator=g), {}),
        "quaternion_adjugate": (torch.randn(B, 10, generator=g), {}),
        "se3_log_map": (0.7 * torch.randn(B, 3, generator=g), {}),
    }
    for name, (r, kw) in cases.items():
        param = "euler_angles" if name.startswith("euler") else name
        T = ref.convert(r, t, parameterization=param, **kw)
        arrays[name + "_in"] = npy(r)
        arrays[name + "_matrix"] = npy(T.matrix)
        # round trip through RigidTransform.convert (pose.py:73-102)
        back_kw = {"convention": kw.get("convention")} if param == "euler_angles" else {}
        rr, tt = T.convert(param, **back_kw)
        arrays[name + "_back_rot"] = npy(rr)
        arrays[name + "_back_xyz"] = npy(tt)
    A = ref.convert(cases["euler_angles"][0], t, parameterization="euler_angles", convention="ZXY")
    Bm = ref.convert(cases["axis_angle"][0], -t, parameterization="axis_angle")
    pts = torch.randn(B, 7, 3, generator=g)
    arrays["apply_pts"] = npy(pts)
    arrays["apply_out"] = npy(A(pts))
    arrays["compose_matrix"] = npy(A.compose(Bm).matrix)
    arrays["inverse_matrix"] = npy(A.inverse().matrix)
    save("pose", **arrays)


def make_registration_fixture():
    """First SGD steps of the registration tutorial's loop at a tiny size
    (registration.ipynb:240-316): Registration + NCC, maximise."""
    dims, spacing = (24, 24, 24), (2.0, 2.0, 2.0)
    g = torch.Generator().manual_seed(50)
    # smooth blobs so that NCC has a useful gradient
    zz, yy, xx = torch.meshgrid(*[torch.arange(d, dtype=F32) for d in dims], indexing="ij")
    vol = torch.zeros(dims)
    for _ in range(6):
        c = torch.rand(3, generator=g) * 14 + 5
        w = torch.rand(3, generator=g) * 4 + 2
        vol += torch.exp(-(((zz - c[0]) / w[0]) ** 2 + ((yy - c[1]) / w[1]) ** 2
                           + ((xx - c[2]) / w[2]) ** 2))
    vol = vol / vol.max()
    affine = np.diag([*spacing, 1.0])
    affine[:3, 3] = [-(d - 1) / 2 * s for d, s in zip(dims, spacing)]
    reorient = torch.tensor([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=F32)
    subject = ref.Subject(volume=ref.ScalarImage(vol.unsqueeze(0), affine),
                          density=ref.ScalarImage(vol.unsqueeze(0), affine), mask=None,
                          reorient=reorient, fiducials=None)
    geo = dict(sdd=400.0, height=16, delx=4.0)
    drr = ref.DRR(subject, **geo)
    true_rot = torch.tensor([[0.0, 0.0, 0.0]])
    true_xyz = torch.tensor([[0.0, 300.0, 0.0]])
    gt = drr(true_rot, true_xyz, parameterization="euler_angles", convention="ZXY").detach()
    rot0 = torch.tensor([[0.15, -0.1, 0.08]])
    xyz0 = torch.tensor([[6.0, 290.0, -5.0]])
    arrays = {"volume": npy(vol), "affine": affine, "reorient": npy(reorient), "gt": npy(gt),
              "true_rot": npy(true_rot), "true_xyz": npy(true_xyz), "rot0": npy(rot0),
              "xyz0": npy(xyz0), **{"geo_" + k: np.asarray(v) for k, v in geo.items()}}
    for stop in (False, True):
        d = ref.DRR(subject, stop_gradients_through_grid_sample=stop, **geo)
        reg = ref.Registration(d, rot0.clone(), xyz0.clone(), parameterization="euler_angles",
                               convention="ZXY")
        crit = ref.NCC()
        opt = torch.optim.SGD([{"params": [reg._rotation], "lr": 5e-2},
                               {"params": [reg._translation], "lr": 1e2}], maximize=True)
        losses, rots, xyzs = [], [], []
        for _ in range(8):
            opt.zero_grad()
            loss = crit(gt, reg()).mean()
            loss.backward()
            losses.append(loss.item())
            rots.append(npy(reg._rotation).copy())
            xyzs.append(npy(reg._translation).copy())
            opt.step()
        tag = "stop" if stop else "full"
        arrays[f"losses_{tag}"] = np.asarray(losses)
        arrays[f"rots_{tag}"] = np.stack(rots)
        arrays[f"xyzs_{tag}"] = np.stack(xyzs)
    # NCC known values on seeded inputs
    a = torch.randn(3, 1, 16, 16, generator=g)
    b = torch.randn(3, 1, 16, 16, generator=g)
    arrays["ncc_a"], arrays["ncc_b"] = npy(a), npy(b)
    arrays["ncc_ab"] = npy(ref.NCC()(a, b))
    arrays["ncc_ab_patch5"] = npy(ref.NCC(patch_size=5)(a, b))
    save("registration", **arrays)


def make_swap_fixture():
    """The renderer seam of the reference (drr.py:94-101, 209-224) at a size where the
    volume-stationary kernels have several bricks: the world-space rays an UNMODIFIED
    ``diffdrr.drr.DRR`` hands to ``DRR.render``, and what the reference's own renderers return
    for them -- image and autograd gradients w.r.t. those rays.  The GPU test
    (tests/test_gpu_parity.py::test_reference_rays_through_swapped_renderers) feeds the same
    rays to ``diffdrr_amd.Siddon`` / ``Trilinear`` as a swapped-in renderer would get them."""
    dims, spacing = (40, 44, 36), (1.5, 1.25, 1.75)
    subject, vol, affine = synthetic_subject(dims, spacing, 70, "AP")
    # (smooth part + noise: gradients that are not all tie-breaking)
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, d) for d in dims], indexing="ij")
    vol = 0.5 * vol + torch.exp(-3 * (zz ** 2 + yy ** 2 + xx ** 2))
    subject = ref.Subject(volume=ref.ScalarImage(vol.unsqueeze(0), affine),
                          density=ref.ScalarImage(vol.unsqueeze(0), affine), mask=None,
                          reorient=subject.reorient, fiducials=None)
    geo = dict(sdd=500.0, height=40, width=36, delx=2.0)
    rot = torch.tensor([[0.05, 0.02, -0.03], [0.6, -0.4, 0.3], [-0.9, 0.2, 0.7]])
    xyz = torch.tensor([[2.0, 330.0, -1.0], [6.0, 300.0, -5.0], [-9.0, 360.0, 4.0]])
    arrays = {"volume": npy(vol), "affine": affine, "rot": npy(rot), "xyz": npy(xyz),
              **{"geo_" + k: np.asarray(v) for k, v in geo.items()}}
    for renderer, kw in (("siddon", {}), ("trilinear", {"n_points": 80})):
        drr = ref.DRR(subject, renderer=renderer, **geo)
        pose = ref.convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        source = source.detach().clone().requires_grad_()
        target = target.detach().clone().requires_grad_()
        img = drr.render(drr.density, source, target, **kw)      # (B, 1, N)
        go = torch.randn(img.shape, generator=torch.Generator().manual_seed(71))
        gs, gt = torch.autograd.grad(img, (source, target), go)
        if renderer == "siddon":
            arrays.update(source=npy(source), target=npy(target),
                          affine_inverse=npy(drr.affine_inverse.matrix[0]), grad_out=npy(go))
        arrays.update({f"{renderer}_img": npy(img), f"{renderer}_g_source": npy(gs),
                       f"{renderer}_g_target": npy(gt)})
        # the same call in float64 (module .to(float64), the same fp32 rays): the yardstick for
        # gradients, which the reference's own fp32 arithmetic misses by up to 6e-3 here
        drr64 = ref.DRR(subject, renderer=renderer, **geo).to(F64)
        s64 = source.detach().to(F64).requires_grad_()
        t64 = target.detach().to(F64).requires_grad_()
        img64 = drr64.render(drr64.density, s64, t64, **kw)
        gs64, gt64 = torch.autograd.grad(img64, (s64, t64), go.to(F64))
        arrays.update({f"{renderer}_img_f64": npy(img64), f"{renderer}_g_source_f64": npy(gs64),
                       f"{renderer}_g_target_f64": npy(gt64)})
    save("reference_drr_rays", **arrays)


def make_metrics_fixture():
    """Image similarities of the reference (diffdrr/metrics.py:21-104) with autograd gradients:
    NormalizedCrossCorrelation2d whole-image and patch-wise, the multiscale sum, and
    GradientNormalizedCrossCorrelation2d (Sobel :69-94) without blur (sigma = 0: the reference's
    own code only) and with the default sigma = 1 (whose Gaussian is torchvision's
    ``gaussian_blur``, absent here and restated in oracle/ref_shims from its published
    algorithm)."""
    g = torch.Generator().manual_seed(80)
    B, H, W = 3, 24, 20
    # smooth structure + noise, a shifted / rescaled partner: similarities well away from 0 and 1
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    base = torch.stack([torch.exp(-3 * ((yy - 0.2 * k) ** 2 + (xx + 0.1 * k) ** 2)) for k in range(B)])
    a = (base + 0.15 * torch.rand(B, H, W, generator=g)).unsqueeze(1)
    b = (1.7 * base.roll(1, dims=-1) + 0.3 + 0.2 * torch.rand(B, H, W, generator=g)).unsqueeze(1)
    w = torch.rand(B, generator=g)  # weights of the per-pair values in the scalar loss
    arrays = {"a": npy(a), "b": npy(b), "w": npy(w)}
    M = ref.metrics
    cases = {
        "ncc": M.NormalizedCrossCorrelation2d(),
        "ncc_patch5": M.NormalizedCrossCorrelation2d(patch_size=5),
        "multiscale": M.MultiscaleNormalizedCrossCorrelation2d([None, 7], [0.5, 0.5]),
        "gncc_sigma0": M.GradientNormalizedCrossCorrelation2d(sigma=0.0),
        "gncc_sigma1": M.GradientNormalizedCrossCorrelation2d(sigma=1.0),
        "gncc_patch7_sigma0": M.GradientNormalizedCrossCorrelation2d(patch_size=7, sigma=0.0),
    }
    for name, crit in cases.items():
        for dtype, tag in ((F32, "f32"), (F64, "f64")):
            crit = crit.to(dtype) if hasattr(crit, "to") else crit
            x1 = a.to(dtype).clone().requires_grad_()
            x2 = b.to(dtype).clone().requires_grad_()
            val = crit(x1, x2)
            g1, g2 = torch.autograd.grad((val * w.to(dtype)).sum(), (x1, x2))
            arrays.update({f"{name}_{tag}": npy(val), f"{name}_g1_{tag}": npy(g1),
                           f"{name}_g2_{tag}": npy(g2)})
    sob = M.Sobel(0.0)
    arrays["sobel_a"] = npy(sob(a))
    save("metrics", **arrays)


def make_general_fixtures():
    """The keyword combinations the fused kernels do not take (the materialising general path,
    csrc/general_core.h): masks / callables / max / stop-gradients together with the midpoint
    lookups, and the marcher's mode="nearest" with a mask.  Their float64 halves -- and those of
    the fixtures above -- also pin the float64 general path."""
    dims = (12, 10, 14)
    R = ref
    g = torch.Generator().manual_seed(19)
    mask = torch.randint(0, 5, dims, generator=g).to(F32)
    rays = lambda gg: random_rays(gg, dims, 2, 32)  # noqa: E731
    renderer_fixture("siddon_bilinear_mask", lambda: R.Siddon(mode="bilinear"), {}, dims, rays, 40,
                     mask=mask)
    renderer_fixture("siddon_align_mask", lambda: R.Siddon(), {"align_corners": True}, dims, rays,
                     41, mask=mask)
    renderer_fixture("siddon_mask_max", lambda: R.Siddon(reducefn="max"), {}, dims, rays, 42,
                     mask=mask)
    renderer_fixture("siddon_bilinear_callable",
                     lambda: R.Siddon(mode="bilinear", reducefn=topk_sum), {}, dims, rays, 43)
    renderer_fixture("siddon_align_max", lambda: R.Siddon(reducefn="max"),
                     {"align_corners": True}, dims, rays, 44)
    renderer_fixture("siddon_bilinear_max", lambda: R.Siddon(mode="bilinear", reducefn="max"), {},
                     dims, rays, 45)
    renderer_fixture("siddon_bilinear_stopgrad",
                     lambda: R.Siddon(mode="bilinear", stop_gradients_through_grid_sample=True),
                     {}, dims, rays, 46)
    renderer_fixture("trilinear_nearest_mask", lambda: R.Trilinear(mode="nearest"),
                     {"n_points": 40}, dims, rays, 47, mask=mask)
    renderer_fixture("trilinear_mask_callable", lambda: R.Trilinear(reducefn=topk_sum),
                     {"n_points": 40}, dims, rays, 48, mask=mask)
    renderer_fixture("trilinear_align_corners", lambda: R.Trilinear(),
                     {"n_points": 45, "align_corners": True}, dims, rays, 49)
    renderer_fixture("trilinear_nearest", lambda: R.Trilinear(mode="nearest"), {"n_points": 45},
                     dims, rays, 50)


if __name__ == "__main__":
    torch.set_num_threads(4)
    if sys.argv[1:] == ["general"]:
        make_general_fixtures()
        sys.exit(0)
    if sys.argv[1:] == ["swap"]:
        make_swap_fixture()
        sys.exit(0)
    if sys.argv[1:] == ["metrics"]:
        make_metrics_fixture()
        sys.exit(0)
    make_renderer_fixtures()
    make_general_fixtures()
    make_drr_fixtures()
    make_pose_fixtures()
    make_registration_fixture()
    make_swap_fixture()
    make_metrics_fixture()
