"""Which part of DRR.ncc's pose gradient is how far from the fp64 oracle chain (development tool, GPU):
one pose of tests/test_gpu_baseline_sizes.py::test_drr_ncc_vs_fp64_oracle_chain, the gradient through
the fused step, the composed step on the bricks (f32 / q16p storage) and the composed step on the per-ray
kernel (every alpha the reference's quotient), each against the fp64 chain; the reference's fp32 chain."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import OracleChain, _ncc_grad64  # noqa: E402
from conftest import rel_err  # noqa: E402
from test_gpu_parity import scene, voxel_rays  # noqa: E402

gpu = torch.device("cuda:0")
D, det, delx, kind = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
picks = [int(a) for a in sys.argv[5:]] or [31]
B = 32
drr, rot, xyz = scene(D, det, delx, B + 1, gpu, seed=5, kind=kind)
with torch.no_grad():
    fixed = drr(rot[:1], xyz[:1], parameterization="euler_angles", convention="ZXY")
rot, xyz = rot[1:].contiguous(), xyz[1:].contiguous()
chain = OracleChain(drr)
fx = fixed.reshape(-1).cpu().numpy()


def grads(route, storage=None, path="bricks", nb=B):
    drr.FUSED_NCC_MAX_POSES = 32 if route == "fused" else 0
    drr.renderer.grid_path = path
    had = drr.renderer.brick_storage
    if storage:
        drr.renderer.brick_storage = storage
    r, x = rot[:nb].clone().requires_grad_(), xyz[:nb].clone().requires_grad_()
    drr.ncc(fixed, r, x, convention="ZXY").sum().backward()
    drr.renderer.brick_storage, drr.renderer.grid_path = had, "bricks"
    return r.grad.cpu().numpy(), x.grad.cpu().numpy()


from diffdrr_amd.renderers import _brick_storage  # noqa: E402
print("module storage for this volume:", _brick_storage(drr.density, {"storage": drr.renderer.brick_storage}))
variants = {"fused": grads("fused"), "composed bricks": grads("composed"),
            "composed per-ray": grads("composed", path="generic")}
for b in picks:
    rays32 = tuple(a.cpu().numpy() for a in voxel_rays(drr, rot[b:b + 1], xyz[b:b + 1]))
    _, _, img64 = chain(rot[b], xyz[b], rays32, np.zeros(fx.size), np.float64)
    W = _ncc_grad64(fx, img64)
    gr64, gx64, _ = chain(rot[b], xyz[b], rays32, W, np.float64)
    gr32, gx32, _ = chain(rot[b], xyz[b], rays32, W, np.float32)
    truth = np.concatenate([gr64, gx64 * 100.0])
    print(f"pose {b}: truth {truth}")
    print(f"   reference fp32 chain: {rel_err(np.concatenate([gr32, gx32 * 100.0]), truth):.2e}")
    for name, (g_r, g_x) in variants.items():
        mine = np.concatenate([g_r[b], g_x[b] * 100.0])
        print(f"   {name:18s}: {rel_err(mine, truth):.2e}   {mine - truth}")
    # the same weights W through plain autograd of the module (no NCC arithmetic on the device)
    r, x = rot[b:b + 1].clone().requires_grad_(), xyz[b:b + 1].clone().requires_grad_()
    img = drr(r, x, parameterization="euler_angles", convention="ZXY")
    (img.reshape(-1) * torch.from_numpy(W).float().to(gpu)).sum().backward()
    mine = np.concatenate([r.grad.cpu().numpy()[0], x.grad.cpu().numpy()[0] * 100.0])
    print(f"   exact weights, bricks: {rel_err(mine, truth):.2e}   {mine - truth}")
