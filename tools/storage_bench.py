"""Kernel-only timing of ddrr_siddon_forward_bricks by brick storage (product library): fp32 bricks
from the volume ("f32"), guarded 16-bit bricks from their packed copy ("q16p") (round 5 also had
"f32p", fp32 bricks from a packed copy: profiles/r05/f32_packed_lookahead_experiment.*); forward and forward + record; 1 / 8 / 32 poses; on the bench's 512^3 and 256^3 noise
volumes, the 512^3 phantom and the CT-like 512 x 512 x 133 volume (diffdrr_amd.data.ct_like_hu_volume
through transform_hu_to_density).
Usage: python tools/storage_bench.py [--scenes noise512,noise256,phantom512,ct] [--poses 1,8,32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, ops  # noqa: E402
from diffdrr_amd.data import (ct_like_hu_volume, make_subject, noise_volume, phantom_volume,  # noqa: E402
                              transform_hu_to_density)
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", default="noise512,noise256,phantom512,ct")
ap.add_argument("--poses", default="1,8,32")
ap.add_argument("--storages", default="f32,q16p")
a = ap.parse_args()
dev = torch.device("cuda:0")
print(f"# {torch.cuda.get_device_name(0)}")
for scene in a.scenes.split(","):
    if scene == "ct":
        vol = transform_hu_to_density(ct_like_hu_volume())
        drr = DRR(make_subject(vol, spacing=(0.703, 0.703, 2.5)), sdd=1020.0, height=200, delx=2.0).to(dev)
        H = 200
    else:
        D = int(scene[-3:])
        vol = noise_volume(D, 0) if scene.startswith("noise") else phantom_volume(D, 0)
        H = 256
        drr = DRR(make_subject(vol), sdd=1020.0, height=H, delx=2.4 * (D / 512)).to(dev)
    V = drr.density
    for B in (int(b) for b in a.poses.split(",")):
        s, t, L = rays(drr, *poses(B, 2, dev))
        _, _, nvox = ops.siddon_forward(V, s, t, L, count_voxels=True, det=(H, H))
        alg = 4 * int(nvox.sum().item()) + B * H * H * 20 + 12 * B
        ref = ops.siddon_forward(V, s, t, L)[0]
        for aux in (False, True):
            row = f"{scene:10s} B {B:3d} {'fwd+rec' if aux else 'fwd    '}"
            for storage in a.storages.split(","):
                fn = lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H), want_aux=aux, storage=storage)  # noqa
                err = float((fn()[0] - ref).abs().max() / ref.abs().max())
                med, best = timeit(fn)
                row += f" | {storage} {med:.4f} ms ({alg / med / 1e6 / 8000 * 100:5.1f} %) err {err:.1e}"
            fb = ops.brick_fallbacks(V, "q16p") if "q16p" in a.storages else None
            print(row + (f" | q16p fallbacks {fb[0]}/{fb[1]}" if fb else ""), flush=True)
