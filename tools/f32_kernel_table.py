"""fp32 bricks: the general kernel (bricks.hip, taken below 8 poses today) against the configurable one
(bricks_fwd.hip CfgF32) by pose count, on the shapes of tools/storage_table.py (development tool, GPU; tools build:
ddrr_set_brick_variant -1 / 0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("exp")
from bench import perturbed_poses, voxel_rays  # noqa: E402
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import ct_like_hu_volume, make_subject, noise_volume, transform_hu_to_density  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.get_lib()


def timed(fn, warm=40, n=40):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cases = [("512x512x133 CT-like -> 200^2", transform_hu_to_density(ct_like_hu_volume((512, 512, 133), seed=0)),
          (0.703, 0.703, 2.5), 200, 2.0),
         ("256^3 noise -> 256^2", noise_volume(256, seed=0), (1.0, 1.0, 1.0), 256, 1.2),
         ("512^3 noise -> 256^2", noise_volume(512, seed=0), (1.0, 1.0, 1.0), 256, 2.4)]
for name, vol, spacing, det, delx in cases:
    drr = DRR(make_subject(vol, spacing=spacing, orientation="AP"), sdd=1020.0, height=det, delx=delx).to(dev)
    V = drr.density
    print(f"== {name}   (ms per launch, fp32 bricks: general kernel / configurable kernel)")
    with torch.no_grad():
        for B in ([int(a) for a in sys.argv[1:]] or (1, 4, 6, 8, 10, 12, 16, 24, 32)):
            rot, xyz = perturbed_poses(B, seed=2, device=dev)
            s, t, L = voxel_rays(drr, rot, xyz)
            row = []
            for aux in (False, True):
                ts = []
                for v in (-1, 0):
                    lib.cdll.ddrr_set_brick_variant(v)
                    ts.append(timed(lambda: ops.siddon_forward_bricks(V, s, t, L, (det, det), want_aux=aux, storage="f32")))
                row.append(f"{'forward + record' if aux else 'forward'} {ts[0]:.3f} / {ts[1]:.3f}"
                           f" {'*general' if ts[0] < 0.98 * ts[1] else ('*configurable' if ts[1] < 0.98 * ts[0] else 'tie')}")
            print(f"   {B:2d} poses: " + "    ".join(row), flush=True)
    lib.cdll.ddrr_set_brick_variant(-2)
    del drr, V
