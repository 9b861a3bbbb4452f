"""Which brick storage renders a volume with FEW double bricks per CU faster (development tool, GPU): the
table behind renderers._brick_storage's policy for volumes below 4 double bricks (32 x 32 x 64) per CU.
Kernel-only times (HIP events around 40 launches after 40) of ddrr_siddon_forward_bricks, forward and forward +
record, "q16p" against "f32", at 1 ... 32 poses: the reference's example shape 512 x 512 x 133 (CT-like volume
through transform_hu_to_density, 200 x 200 detector; 768 double bricks = 3 per CU), 256^3 noise (256 = 1 per CU,
256 x 256) and 384 x 384 x 256 noise (576 = 2.25 per CU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import perturbed_poses, voxel_rays  # noqa: E402
from diffdrr_amd import DRR, ops  # noqa: E402
from diffdrr_amd.data import ct_like_hu_volume, make_subject, noise_volume, transform_hu_to_density  # noqa: E402

dev = torch.device("cuda:0")


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
         ("384x384x256 noise -> 256^2", torch.rand(384, 384, 256, generator=torch.Generator().manual_seed(0)),
          (1.0, 1.0, 1.0), 256, 1.6)]
for name, vol, spacing, det, delx in cases:
    drr = DRR(make_subject(vol, spacing=spacing, orientation="AP"), sdd=1020.0, height=det, delx=delx).to(dev)
    V = drr.density
    nb = (-(-V.shape[0] // 32)) * (-(-V.shape[1] // 32)) * (-(-V.shape[2] // 64))
    print(f"== {name}: {nb} double bricks = {nb / 256:.2f} per CU   (ms per launch: q16p / f32; * = the faster by > 2 %)")
    with torch.no_grad():
        for B in ([int(a) for a in sys.argv[1:]] or (1, 2, 4, 8, 16, 32)):
            rot, xyz = perturbed_poses(B, seed=2, device=dev)
            s, t, L = voxel_rays(drr, rot, xyz)
            row = []
            for aux in (False, True):
                tq = timed(lambda: ops.siddon_forward_bricks(V, s, t, L, (det, det), want_aux=aux, storage="q16p"))
                tf = timed(lambda: ops.siddon_forward_bricks(V, s, t, L, (det, det), want_aux=aux, storage="f32"))
                mark = "*q16p" if tq < 0.98 * tf else ("*f32" if tf < 0.98 * tq else "tie")
                row.append(f"{'forward + record' if aux else 'forward'} {tq:.3f} / {tf:.3f} {mark}")
            print(f"   {B:2d} poses: " + "    ".join(row), flush=True)
    del drr, V
