"""What a brick launch costs when no ray meets the volume (development tool, product library): the
staging loop alone -- claim, 133 KB image, row table, barriers -- per brick, against launches whose
rays do cross it.  512^3 -> 256^2, packed 16-bit bricks, forward and forward + record.
Usage: python tools/empty_launch_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

dev = torch.device("cuda:0")
D, H = 512, 256
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4).to(dev)
V = drr.density
print(f"# {torch.cuda.get_device_name(0)}  volume {D}^3  detector {H}^2  storage q16p")
for B in (1, 2, 8, 32):
    rot, xyz = poses(B, 2, dev)
    for label, shift in (("rays through the volume", 0.0), ("every ray beside the volume", 5000.0)):
        x = xyz.clone()
        x[:, 0] += shift
        s, t, L = rays(drr, rot, x)
        for aux in (False, True):
            fn = lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H), want_aux=aux, storage="q16p")  # noqa: E731
            out = fn()[0]
            med, best = timeit(fn)
            print(f"B {B:3d} {'fwd+rec' if aux else 'fwd    '} {label:30s} {med:.4f} ms  (image max {float(out.abs().max()):.3g})",
                  flush=True)
