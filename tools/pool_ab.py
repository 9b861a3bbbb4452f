"""The pooled end of a brick in the general brick kernel (bricks.hip) on / off (tools build, switch 2048): the
marcher's forward and volume gradient, the Siddon volume gradient, the fp32 bricks at few poses, the channel render.
Usage: python tools/pool_ab.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("exp")
from diffdrr_amd import DRR, _lib, convert, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from diffdrr_amd.renderers import _labels_u8, get_alpha_minmax  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

lib = _lib.get_lib().cdll
dev = torch.device("cuda:0")
cases = []
D, P, H = 512, 512, 512
tri = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=1.2, renderer="trilinear").to(dev)
V = tri.density
for B in (1, 4):
    s, t, L = rays(tri, *poses(B, 2, dev))
    lo, hi = get_alpha_minmax(s, t, torch.tensor(V.shape, device=dev), 0.5, 1e-8)
    amin, amax = lo.min().reshape(1).contiguous(), hi.max().reshape(1).contiguous()
    go = torch.rand(B, H * H, device=dev)
    cases.append((f"marcher forward 512^2 B={B}", lambda s=s, t=t, L=L, a=amin, b=amax: ops.trilinear_forward_bricks(V, s, t, L, a, b, (H, H), n_points=P)))
    cases.append((f"marcher volume gradient 512^2 B={B}", lambda s=s, t=t, L=L, a=amin, b=amax, go=go: ops.trilinear_backward_volume_bricks(V.shape, s, t, L, go, a, b, (H, H), n_points=P)))
sid = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=256, delx=2.4).to(dev)
for B in (1, 4, 32):
    s, t, L = rays(sid, *poses(B, 2, dev))
    go = torch.rand(B, 256 * 256, device=dev)
    cases.append((f"Siddon volume gradient 256^2 B={B}", lambda s=s, t=t, L=L, go=go: ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (256, 256))))
    if B < 8:
        cases.append((f"Siddon forward fp32 bricks 256^2 B={B}", lambda s=s, t=t, L=L: ops.siddon_forward_bricks(V, s, t, L, (256, 256), storage="f32")))
        cases.append((f"Siddon forward + record fp32 bricks 256^2 B={B}", lambda s=s, t=t, L=L: ops.siddon_forward_bricks(V, s, t, L, (256, 256), storage="f32", want_aux=True)))
# the channel render on the reference's example label map
dims, C, Hc = (512, 512, 133), 119, 200
g = torch.Generator().manual_seed(0)
vol = torch.rand(*dims, generator=g)
fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_mask_ds2.npz"))
mask = torch.from_numpy(fx["labels"]).repeat_interleave(2, 0).repeat_interleave(2, 1)[:512, :512]
ch = DRR(make_subject(vol, spacing=(0.703, 0.703, 2.5), mask=mask), sdd=1020.0, height=Hc, delx=2.0).to(dev)
(labels, _, _), = _labels_u8(ch.mask)
for B in (1, 8):
    rot = torch.zeros(B, 3, device=dev) + torch.linspace(0, 0.3, B, device=dev)[:, None]
    xyz = torch.tensor([[0.0, 850.0, 0.0]], device=dev).expand(B, 3).contiguous()
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = ch.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s_, t_ = ch.affine_inverse(source).contiguous(), ch.affine_inverse(target).contiguous()
    cases.append((f"channel render 119 labels 200^2 B={B}", lambda s_=s_, t_=t_, L=L: ops.siddon_forward_channels_bricks(ch.density, labels, C, s_, t_, L, (Hc, Hc))))
for name, fn in cases:
    row = []
    for dbg in (0, 2048):
        lib.ddrr_set_brick_debug(dbg)
        med, _ = timeit(fn)
        row.append(med)
    print(f"{name:52s} pooled {row[0]:7.3f} ms | every wave drains its own {row[1]:7.3f} ms  ({100 * (row[0] / row[1] - 1):+.1f} %)", flush=True)
