"""Debug: channel volume gradient on the bricks vs per-ray vs the plain volume gradient."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffdrr_amd import DRR, convert, ops
from diffdrr_amd.data import make_subject
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
D = (512, 512, 133); H = 200; C = 119
vol = torch.rand(*D, generator=g)
blocks = torch.randint(0, C, (16, 16, 8), generator=g).to(torch.uint8)
lab = blocks.repeat_interleave(32, 0).repeat_interleave(32, 1).repeat_interleave(17, 2)[:, :, :133].contiguous().to(dev)
drr = DRR(make_subject(vol, spacing=(0.7, 0.7, 2.5)), sdd=1020.0, height=H, delx=2.0).to(dev)
for B in (1, 8):
    rot = torch.zeros(B, 3, device=dev) + torch.linspace(0, 0.3, B, device=dev)[:, None]
    xyz = torch.tensor([[0.0, 850.0, 0.0]], device=dev).expand(B, 3).contiguous()
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s, t = drr.affine_inverse(source).contiguous(), drr.affine_inverse(target).contiguous()
    go1 = torch.rand(B, 1, H * H, device=dev)
    go_same = go1.expand(B, C, H * H).contiguous()
    go_rand = torch.rand(B, C, H * H, device=dev)
    plain_b = ops.siddon_backward_volume_bricks(D, s, t, L, go1[:, 0].contiguous(), (H, H))
    plain_r = ops.siddon_backward_volume(drr.density, s, t, L, go1[:, 0].contiguous())
    ch_b = ops.siddon_backward_channels_volume_bricks(lab, s, t, L, go_same, (H, H))
    ch_r = ops.siddon_backward_channels(drr.density, lab, s, t, L, go_same, want_rays=False, want_img=False, want_volume=True, det=(H, H))[3]
    m = float(plain_r.abs().max())
    e = lambda a, b: float((a - b).abs().max()) / m
    print(f"B {B}: same go in every channel: plain bricks vs plain per-ray {e(plain_b, plain_r):.1e} | channel bricks vs plain per-ray {e(ch_b, plain_r):.1e} | channel per-ray vs plain per-ray {e(ch_r, plain_r):.1e}")
    ch_b = ops.siddon_backward_channels_volume_bricks(lab, s, t, L, go_rand, (H, H))
    ch_r = ops.siddon_backward_channels(drr.density, lab, s, t, L, go_rand, want_rays=False, want_img=False, want_volume=True, det=(H, H))[3]
    ch_r2 = ops.siddon_backward_channels(drr.density, lab, s, t, L, go_rand, want_rays=False, want_img=False, want_volume=True, det=(H, H))[3]
    d = (ch_b - ch_r).abs()
    i = int(d.argmax()); x, y, z = i // (D[1] * D[2]), (i // D[2]) % D[1], i % D[2]
    print(f"      random go: channel bricks vs per-ray {e(ch_b, ch_r):.1e} at voxel {(x, y, z)} (bricks {float(ch_b[x, y, z]):.6f} per-ray {float(ch_r[x, y, z]):.6f}, label {int(lab[x,y,z])}); per-ray twice {e(ch_r, ch_r2):.1e}; voxels off by > 1e-5: {int((d > 1e-5 * m).sum())}")
