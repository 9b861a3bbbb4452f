"""Stage stamps of the channel render's bricks (bricks.hip BRICK_CHANNELS) on the reference's example label map,
512 x 512 x 133 -> 200^2, 119 labels (profile build; development tool).  Usage: python tools/channel_stamps.py [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("prof")
from diffdrr_amd import DRR, _lib, convert, ops  # noqa: E402
from diffdrr_amd.data import make_subject  # noqa: E402
from diffdrr_amd.renderers import _labels_u8  # noqa: E402
from tools.kernel_sweep import timeit  # noqa: E402

lib = _lib.get_lib().cdll
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dims, C, H = (512, 512, 133), 119, 200
g = torch.Generator().manual_seed(0)
vol = torch.rand(*dims, generator=g)
fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_mask_ds2.npz"))
mask = torch.from_numpy(fx["labels"]).repeat_interleave(2, 0).repeat_interleave(2, 1)[:512, :512]
drr = DRR(make_subject(vol, spacing=(0.703, 0.703, 2.5), mask=mask), sdd=1020.0, height=H, delx=2.0).to(dev)
(labels, _, _), = _labels_u8(drr.mask)
rot = torch.zeros(B, 3, device=dev) + torch.linspace(0, 0.3, B, device=dev)[:, None]
xyz = torch.tensor([[0.0, 850.0, 0.0]], device=dev).expand(B, 3).contiguous()
with torch.no_grad():
    pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
    source, target = drr.detector(pose, None)
    L = (target - source).norm(dim=-1).contiguous()
    s, t = drr.affine_inverse(source).contiguous(), drr.affine_inverse(target).contiguous()
NAMES = ("claimed", "wave 0 staged", "last wave staged", "behind staging barrier", "wave 0 out of units",
         "last wave out of units", "wave 0 walks done", "last walk done")
nb = 16 * 16 * 5
for name, fn in (("channel render", lambda: ops.siddon_forward_channels_bricks(drr.density, labels, C, s, t, L, (H, H))),
                 ("plain render, fp32 bricks (general kernel below 8 poses)", lambda: ops.siddon_forward_bricks(drr.density, s, t, L, (H, H), storage="f32"))):
    times = torch.zeros(nb * 17, dtype=torch.int32, device=dev)
    lib.ddrr_set_brick_times(ctypes.c_void_p(0))
    med, _ = timeit(fn)
    lib.ddrr_set_brick_times(ctypes.c_void_p(times.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.ddrr_set_brick_times(ctypes.c_void_p(0))
    tr = times.cpu().float().numpy()[nb:nb + 16 * nb].reshape(nb, 16) * 0.01
    tr = tr[tr[:, 0] > 0]
    if not len(tr):
        print(f"## {name}: no stamps (not on the general kernel)")
        continue
    end = tr[:, :8].max(axis=1)
    print(f"## {name}, {B} pose(s): kernel {med * 1e3:.0f} us (profile build, stamps off), {len(tr)} bricks with work, "
          f"{len(tr) / 256:.1f} per workgroup; a brick ends {end.mean():.1f} us after its claim was asked for")
    print("   us from the brick's start, mean: " + ", ".join(f"{n} {tr[:, k].mean():.1f}" for k, n in enumerate(NAMES)), flush=True)
