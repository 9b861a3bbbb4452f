"""The reference's example CT shape -- 512 x 512 x 133, 119 labels, 200 x 200 detector
(introduction.ipynb:230-272) -- kernel launches only (rays precomputed): the channel render
(mask_to_channels, renderers.py:77-89; the general brick kernel's channel mode) next to the plain
render of the same volume on fp32 bricks, on 16-bit bricks from the packed copy (any D.z is
staged: brick_shared.h quad_load) and on the general kernel (tools switch variant -1); the
look-ahead stopping 2 / 3 / 4 rounds before the launch's end (dbg 8192 / 16384 / 24576; product: 3)
and off (4096).  --cube: 512^3 / 256^2 instead."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

tools.explib.use("exp")
from diffdrr_amd import DRR, _lib, convert, ops  # noqa: E402
from diffdrr_amd.data import make_subject  # noqa: E402
from diffdrr_amd.renderers import _labels_u8  # noqa: E402
from tools.kernel_sweep import timeit  # noqa: E402

dev = torch.device("cuda:0")
dims, C, H = (512, 512, 133), 119, 200
if "--cube" in sys.argv:
    dims, H = (512, 512, 512), 256
lib = _lib.get_lib()
print(f"# {torch.cuda.get_device_name(0)}: {dims} volume, {C} labels, {H}x{H} detector; kernel ms by back-to-back launches")
for which in ("synthetic",) if "--cube" in sys.argv else ("synthetic", "real"):
    g = torch.Generator().manual_seed(0)
    vol = torch.rand(*dims, generator=g)
    if which == "real":
        fx = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                  "reference_mask_ds2.npz"))
        mask = torch.from_numpy(fx["labels"]).repeat_interleave(2, 0).repeat_interleave(2, 1)[:512, :512]
        vol[mask == 0] *= (torch.rand(*dims, generator=g) < 0.3)[mask == 0]  # (air around the body: mostly 0)
    else:
        coarse = torch.randint(0, C, (16, 16, 8), generator=g)
        mask = coarse
        for ax, d in enumerate(dims):
            idx = (torch.arange(d) * coarse.shape[ax] // d).clamp_max(coarse.shape[ax] - 1)
            mask = mask.index_select(ax, idx)
    cube = "--cube" in sys.argv
    subject = make_subject(vol, spacing=(0.703, 0.703, 0.703 if cube else 2.5), mask=mask)
    drr = DRR(subject, sdd=1020.0, height=H, delx=1.6 if cube else 2.0).to(dev)
    (labels, _, _), = _labels_u8(drr.mask)
    V = drr.density
    for B in (1, 8, 32):
        rot = torch.zeros(B, 3, device=dev) + torch.linspace(0, 0.3, B, device=dev)[:, None]
        xyz = torch.tensor([[0.0, 850.0, 0.0]], device=dev).expand(B, 3).contiguous()
        with torch.no_grad():
            pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
            source, target = drr.detector(pose, None)
            L = (target - source).norm(dim=-1).contiguous()
            s_, t_ = drr.affine_inverse(source).contiguous(), drr.affine_inverse(target).contiguous()
            row = {}
            for name, var, dbg in (("", -2, 0), (", look 2", -2, 8192), (", look 4", -2, 24576),
                                   (", no look-ahead", -2, 4096), (", general kernel", -1, 0)):
                lib.cdll.ddrr_set_brick_variant(var)
                lib.cdll.ddrr_set_brick_debug(dbg)
                if var == -2 and dbg == 0:
                    for _ in range(2):
                        row["channels"], _ = timeit(
                            lambda: ops.siddon_forward_channels_bricks(V, labels, C, s_, t_, L, (H, H)))
                    c_new = ops.siddon_forward_channels_bricks(V, labels, C, s_, t_, L, (H, H))
                    # ... and from the volume's ready-packed words (ops.channel_words, cached per pair)
                    words = ops.channel_words(V, labels, C)
                    for _ in range(2):
                        row["channels, words"], _ = timeit(
                            lambda: ops.siddon_forward_channels_bricks(V, labels, C, s_, t_, L, (H, H), words=words))
                    # ... as the module calls it: the fingerprint comparison in front of every render
                    for _ in range(2):
                        row["channels, words + check"], _ = timeit(
                            lambda: ops.siddon_forward_channels_bricks(V, labels, C, s_, t_, L, (H, H),
                                                                       words=ops.channel_words(V, labels, C)))
                    c_words = ops.siddon_forward_channels_bricks(V, labels, C, s_, t_, L, (H, H), words=words)
                    assert float((c_words - c_new).abs().max()) <= 2e-6 * float(c_new.abs().max())
                for st in ("f32", "q16p"):
                    if (var == -1 or dbg != 0) and st != ("f32" if var == -1 else "q16p"):
                        continue  # (the general kernel: fp32 bricks; the look-ahead: packed bricks)
                    for _ in range(2):
                        row[f"plain {st}{name}"], _ = timeit(
                            lambda: ops.siddon_forward_bricks(V, s_, t_, L, (H, H), storage=st))
            lib.cdll.ddrr_set_brick_variant(-2)
            lib.cdll.ddrr_set_brick_debug(0)
            ref = ops.siddon_forward_channels(V, labels, C, s_, t_, L, det=(H, H))
            plain = ops.siddon_forward_bricks(V, s_, t_, L, (H, H), storage="q16p")[0]
        sc = float(ref.abs().max())
        e_new = float((c_new - ref).abs().max()) / sc
        e_sum = float((c_new.sum(1) - plain).abs().max() / plain.abs().max())
        print(f"{which} mask, B {B:2d}: " + " | ".join(f"{k} {v:6.3f}" for k, v in row.items()) +
              f" | channels / plain f32 = {row['channels'] / row['plain f32']:4.2f} x, / plain q16p = "
              f"{row['channels'] / row['plain q16p']:4.2f} x | channels vs per-ray kernel {e_new:.1e}, "
              f"sum vs plain {e_sum:.1e}", flush=True)
