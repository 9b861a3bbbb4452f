import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, ops
from diffdrr_amd.data import make_subject, noise_volume
from tools.kernel_sweep import poses, rays
dev = torch.device("cuda:0")
D, H = 512, 256
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4).to(dev)
V = drr.density
s, t, L = rays(drr, *poses(128, 2, dev))
ref = ops.siddon_forward(V, s, t, L, det=(H, H))[0]
for B in (128, 64, 65):
    out = ops.siddon_forward_bricks(V, s[:B], t[:B], L[:B], (H, H))[0]
    e = (out - ref[:B]).abs()
    per = e.amax(1) / ref[:B].abs().amax()
    worst = per.argmax().item()
    print(f"B={B}: max err {per.max().item():.2e} at pose {worst}; poses with err>3e-5: {(per > 3e-5).nonzero().flatten().tolist()}")
    pix = e[worst].argmax().item()
    print(f"   pixel {divmod(pix, H)} brick {out[worst, pix].item():.6f} generic {ref[worst, pix].item():.6f}  n bad pixels(>1e-4*max): {(e[worst] > 1e-4 * ref.abs().max()).sum().item()}")
# single-pose render of the worst pose
b = worst
o1 = ops.siddon_forward_bricks(V, s[b:b+1], t[b:b+1], L[b:b+1], (H, H))[0]
print("single-pose err", ((o1 - ref[b:b+1]).abs().max() / ref.abs().max()).item())
print("source", s[b].tolist(), "L range", L[b].min().item(), L[b].max().item())
from diffdrr_amd.plan import slab_plan
plan, shear = slab_plan(s, t, H, H)
o2 = ops.siddon_forward_slab(V, s, t, L, (H, H), plan, shear)[0]
print("slab vs generic, same pose:", ((o2[b] - ref[b]).abs().max() / ref.abs().max()).item(), " brick vs slab:", ((o2[b]-o1[0]).abs().max()/ref.abs().max()).item())
