"""Empty-space skipping (development tool, GPU): the brick kernels do not look at the candidates of
a brick whose voxels are all zero -- air around the patient after the HU -> density transform
(reference data.py:214-227).  Times forward and forward + record at 512^3 / 256^2 / 32 poses for a
noise-filled body occupying all, half and a quarter of the volume."""
import sys, torch, math
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, convert, ops
from diffdrr_amd.data import make_subject, noise_volume
from tools.kernel_sweep import poses, rays, timeit
dev=torch.device('cuda:0')
D,H=512,256
for frac in (1.0, 0.5, 0.25):
    vol=noise_volume(D,0)
    if frac<1.0:
        # a centred body filling `frac` of the volume, air (exact zeros) around it
        r=int(round(D*frac**(1/3)/2)); c=D//2
        m=torch.zeros(D,D,D,dtype=torch.bool); m[c-r:c+r,c-r:c+r,c-r:c+r]=True
        vol=vol*m
    drr=DRR(make_subject(vol,(1.0,1.0,1.0),"AP",None), sdd=1020.0, height=H, delx=2.4).to(dev)
    s,t,L=rays(drr,*poses(32,2,dev))
    V=drr.density
    f,_=timeit(lambda: ops.siddon_forward_bricks(V,s,t,L,(H,H),storage='q16p'))
    a,_=timeit(lambda: ops.siddon_forward_bricks(V,s,t,L,(H,H),want_aux=True,storage='q16p'))
    print(f"body fills {frac:4.2f} of the 512^3 volume (zeros around it), 32 poses: forward {f:.3f} ms, forward + record {a:.3f} ms (default 16-bit packed bricks; tool timings include ~0.06 ms of host latency)", flush=True)
