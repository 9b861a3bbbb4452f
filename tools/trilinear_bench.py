"""BASELINE.json config 3: 512^3 volume, 512x512 detector, trilinear march with 512 samples
per ray, forward + backward incl. the volume gradient, one MI355X (development tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, ops
from diffdrr_amd.data import make_subject, noise_volume
from diffdrr_amd.renderers import get_alpha_minmax
from tools.kernel_sweep import poses, rays, timeit
dev = torch.device("cuda:0")
D, P = 512, 512
for H, Bs in ((512, (1, 4)), (256, (8,))):
    drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4 * 256 / H, renderer="trilinear").to(dev)
    V = drr.density
    for B in Bs:
        s, t, L = rays(drr, *poses(B, 2, dev))
        lo, hi = get_alpha_minmax(s, t, torch.tensor(V.shape, device=dev), 0.5, 1e-8)
        amin, amax = lo.min().reshape(1).contiguous(), hi.max().reshape(1).contiguous()
        go = torch.rand(B, H * H, device=dev)
        fb, _ = timeit(lambda: ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (H, H), n_points=P))
        fr, _ = timeit(lambda: ops.trilinear_forward(V, s, t, L, amin, amax, n_points=P, det=(H, H)), reps=3, warm=1)
        gb, _ = timeit(lambda: ops.trilinear_backward_volume_bricks(V.shape, s, t, L, go, amin, amax, (H, H), n_points=P))
        gr, _ = timeit(lambda: ops.trilinear_backward(V, s, t, L, go, amin, amax, n_points=P, want_rays=False, want_img=False, want_alpha=False, want_volume=True, det=(H, H)), reps=3, warm=1)
        fa, _ = timeit(lambda: ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (H, H), n_points=P, want_aux=True))
        _, aux = ops.trilinear_forward_bricks(V, s, t, L, amin, amax, (H, H), n_points=P, want_aux=True)
        ba, _ = timeit(lambda: ops.trilinear_backward_rays(aux, go, s, t, L, amin, amax, n_points=P))
        br, _ = timeit(lambda: ops.trilinear_backward(V, s, t, L, go, amin, amax, n_points=P, det=(H, H)), reps=3, warm=1)
        print(f"   ray/range gradients: forward+record bricks {fa:7.3f} ms + from record {ba:7.3f} ms | per-ray re-march {br:7.3f} ms", flush=True)
        # samples whose 8-cell touches the volume: count via a forward of a ones volume? use alpha range estimate
        nsamp = B * H * H * P
        print(f"{D}^3 det {H}^2 P={P} B={B}: forward bricks {fb:7.3f} ms | per-ray {fr:7.3f} ms   "
              f"volume-grad bricks {gb:7.3f} ms | per-ray (global atomics) {gr:7.3f} ms   "
              f"[{nsamp / fb / 1e6:.1f} Gsamples/s fwd]", flush=True)
