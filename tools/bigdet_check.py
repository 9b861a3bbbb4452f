"""Large and non-square detectors (up to 2048^2 = 4 M rays per pose): brick kernels vs per-ray kernels (development tool, GPU)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from diffdrr_amd import DRR, convert, ops
from diffdrr_amd.data import make_subject, noise_volume
dev = torch.device("cuda:0")
for (D, H, W, B, delx) in ((256, 1024, 768, 2, 0.5), (512, 2048, 2048, 1, 0.3), (200, 37, 1500, 3, 0.8)):
    vol = noise_volume(D, 0) if D != 200 else torch.rand(200, 130, 77)
    drr = DRR(make_subject(vol), sdd=1020.0, height=H, width=W, delx=delx).to(dev)
    rot = torch.tensor([[0.1, -0.2, 0.15]] * B, device=dev) * torch.arange(1, B + 1, device=dev)[:, None]
    xyz = torch.tensor([[5.0, 800.0, -3.0]] * B, device=dev)
    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        s = drr.affine_inverse(source).contiguous(); t = drr.affine_inverse(target).contiguous()
    V = drr.density
    ref, aux_ref, _ = ops.siddon_forward(V, s, t, L, want_aux=True)
    out, aux = ops.siddon_forward_bricks(V, s, t, L, (H, W), want_aux=True)
    e = ((out - ref).abs().max() / ref.abs().max()).item()
    go = torch.rand_like(ref)
    gv = ops.siddon_backward_volume_bricks(V.shape, s, t, L, go, (H, W))
    gv_ref = ops.siddon_backward_volume(V, s, t, L, go)
    ev = ((gv - gv_ref).abs().max() / gv_ref.abs().max()).item()
    img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
    e2 = ((img.reshape(B, -1) - ref).abs().max() / ref.abs().max()).item()
    print(f"vol {tuple(V.shape)} det {H}x{W} B {B}: bricks vs generic {e:.1e}, volgrad {ev:.1e}, DRR fused path {e2:.1e}", flush=True)
