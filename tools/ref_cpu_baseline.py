"""The CPU baseline BASELINE.md section 2 promised: the UNMODIFIED reference
(`/root/reference/diffdrr`, loaded through the test-only shims of oracle/ref_shims) timed with
PyTorch on the host cores, on the inputs bench.py renders: SURVEY.md section 8(d) common scene,
perturbed poses of bench.perturbed_poses(seed=2), Siddon forward and forward + backward w.r.t.
the 6-DoF pose through NCC, one pose at a time (the reference materialises (B, N, M) tensors).
Prints CPU model, thread count and DRRs/s; runs wherever the reference checkout is (the build
container), never on the GPU box:  python tools/ref_cpu_baseline.py [--size 256] [--poses 3]"""
import argparse
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import perturbed_poses  # noqa: E402
from diffdrr_amd.data import centered_affine, noise_volume  # noqa: E402
from oracle import ref_loader  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--det", type=int, default=256)
ap.add_argument("--poses", type=int, default=3)
ap.add_argument("--patch", type=int, default=None, help="patch_size of the reference DRR (512^3)")
a = ap.parse_args()
torch.set_num_threads(os.cpu_count())
ref = ref_loader.load()
D, H = a.size, a.det
delx = 2.4 * (256 / H) * (D / 512)
vol = noise_volume(D, seed=0)
affine = centered_affine(vol.shape, (1.0, 1.0, 1.0))
reorient = torch.tensor([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
subject = ref.Subject(volume=ref.ScalarImage(vol[None], affine), density=ref.ScalarImage(vol[None], affine),
                      reorient=reorient, mask=None, fiducials=None)
drr = ref.DRR(subject, sdd=1020.0, height=H, delx=delx, patch_size=a.patch)
ncc = ref.NCC()
rot, xyz = perturbed_poses(32, 2, "cpu")
with torch.no_grad():
    base = drr(torch.zeros(1, 3), torch.tensor([[0.0, 850.0, 0.0]]), parameterization="euler_angles",
               convention="ZXY")
cpu = platform.processor() or ""
try:
    cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
except Exception:  # noqa: BLE001
    pass
print(f"# reference: {ref_loader.REFERENCE_ROOT} (unmodified), torch {torch.__version__} CPU, "
      f"{torch.get_num_threads()} threads on {os.cpu_count()} logical cores: {cpu}")
print(f"# {D}^3 fp32 noise volume -> {H}x{H} detector (delx {delx:g}, sdd 1020), Siddon, one pose per call")
fwd, fb = [], []
for b in range(a.poses):
    r = rot[b:b + 1].clone().requires_grad_()
    x = xyz[b:b + 1].clone().requires_grad_()
    with torch.no_grad():
        t0 = time.perf_counter()
        drr(r, x, parameterization="euler_angles", convention="ZXY")
        fwd.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    img = drr(r, x, parameterization="euler_angles", convention="ZXY")
    ncc(base, img).sum().backward()
    fb.append(time.perf_counter() - t0)
    print(f"pose {b}: forward {fwd[-1]:.2f} s, forward + backward (pose, NCC) {fb[-1]:.2f} s", flush=True)
# the first pose warms the allocator up; report the minimum like the reference's %timeit
print(f"forward:            {1 / min(fwd):.3f} DRRs/s (min of {len(fwd)}: {min(fwd):.2f} s)")
print(f"forward + backward: {1 / min(fb):.3f} DRRs/s (min of {len(fb)}: {min(fb):.2f} s)")
