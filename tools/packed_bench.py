"""16-bit bricks staged from the fp32 volume ("q16") or from the packed copy kept in the cached
workspace ("q16p"): kernel time per launch and agreement, 512^3 -> 256^2 (development tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.explib  # noqa: E402

if "product" not in sys.argv:
    tools.explib.use("exp")  # (the development build: its debug switches, argv[1])
from diffdrr_amd import DRR, _lib, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from tools.kernel_sweep import poses, rays, timeit  # noqa: E402

dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] != "product":
    _lib.get_lib().cdll.ddrr_set_brick_debug(int(sys.argv[1]))
    print(f"## debug flags {sys.argv[1]}")
D, H = 512, 256
drr = DRR(make_subject(noise_volume(D, 0)), sdd=1020.0, height=H, delx=2.4).to(dev)
V = drr.density
for B in (1, 2, 8, 32, 128):
    s, t, L = rays(drr, *poses(B, 2, dev))
    res = {}
    for aux in (False, True):
        for st in ("f32", "q16", "q16p"):
            fn = lambda: ops.siddon_forward_bricks(V, s, t, L, (H, H), want_aux=aux, storage=st)  # noqa: E731
            med, best = timeit(fn)
            res[(aux, st)] = (med, fn()[0])
        a, b = res[(aux, "q16")], res[(aux, "q16p")]
        err = float((a[1] - b[1]).abs().max() / b[1].abs().max())
        print(f"B {B:4d} {'fwd+record' if aux else 'forward   '}: f32 bricks {res[(aux, 'f32')][0]:7.3f} ms | "
              f"q16 {a[0]:7.3f} ms | q16p {b[0]:7.3f} ms ({100 * (b[0] / a[0] - 1):+.1f} %) | q16 vs q16p {err:.1e}",
              flush=True)
# first-call cost: ranges + packed copy
V2 = V.clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
ops.siddon_forward_bricks(V2, s[:1], t[:1], L[:1], (H, H), storage="q16p")
torch.cuda.synchronize()
t1 = time.perf_counter()
ops.siddon_forward_bricks(V2, s[:1], t[:1], L[:1], (H, H), storage="q16p")
torch.cuda.synchronize()
t2 = time.perf_counter()
ws, _ = ops.brick_workspace(V2, "q16p")
print(f"first call (ranges + packed copy + render) {1e3 * (t1 - t0):.3f} ms, second {1e3 * (t2 - t1):.3f} ms; "
      f"workspace {ws.numel() * 4 / 2**20:.1f} MiB for a {V.numel() * 4 / 2**20:.0f} MiB volume")
