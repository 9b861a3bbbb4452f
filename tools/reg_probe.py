"""Host/GPU time split of one registration iteration (development tool)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdrr_amd import DRR, NormalizedCrossCorrelation2d, Registration
from diffdrr_amd.data import synthetic_subject
dev = torch.device("cuda:0")
drr = DRR(synthetic_subject(512, kind="phantom", seed=0), sdd=1020.0, height=256, delx=2.4,
          stop_gradients_through_grid_sample=True).to(dev)
rot = torch.tensor([[0.1, -0.05, 0.08]], device=dev); xyz = torch.tensor([[10.0, 840.0, -8.0]], device=dev)
with torch.no_grad():
    gt = drr(torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 850.0, 0.0]], device=dev), parameterization="euler_angles", convention="ZXY")
reg = Registration(drr, rot.clone(), xyz.clone(), parameterization="euler_angles", convention="ZXY")
ncc = NormalizedCrossCorrelation2d()
opt = torch.optim.SGD([{"params": [reg._rotation], "lr": 5e-2}, {"params": [reg._translation], "lr": 1e2}], maximize=True)

def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    return t_host / n * 1e3, t_all / n * 1e3

def f_fwd():
    with torch.no_grad(): reg()
def f_fwd_grad(): reg()
def f_loss(): ncc(gt, reg()).sum()
def f_bwd():
    opt.zero_grad(); ncc(gt, reg()).sum().backward()
def f_full():
    opt.zero_grad(); ncc(gt, reg()).sum().backward(); opt.step()
for name, fn in (("forward no_grad", f_fwd), ("forward (grad mode)", f_fwd_grad), ("+ncc", f_loss), ("+backward", f_bwd), ("+opt.step", f_full)):
    h, a = timeit(fn)
    print(f"{name:22s} host {h:.3f} ms/it   wall {a:.3f} ms/it", flush=True)
