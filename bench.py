"""bench.py -- DRRs/sec forward+backward on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`; SURVEY.md section 8d common scene): 512^3 fp32 volume
resident in HBM, 256x256 detector (delx 2.4, sdd 1020, AP), Siddon renderer, a
batch of 32 perturbed poses per GPU per step.  One step = pose parameters ->
`convert` -> fused ray generation (HIP) -> HIP Siddon forward (+ backward record) ->
per-pose NCC against a fixed target image -> backward to the 6-DoF pose parameters
(HIP pose-gradient kernel + autograd through the 4x4 pose chain).  With N > 1 every rank
renders its own 32 poses (weak scaling, volume replicated) and the per-pose
losses are all-gathered over RCCL each step.

Rank 0 prints ONE JSON line on stdout: the driver's contract plus
  "roofline":     the Siddon forward kernel's algorithmic HBM-read rate, timed
                  with HIP events around every launch inside the timed region,
  "cpu_baseline": the CPU oracle (a port of the reference's algorithm, OpenMP)
                  on a bounded sample of the same workload, rank 0, N = 1 only.
Diagnostics go to stderr.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffdrr_amd import DRR, NormalizedCrossCorrelation2d, ops  # noqa: E402
from diffdrr_amd.data import make_subject, noise_volume  # noqa: E402
from diffdrr_amd.pose import convert  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6290 measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def perturbed_poses(B, seed, device):
    """Base AP pose (0,0,0)/(0,850,0) + U(+-pi/4)^3 rad, U(+-30)^3 mm."""
    g = torch.Generator().manual_seed(seed)
    rot = (torch.rand(B, 3, generator=g) - 0.5) * (math.pi / 2)
    xyz = torch.tensor([0.0, 850.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 60.0
    return rot.to(device), xyz.to(device)


class KernelTimer:
    """HIP-event timing of every C-ABI launch inside the timed region, on the stream
    the kernels run on (ops._launch launches on torch's current stream)."""

    def __init__(self):
        self.enabled = False
        self.events = {}
        self._orig = ops._launch

    def install(self):
        def timed(name, device, *args):
            if self.enabled:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                self._orig(name, device, *args)
                e1.record()
                self.events.setdefault(name, []).append((e0, e1))
            else:
                self._orig(name, device, *args)

        ops._launch = timed

    def total_ms(self, name):
        ev = self.events.get(name, [])
        return sum(a.elapsed_time(b) for a, b in ev), len(ev)


def cpu_baseline(drr, rot, xyz, budget_s=12.0):
    """Oracle (C port of the reference algorithm, OpenMP over rays) forward +
    analytic backward on a bounded sample of the same workload."""
    import numpy as np

    import oracle

    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1)
        s = drr.affine_inverse(source).cpu().numpy()
        t = drr.affine_inverse(target).cpu().numpy()
        L = L.cpu().numpy()
    vol = drr.density.cpu().numpy()
    cores = os.cpu_count() or 1
    N = t.shape[1]
    go = np.ones((1, N), np.float32)
    # calibrate on a strip of rays, then size the sample to ~budget_s
    n0 = 4096
    t0 = time.perf_counter()
    oracle.siddon(vol, s[:1], t[:1, :n0], L[:1, :n0], grad_out=go[:, :n0])
    per_ray = (time.perf_counter() - t0) / n0
    n_drr = max(1, min(rot.shape[0], int(budget_s / (per_ray * N))))
    t0 = time.perf_counter()
    for b in range(n_drr):
        oracle.siddon(vol, s[b:b + 1], t[b:b + 1], L[b:b + 1], grad_out=go)
    dt = time.perf_counter() - t0
    return {
        "value": n_drr / dt, "unit": "DRRs/s", "cores": cores, "kind": "port",
        "sample": f"{n_drr} of the step's poses, 512^3 -> 256x256 Siddon fwd + analytic bwd "
                  f"(oracle/drr_oracle.c, OpenMP {cores} threads, {dt:.1f} s)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="poses per GPU per step")
    ap.add_argument("--size", type=int, default=512, help="volume edge (voxels)")
    ap.add_argument("--det", type=int, default=256, help="detector edge (pixels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" == RCCL on ROCm

    D, H, B = args.size, args.det, args.batch
    delx = 2.4 * (256 / H) * (D / 512)  # the detector always spans the volume's shadow
    subject = make_subject(noise_volume(D, seed=0), spacing=(1.0, 1.0, 1.0), orientation="AP")
    drr = DRR(subject, sdd=1020.0, height=H, delx=delx, renderer="siddon").to(device)
    ncc = NormalizedCrossCorrelation2d()
    rot0, xyz0 = perturbed_poses(B, seed=2 + rank, device=device)
    with torch.no_grad():
        base = drr(torch.zeros(1, 3, device=device), torch.tensor([[0.0, 850.0, 0.0]],
                                                                  device=device),
                   parameterization="euler_angles", convention="ZXY")
    rot = rot0.clone().requires_grad_()
    xyz = xyz0.clone().requires_grad_()
    gathered = torch.empty(world * B, device=device) if world > 1 else None
    pending = []  # the in-flight all_gather of the last step

    def step():
        rot.grad = None
        xyz.grad = None
        img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
        loss = ncc(base.expand(B, -1, -1, -1), img)  # (B,) one similarity per pose
        loss.sum().backward()
        if world > 1:
            # the 4 B/pose of losses travel on RCCL's own stream while the next step renders:
            # nothing on the compute stream waits for them before fence()
            pending[:] = [dist.all_gather_into_tensor(gathered, loss.detach(), async_op=True)]
        return loss

    timer = KernelTimer()
    timer.install()

    def fence():
        for work in pending:
            work.wait()
        pending.clear()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    assert torch.isfinite(loss).all() and torch.isfinite(rot.grad).all()

    t_max = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = t_max.item()
    total_drrs = world * B * args.steps

    if rank == 0:
        # algorithmic bytes of one forward launch:
        #   4 B per visited voxel + 20 B per ray (target 12 + img 4 + out 4) + 12 B per source
        with torch.no_grad():
            pose = convert(rot0, xyz0, parameterization="euler_angles", convention="ZXY")
            source, target = drr.detector(pose, None)
            L = (target - source).norm(dim=-1).contiguous()
            s_v = drr.affine_inverse(source).contiguous()
            t_v = drr.affine_inverse(target).contiguous()
            _, _, nvox = ops.siddon_forward(drr.density, s_v, t_v, L, count_voxels=True,
                                            det=(H, H))
        n_vox = int(nvox.sum().item())
        alg_bytes = 4 * n_vox + B * H * H * 20 + 12 * B
        # the forward of one step = every launch of the dominant forward entry point in that
        # step (the slab march renders the volume in Infinity-Cache-sized passes, one launch
        # each); bytes and time are both per step, so achieved = bytes / time of a forward
        fwd_name = max((n for n in timer.events if "forward" in n),
                       key=lambda n: timer.total_ms(n)[0])
        fwd_total, n_fwd = timer.total_ms(fwd_name)
        fwd_ms = fwd_total / args.steps
        bwd_ms = sum(timer.total_ms(n)[0] for n in timer.events if "backward" in n) / args.steps
        achieved = alg_bytes / (fwd_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this very command
        # (they cannot be collected from inside the process); only for the workload they
        # were measured on
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01", "traffic.json")
        if (D, H, B) == (512, 256, 32) and fwd_name == "ddrr_siddon_forward_bricks" \
                and os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f)["forward_record"]["hbm_bytes_per_launch"]
            traffic_src = "profiles/r01/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
        ms_per_step = dt / args.steps * 1e3
        log(f"[bench] step {ms_per_step:.3f} ms | {fwd_name} {fwd_ms:.3f} ms/step in "
            f"{n_fwd // args.steps} launch(es) | backward kernels {bwd_ms:.3f} ms | raygen "
            f"{timer.total_ms('ddrr_raygen_forward')[0] / args.steps:.3f} ms | host+torch remainder "
            f"{ms_per_step - fwd_ms - bwd_ms:.3f} ms | voxels/ray {n_vox / (B * H * H):.1f} "
            f"| {alg_bytes / B / 1e6:.1f} MB algorithmic per DRR")
        result = {
            "metric": f"DRRs/sec fwd+bwd, {D}^3 vol -> {H}^2 det, batched poses",  # BASELINE.json's at the defaults
            "value": total_drrs / dt,
            "unit": "DRRs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{D}^3 fp32 noise volume -> {H}x{H} detector, Siddon forward + "
                            f"backward w.r.t. 6-DoF pose (euler ZXY) through NCC, "
                            f"{B} perturbed poses per GPU per step",
                "volume": f"{D}x{D}x{D} f32 ({D ** 3 * 4 / 2 ** 20:.0f} MiB, replicated per GPU)",
                "detector": f"{H}x{H}",
                "batch_per_gpu": B,
                "global_batch": B * world,
                "parallelism": f"pose-sharded x{world}, all_gather of per-pose losses (RCCL)"
                               if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": fwd_name,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": fwd_ms,
                "launches_per_step": n_fwd // args.steps,
                "launches_timed": n_fwd,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(drr, rot0, xyz0)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
