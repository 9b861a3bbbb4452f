"""bench.py -- DRRs/sec on the BASELINE.json workloads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config headline|2|3|4|5]

`--config headline` (default; BASELINE.json `metric`; SURVEY.md section 8d common scene): 512^3
fp32 volume resident in HBM, 256x256 detector (delx 2.4, sdd 1020, AP), Siddon renderer, a batch
of 32 perturbed poses per GPU per step.  One step = pose parameters -> fused Euler pose kernel ->
fused ray generation -> HIP Siddon forward (+ backward record) -> per-pose NCC against a fixed
target image -> backward to the 6-DoF pose parameters.  With N > 1 every rank renders its own 32
poses (weak scaling, volume replicated) and the per-pose losses are all-gathered over RCCL.
Other configs (BASELINE.json `configs`):
  2  the same step at 256^3 -> 256^2, B = 32            (configs[1])
  3  512^3 -> 512^2 trilinear march, 512 samples per ray, forward + backward incl. the volume
     gradient, B = 1                                     (configs[2])
  4  2D/3D registration loop 512^3 -> 256^2, B = 1, NCC, Adam (configs[3]); metric: iterations/s
  5  4096 candidate poses, forward + per-pose NCC, pose-sharded over the ranks (strong
     scaling), all_gather of the losses                  (configs[4])

N > 1 without a launcher: the script re-executes itself under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL); under a
launcher (RANK / WORLD_SIZE set) it uses the ranks it is given.  `n_gpus` in the result is the
world size RCCL actually has.

Rank 0 prints ONE JSON line on stdout: the driver's contract plus
  "roofline":     the dominant kernel's algorithmic byte rate (SURVEY.md section 8d formulas), timed
                  with HIP events around every launch inside the timed region; "roofline.forward":
                  the forward-only kernel (the north star's target) on the step's own poses, primed
                  with 60 launches directly after the timed region, then 30 timed;
                  "roofline.forward_sweep": the same kernel at 512 of config 5's poses per launch,
                  with its parity; "roofline.kernels": every renderer kernel of the step with its
                  own algorithmic bytes and, where the counters were taken, "issue_bound": VALU issue
                  time, its share of the kernel and the useful fraction (profiles/rNN/issue_bound.json),
                  "roofline.forward_f32": the forward-only kernel on the same 32 poses with
                  brick_storage = "f32" (the volume's own values, as the reference gathers them),
  "configs":      (default run, one GPU) short runs of BASELINE configs 2, 3 (and its B = 4 timing,
                  "b4"), 4 (the registration loop as one HIP graph per iteration: iterations/s) and 5:
                  ms per step, dominant-kernel rate, parity; "ct": the reference's example geometry
                  on a CT-like 512 x 512 x 133 volume after transform_hu_to_density -- forward and
                  forward + record at 1 / 8 / 32 poses on the guarded 16-bit bricks and on fp32
                  bricks, the bricks the guard sent to the fp32 path, parity against the oracle;
                  "few_poses": the brick kernel at the headline size with 1 / 2 / 8 poses per launch on
                  the noise volume and on the phantom;
                  "sweep": config 5's figure -- with --gpus N the strong-scaling point (4096 poses
                  over the ranks) next to the weak-scaling headline,
  "cpu_baseline": the CPU oracle (a port of the reference's algorithm, OpenMP) on a bounded
                  sample of the same workload, rank 0, N = 1 only; "reference_cpu": the
                  UNMODIFIED reference on CPU torch, measured in the build container (it does
                  not exist on the GPU box), with its core count and source file,
  "parity":       images of the timed step, and the 6-DoF pose gradients its backward produced,
                  for sampled poses against the oracle's fp32 (the reference's arithmetic) and
                  fp64 (exact) chains; the same on a phantom volume (absolute bound).
Diagnostics go to stderr.
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    """HIP-event timing of C-ABI launches, on the stream the kernels run on (ops._launch launches
    on torch's current stream).  Inside the timed region only the dominant kernel carries events
    (`only`): a pair of events per launch costs the step 5-10 us of pipeline, and a step has seven
    launches; the other kernels are timed in a few steps of their own after it (`skip`: all but
    the dominant one)."""

    def __init__(self, ops, on_gpu=True):
        self.enabled = False
        self.only = None
        self.skip = None
        self.events = {}
        self.ops = ops
        self.on_gpu = on_gpu
        self._orig = ops._launch

    def install(self):
        def timed(name, device, *args):
            if self.enabled and ((self.only is not None and name != self.only) or name == self.skip):
                self._orig(name, device, *args)
            elif self.enabled and self.on_gpu:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                self._orig(name, device, *args)
                e1.record()
                self.events.setdefault(name, []).append((e0, e1))
            elif self.enabled:  # (host clock: the cpu harness test)
                t0 = time.perf_counter()
                self._orig(name, device, *args)
                self.events.setdefault(name, []).append((t0, time.perf_counter()))
            else:
                self._orig(name, device, *args)

        self.ops._launch = timed

    def total_ms(self, name):
        ev = self.events.get(name, [])
        if self.on_gpu:
            return sum(a.elapsed_time(b) for a, b in ev), len(ev)
        return sum((b - a) * 1e3 for a, b in ev), len(ev)


def voxel_rays(drr, rot, xyz):
    from diffdrr_amd.pose import convert

    with torch.no_grad():
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        source, target = drr.detector(pose, None)
        L = (target - source).norm(dim=-1).contiguous()
        return (drr.affine_inverse(source).contiguous(), drr.affine_inverse(target).contiguous(), L)


def _ncc_grad64(fixed, img, eps=1e-5):
    """d NCC(fixed, img) / d img in float64 (reference metrics.py:21-44)."""
    import numpy as np

    a, b = fixed.astype(np.float64).ravel(), img.astype(np.float64).ravel()
    s1, s2 = np.sqrt(a.var() + eps), np.sqrt(b.var() + eps)
    z1, z2 = (a - a.mean()) / s1, (b - b.mean()) / s2
    return (z1 - z2 * (z1 * z2).mean()) / (a.size * s2)


class OracleChain:
    """(g_rot, g_xyz) of sum(weights * DRR) for one pose through the ORACLE: rays generated in
    float64 torch on the CPU exactly as DRR.forward does (convert -> Detector -> affine_inverse),
    the oracle's analytic ray gradients (fp32 = the reference's arithmetic, fp64 = exact) chained
    back to the pose parameters by autograd of that float64 ray generation."""

    def __init__(self, drr):
        import copy

        import numpy as np

        from diffdrr_amd.pose import RigidTransform

        self.detector = copy.deepcopy(drr.detector).cpu().double()
        self.affine_inverse = RigidTransform(drr._affine_inverse.detach().cpu().double())
        self.vol = {np.float32: drr.density.detach().cpu().numpy()}
        self.vol[np.float64] = self.vol[np.float32].astype(np.float64)

    def __call__(self, rot_b, xyz_b, rays32, weights, dtype):
        """rays32: the fp32 voxel-space rays (s, t, L) the kernels rendered for this pose -- the
        oracle renders exactly those (in `dtype` arithmetic); the float64 ray generation only
        supplies the Jacobian d rays / d pose.  (On a noise volume an image changes by 1e-4 of its
        scale at single pixels when ray endpoints move by one fp32 ulp: rays gliding along voxel
        planes.  Parity is about the renderer, so both sides get the same rays.)"""
        import numpy as np

        import oracle
        from diffdrr_amd.pose import convert

        r64 = rot_b.detach().cpu().double().reshape(1, 3).requires_grad_()
        x64 = xyz_b.detach().cpu().double().reshape(1, 3).requires_grad_()
        pose = convert(r64, x64, parameterization="euler_angles", convention="ZXY")
        source, target = self.detector(pose, None)
        L = (target - source).norm(dim=-1)
        s, t = self.affine_inverse(source), self.affine_inverse(target)
        o = oracle.siddon(self.vol[dtype], *(np.asarray(a, dtype) for a in rays32),
                          grad_out=np.asarray(weights, dtype).reshape(1, -1))
        as64 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))  # noqa: E731
        ((as64(o["g_source"]) * s).sum() + (as64(o["g_target"]) * t).sum()
         + (as64(o["g_img"]).reshape(L.shape) * L).sum()).backward()
        return r64.grad.numpy().ravel(), x64.grad.numpy().ravel(), o["out"].reshape(-1)


def cpu_baseline_and_parity(drr, rot, xyz, images, g_rot, g_xyz, base, det, budget_s=12.0, n_parity=4):
    """Oracle (C port of the reference algorithm, OpenMP over rays) forward + analytic backward
    on a bounded sample of the step's poses: the timing is `cpu_baseline`.  `parity`: the images
    the timed step produced and the pose gradients its backward left in rot.grad / xyz.grad,
    for the first poses, against the oracle chains in fp32 (the reference's arithmetic) and
    fp64 (exact)."""
    import numpy as np

    import oracle

    s_d, t_d, L_d = voxel_rays(drr, rot, xyz)
    s, t, L = s_d.cpu().numpy(), t_d.cpu().numpy(), L_d.cpu().numpy()
    vol = drr.density.cpu().numpy()
    D = vol.shape[0]
    cores = os.cpu_count() or 1
    B, N = t.shape[0], t.shape[1]
    go = np.ones((1, N), np.float32)
    # calibrate on a strip of rays, then size the sample to ~budget_s
    n0 = 4096
    t0 = time.perf_counter()
    oracle.siddon(vol, s[:1], t[:1, :n0], L[:1, :n0], grad_out=go[:, :n0])
    per_ray = (time.perf_counter() - t0) / n0
    n_drr = max(1, min(B, int(budget_s / (per_ray * N))))
    t0 = time.perf_counter()
    for b in range(n_drr):
        oracle.siddon(vol, s[b:b + 1], t[b:b + 1], L[b:b + 1], grad_out=go)
    dt = time.perf_counter() - t0
    baseline = {
        "value": n_drr / dt, "unit": "DRRs/s", "cores": cores, "kind": "port",
        "sample": f"{n_drr} of the step's poses, {D}^3 -> {det}x{det} Siddon fwd + analytic bwd "
                  f"(oracle/drr_oracle.c, OpenMP {cores} threads, {dt:.1f} s)",
    }
    parity = pose_parity(drr, rot, xyz, images, g_rot, g_xyz, base, min(n_parity, B))
    return baseline, parity


def pose_parity(drr, rot, xyz, images, g_rot, g_xyz, base, n_par):
    """Images and 6-DoF pose gradients of a step (loss = sum of per-pose NCC against `base`)
    against the oracle chains.  rel err = max |a - b| / max |b| (image-normalised)."""
    import numpy as np

    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-300))  # noqa
    fx = base.reshape(-1).cpu().numpy()
    chain = OracleChain(drr)
    out = {k: 0.0 for k in ("fwd_rel_err", "fwd_rel_err_vs_fp64", "ref_fp32_fwd_rel_err_vs_fp64",
                            "pose_grad_rel_err_vs_fp64", "ref_fp32_pose_grad_rel_err_vs_fp64")}
    for b in range(n_par):
        mine = images[b].reshape(-1).cpu().numpy()
        # d loss / d image of this pose, at the exact image (the step's own weights differ from
        # these by the image error, 1e-5)
        rays32 = tuple(a.cpu().numpy() for a in voxel_rays(drr, rot[b:b + 1], xyz[b:b + 1]))
        _, _, img64 = chain(rot[b], xyz[b], rays32, np.zeros(mine.size), np.float64)
        W = _ncc_grad64(fx, img64)
        gr64, gx64, img64 = chain(rot[b], xyz[b], rays32, W, np.float64)
        gr32, gx32, img32 = chain(rot[b], xyz[b], rays32, W, np.float32)
        # against the reference's fp32 image where that image is itself within 1e-4 of the exact
        # one (rays gliding along voxel planes: the fp32 reference is up to 6e-4 off at single
        # pixels of these scenes); everywhere against the exact image
        ok = np.abs(img32 - img64) <= 1e-4 * np.abs(img32).max()
        out["fwd_rel_err"] = max(out["fwd_rel_err"],
                                 float(np.abs(mine - img32)[ok].max() / np.abs(img32).max()))
        out["ref_off_pixels"] = out.get("ref_off_pixels", 0) + int((~ok).sum())
        out["fwd_rel_err_all_pixels"] = max(out.get("fwd_rel_err_all_pixels", 0.0), rel(mine, img32))
        out["fwd_rel_err_vs_fp64"] = max(out["fwd_rel_err_vs_fp64"], rel(mine, img64))
        out["ref_fp32_fwd_rel_err_vs_fp64"] = max(out["ref_fp32_fwd_rel_err_vs_fp64"], rel(img32, img64))
        truth = np.concatenate([gr64, gx64 * 100.0])  # (mm -> comparable scale with radians)
        ours = np.concatenate([g_rot[b].cpu().numpy(), g_xyz[b].cpu().numpy() * 100.0])
        ref = np.concatenate([gr32, gx32 * 100.0])
        out["pose_grad_rel_err_vs_fp64"] = max(out["pose_grad_rel_err_vs_fp64"], rel(ours, truth))
        out["ref_fp32_pose_grad_rel_err_vs_fp64"] = max(out["ref_fp32_pose_grad_rel_err_vs_fp64"],
                                                         rel(ref, truth))
    out["poses"] = n_par
    return out


def sweep_parity(drr, fixed, rot, xyz, images, vals, picks, eps=1e-5):
    """Sampled poses of a sweep launch against the oracle: the launch's images vs the oracle's
    fp32 / fp64 renders, its per-pose NCC vs NCC (float64) of the oracle's images."""
    import numpy as np

    import oracle

    def ncc64(a, b):
        z = lambda x: (x - x.mean()) / np.sqrt(x.var() + eps)  # noqa: E731
        return float((z(a.astype(np.float64)) * z(b.astype(np.float64))).mean())

    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())  # noqa
    vol = drr.density.cpu().numpy()
    fx = fixed.reshape(-1).cpu().numpy()
    res = {"poses": [int(b) for b in picks], "fwd_rel_err": 0.0, "fwd_rel_err_vs_fp64": 0.0,
           "ref_fp32_fwd_rel_err_vs_fp64": 0.0, "ncc_abs_err": 0.0, "ncc_abs_err_vs_fp64": 0.0}
    for b in picks:
        s, t, L = voxel_rays(drr, rot[b:b + 1], xyz[b:b + 1])
        a32 = (vol, s.cpu().numpy(), t.cpu().numpy(), L.cpu().numpy())
        r32 = oracle.siddon(*a32)["out"].reshape(-1)
        r64 = oracle.siddon(*(a.astype(np.float64) for a in a32))["out"].reshape(-1)
        mine = images[b].reshape(-1).cpu().numpy()
        ok = np.abs(r32 - r64) <= 1e-4 * np.abs(r32).max()  # (the fp32 reference itself within 1e-4)
        res["fwd_rel_err"] = max(res["fwd_rel_err"], float(np.abs(mine - r32)[ok].max() / np.abs(r32).max()))
        res["ref_off_pixels"] = res.get("ref_off_pixels", 0) + int((~ok).sum())
        res["fwd_rel_err_all_pixels"] = max(res.get("fwd_rel_err_all_pixels", 0.0), rel(mine, r32))
        res["fwd_rel_err_vs_fp64"] = max(res["fwd_rel_err_vs_fp64"], rel(mine, r64))
        res["ref_fp32_fwd_rel_err_vs_fp64"] = max(res["ref_fp32_fwd_rel_err_vs_fp64"], rel(r32, r64))
        v = float(vals[b].item())
        res["ncc_abs_err"] = max(res["ncc_abs_err"], abs(v - ncc64(fx, r32)))
        res["ncc_abs_err_vs_fp64"] = max(res["ncc_abs_err_vs_fp64"], abs(v - ncc64(fx, r64)))
    return res


REFERENCE_CPU = {  # the UNMODIFIED reference on CPU torch (tools/ref_cpu_baseline.py, build container)
    512: {"value": 0.071, "forward_only": 0.208}, 256: {"value": 0.192, "forward_only": 0.468}}


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def traffic_record(kind):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (they
    cannot be collected from inside the process): newest profiles/rNN/traffic.json."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for r in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        path = os.path.join(pdir, r, "traffic.json")
        if os.path.exists(path):
            with open(path) as f:
                rec = json.load(f)
            if kind in rec:
                best = (rec[kind]["hbm_bytes_per_launch"], f"profiles/{r}/traffic.json")
    return best


class Runtime:
    """What every config run of one bench process shares: the ranks, the device, the launch timer."""

    def __init__(self, args):
        self.on_gpu = args.device == "cuda"
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = torch.device("cuda", local_rank) if self.on_gpu else torch.device("cpu")
        if self.on_gpu:
            torch.cuda.set_device(self.device)
        if os.environ.get("DDRR_BENCH_HOOK"):  # test harness only: e.g. route ops to the host emulation
            import importlib

            importlib.import_module(os.environ["DDRR_BENCH_HOOK"])
        self.dist = None
        self.rccl_ranks = 1
        if self.world > 1:
            import torch.distributed as dist

            self.dist = dist
            if self.on_gpu:
                dist.init_process_group(backend="nccl", device_id=self.device)  # "nccl" == RCCL on ROCm
            else:
                dist.init_process_group(backend="gloo")
            self.world = dist.get_world_size()
            # proof that the N ranks of this run each hold a device and reach each other: an
            # all_reduce of ones ON the device (RCCL over xGMI on the GPU box, gloo on the harness)
            ones = torch.ones(1, device=self.device)
            dist.all_reduce(ones)
            self.rccl_ranks = int(ones.item())
            assert self.rccl_ranks == self.world, (self.rccl_ranks, self.world)
        from diffdrr_amd import ops

        self.timer = KernelTimer(ops, self.on_gpu)
        self.timer.install()
        self.fence_wait = {"all_gather_s": 0.0, "barrier_s": 0.0}

    def fence(self, pending=()):
        t_a = time.perf_counter()
        for work in pending:
            work.wait()
        if self.on_gpu:
            torch.cuda.synchronize()
        t_b = time.perf_counter()
        if self.world > 1:
            self.dist.barrier()
        if self.on_gpu:
            torch.cuda.synchronize()
        # (what this rank waited for its own work + the losses' all_gather, and then for the others)
        self.fence_wait["all_gather_s"], self.fence_wait["barrier_s"] = t_b - t_a, time.perf_counter() - t_b


# (steps, warmup, priming steps) of a config: the full run, and the short run whose summary the
# default line carries as "configs" (each short timed region >= ~0.3 s on one MI355X)
FULL = {"headline": (400, 10, 60), "2": (800, 10, 60), "3": (400, 10, 60), "4": (1500, 20, 0), "5": (5, 1, 1)}
SHORT = {"2": (300, 5, 40), "3": (150, 5, 30), "4": (1000, 20, 0), "5": (2, 0, 1)}


def issue_bound_record(kernel_key, kernel_ms, visits=None):
    """What binds a brick kernel, from the committed counter passes of this round (they cannot be
    collected from inside the process): newest profiles/rNN/issue_bound.json (tools/issue_bound.py
    wrote it from rocprofv3 --pmc SQ_INSTS_VALU ... and the walk counters of the profile build).
    issue_ms = VALU wave-instructions per launch x the measured issue time per wave-instruction
    and SIMD / SIMDs; useful_frac = (walk steps x lanes that hold a live ray x instructions per
    step) / (all VALU lane-slots issued)."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for r in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        path = os.path.join(pdir, r, "issue_bound.json")
        if os.path.exists(path):
            with open(path) as f:
                rec = json.load(f)
            if kernel_key in rec.get("kernels", {}):
                best = (rec, rec["kernels"][kernel_key], f"profiles/{r}/issue_bound.json")
    if best is None:
        return None
    rec, k, src = best
    simds = rec["simds"]
    issue_ms = k["valu_wave_insts"] * rec["ns_per_valu_wave_inst_per_simd"] / simds * 1e-6
    out = {"bound": "VALU issue (4 waves per SIMD)", "valu_wave_insts_per_launch": k["valu_wave_insts"],
           "ns_per_wave_inst_per_simd": rec["ns_per_valu_wave_inst_per_simd"], "simds": simds,
           "issue_ms": issue_ms, "issue_frac": issue_ms / kernel_ms if kernel_ms else None,
           "profiled_kernel_ms": k.get("kernel_ms"),
           "source": src + " (separate rocprofv3 --pmc passes + the profile build's walk counters, "
                           "committed; not measured in this run)"}
    if "walk_wave_steps" in k and visits:
        # lanes of the walk that hold a live ray = voxel visits of this launch (counted by the
        # bench) / (64 x wave-steps of the profiled launch of the same workload)
        lanes = visits / (64.0 * k["walk_wave_steps"])
        useful = k["walk_wave_steps"] * lanes * k["insts_per_step"]
        out.update(walk_wave_steps=k["walk_wave_steps"], walk_useful_lane_frac=lanes,
                   insts_per_step=k["insts_per_step"], hits=k.get("hits"), batches=k.get("batches"),
                   useful_frac=useful / k["valu_wave_insts"],
                   useful_frac_of_kernel=useful / k["valu_wave_insts"] * (issue_ms / kernel_ms) if kernel_ms else None)
    if "lds_busy_ms" in k:
        # SQ_LDS_IDX_ACTIVE summed over the CUs / 256 CUs / 2.4 GHz: how long the LDS pipe of a CU is busy
        out.update(lds_wave_insts_per_launch=k.get("lds_wave_insts"), lds_busy_ms=k["lds_busy_ms"],
                   lds_busy_frac=k["lds_busy_ms"] / kernel_ms if kernel_ms else None)
    return out


def run_config(cfg, args, rt, short=False, batch=None, parity=True):
    """One timed run of a BASELINE config on the ranks of `rt`; -> the result dict (rank 0) or
    None.  short: few steps, no CPU baseline -- the summaries of the default line (batch: poses per
    step of such a run, e.g. config 3 at B = 4; parity = False: timing only)."""
    on_gpu, world, rank, device, dist = rt.on_gpu, rt.world, rt.rank, rt.device, rt.dist
    fence, fence_wait, timer = rt.fence, rt.fence_wait, rt.timer
    timer.events = {}

    from diffdrr_amd import DRR, NormalizedCrossCorrelation2d, Registration, ops
    from diffdrr_amd import dist as ddist
    from diffdrr_amd.data import make_subject, noise_volume, synthetic_subject

    top = not short  # the run the command line asked for
    sized = top or (cfg == "5" and args.sweep_poses is not None)  # (the harness test's small sweep)
    D = (args.size if sized else None) or (256 if cfg == "2" else 512)
    H = (args.det if sized else None) or (512 if cfg == "3" else 256)
    B = (args.batch if top else batch) or {"headline": 32, "2": 32, "3": 1, "4": 1, "5": 4096}[cfg]
    if short and cfg == "5" and args.sweep_poses is not None:
        B = args.sweep_poses
    steps, warmup, prime = SHORT[cfg] if short else FULL[cfg]
    if short and batch:
        steps = max(20, steps // batch)
    if top and args.steps is not None:
        steps = args.steps
    if top and args.warmup is not None:
        warmup = args.warmup
    delx = 2.4 * (256 / H) * (D / 512)  # the detector always spans the volume's shadow
    ncc = NormalizedCrossCorrelation2d()
    extra = {}
    images = None

    def set_storage(d):
        if args.storage is not None and hasattr(d.renderer, "brick_storage"):
            d.renderer.brick_storage = args.storage
        if args.packed_record and hasattr(d.renderer, "packed_record"):
            d.renderer.packed_record = True
        return d

    if cfg in ("headline", "2"):
        subject = make_subject(noise_volume(D, seed=0), spacing=(1.0, 1.0, 1.0), orientation="AP")
        drr = set_storage(DRR(subject, sdd=1020.0, height=H, delx=delx, renderer="siddon").to(device))
        if args.fused_max_poses is not None:
            drr.FUSED_NCC_MAX_POSES = args.fused_max_poses
        rot0, xyz0 = perturbed_poses(B, seed=2 + rank, device=device)
        with torch.no_grad():
            base = drr(torch.zeros(1, 3, device=device), torch.tensor([[0.0, 850.0, 0.0]], device=device),
                       parameterization="euler_angles", convention="ZXY")
        rot = rot0.clone().requires_grad_()
        xyz = xyz0.clone().requires_grad_()
        gathered = torch.empty(world * B, device=device) if world > 1 else None
        pending = []  # the in-flight all_gather of the last step
        keep = {}
        one = torch.ones((), device=device)  # d (sum of the per-pose values) / d itself, ready-made

        def step():
            rot.grad = None
            xyz.grad = None
            if args.unfused:
                img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY")
                loss = ncc(base.expand(B, -1, -1, -1), img)  # (B,) one similarity per pose
                loss.sum().backward()
            else:
                # the same objective through DRR.ncc: pose -> rays, image-from-record + NCC and NCC
                # backward -> pose parameters as three fused launches around the brick kernel; the
                # batch's sum comes out of the forward epilogue and its gradient goes back in as one
                # ready-made value: no reduction, no fill
                total = drr.ncc(base, rot, xyz, convention="ZXY", eps=ncc.eps, reduction="sum")
                total.backward(gradient=one)
                # (the per-pose values: what the epilogue wrote on its way to the sum)
                loss = drr.ncc_per_pose
            if world > 1:
                # the 4 B/pose of losses travel on RCCL's own stream while the next step renders:
                # nothing on the compute stream waits for them before fence()
                pending[:] = [dist.all_gather_into_tensor(gathered, loss.detach(), async_op=True)]
            return loss

        units_per_step, unit, scaling = world * B, "DRRs/s", "weak"
        metric = f"DRRs/sec fwd+bwd, {D}^3 vol -> {H}^2 det, batched poses"
        workload = (f"{D}^3 fp32 noise volume -> {H}x{H} detector, Siddon forward + backward w.r.t. "
                    f"6-DoF pose (euler ZXY) through NCC, {B} perturbed poses per GPU per step")
        dominant = "ddrr_siddon_forward_bricks"
    elif cfg == "3":
        subject = make_subject(noise_volume(D, seed=0), spacing=(1.0, 1.0, 1.0), orientation="AP")
        drr = DRR(subject, sdd=1020.0, height=H, delx=delx, renderer="trilinear").to(device)
        drr.density.requires_grad_()   # reconstruction: the volume is the parameter
        rot0, xyz0 = perturbed_poses(B, seed=2 + rank, device=device)
        go = torch.rand(B, 1, H, H, generator=torch.Generator().manual_seed(7)).to(device)
        pending, keep = [], {}
        P = 512

        def step():
            drr.density.grad = None
            img = drr(rot0, xyz0, parameterization="euler_angles", convention="ZXY", n_points=P)
            (img * go).sum().backward()
            keep["img"] = img
            return img.detach().sum().reshape(1)

        units_per_step, unit, scaling = world * B, "DRRs/s", "weak"
        metric = f"DRRs/sec fwd+bwd incl. volume gradient, {D}^3 vol -> {H}^2 det, trilinear {P} samples/ray"
        workload = (f"{D}^3 fp32 noise volume -> {H}x{H} detector, trilinear march {P} samples per ray, "
                    f"forward + backward w.r.t. the volume, {B} pose(s) per GPU per step")
        dominant = "ddrr_trilinear_forward_bricks"
    elif cfg == "4":
        drr = set_storage(DRR(synthetic_subject(D, kind="phantom", seed=0), sdd=1020.0, height=H, delx=delx,
                              stop_gradients_through_grid_sample=True).to(device))
        true_rot = torch.zeros(1, 3, device=device)
        true_xyz = torch.tensor([[0.0, 850.0, 0.0]], device=device)
        with torch.no_grad():
            gt = drr(true_rot, true_xyz, parameterization="euler_angles", convention="ZXY")
        g = torch.Generator().manual_seed(1 + rank)
        r0 = true_rot + ((torch.rand(1, 3, generator=g) - 0.5) * 0.4).to(device)   # +-0.2 rad
        x0 = true_xyz + ((torch.rand(1, 3, generator=g) - 0.5) * 60.0).to(device)  # +-30 mm
        # Adam (registration.ipynb:569 learning rates); capturable: the whole iteration is one
        # HIP graph (diffdrr_amd.registration.GraphedIteration), falling back to the eager loop
        from diffdrr_amd import GraphedIteration

        reg = Registration(drr, r0.clone(), x0.clone(), parameterization="euler_angles", convention="ZXY")
        if args.torch_adam or not on_gpu:
            opt = torch.optim.Adam([{"params": [reg._rotation], "lr": 1e-1},
                                    {"params": [reg._translation], "lr": 5e0}], maximize=True,
                                   capturable=on_gpu, fused=on_gpu)  # fused: two launches per group
            opt_name = "torch.optim.Adam(1e-1 / 5e0, fused)"
        else:
            # the same update rule, both groups in one launch (four with torch's fused Adam)
            from diffdrr_amd import PoseAdam

            opt = PoseAdam(reg._rotation, reg._translation, 1e-1, 5e0, maximize=True)
            opt_name = "diffdrr_amd.PoseAdam(1e-1 / 5e0) = torch.optim.Adam's update, both groups in one launch"
        pending, keep = [], {}
        try:
            if not on_gpu:
                raise RuntimeError("no HIP graphs on the cpu harness")
            crit = ncc
            if args.criterion != "ncc":
                from diffdrr_amd.metrics import (GradientNormalizedCrossCorrelation2d,
                                                 MultiscaleNormalizedCrossCorrelation2d)
                # ("gradient": the class's own defaults, metrics.py:99 -- whole-image NCC of the blurred Sobel
                #  pair; with patch_size=9 the phantom's smooth projections leave most windows flat and the
                #  notebook's learning rates walk away from the optimum: a timing run only)
                crit = {"multiscale": lambda: MultiscaleNormalizedCrossCorrelation2d([13, None], [0.5, 0.5]),
                        "gradient": lambda: GradientNormalizedCrossCorrelation2d(),
                        "gradient_patch9": lambda: GradientNormalizedCrossCorrelation2d(patch_size=9, sigma=1.0),
                        }[args.criterion]()
            graphed = GraphedIteration(reg, crit, opt, gt)
            extra["registration"] = {"hip_graph": True, "criterion": args.criterion}
        except Exception as exc:  # noqa: BLE001
            log(f"[bench] config 4: eager loop ({type(exc).__name__}: {exc})")
            graphed = None
            extra["registration"] = {"hip_graph": False}

        def step():
            # (the loop keeps running at the optimum once converged: same work per iteration)
            if graphed is not None:
                return graphed().reshape(1)
            opt.zero_grad()
            loss = ncc(gt, reg()).sum()
            loss.backward()
            opt.step()
            return loss.detach().reshape(1)

        units_per_step, unit, scaling = world, "iterations/s", "weak"
        metric = f"registration iterations/sec, {D}^3 vol -> {H}^2 det, SE(3) gradient ascent on NCC"
        workload = (f"{D}^3 fp32 phantom volume -> {H}x{H} detector, Siddon "
                    f"(stop_gradients_through_grid_sample), Registration + NCC + {opt_name}, "
                    f"one pose per GPU, one HIP graph per iteration")
        dominant = "ddrr_siddon_forward_bricks"
    else:  # "5": the candidate-pose sweep, sharded (strong scaling)
        subject = make_subject(noise_volume(D, seed=0), spacing=(1.0, 1.0, 1.0), orientation="AP")
        drr = set_storage(DRR(subject, sdd=1020.0, height=H, delx=delx, renderer="siddon").to(device))
        rot0, xyz0 = perturbed_poses(B, seed=2, device=device)  # the same candidates on every rank
        with torch.no_grad():
            fixed = drr(torch.zeros(1, 3, device=device), torch.tensor([[0.0, 850.0, 0.0]], device=device),
                        parameterization="euler_angles", convention="ZXY")
        pending, keep = [], {}

        def step():
            vals = ddist.sweep(drr, ncc, fixed, rot0, xyz0, chunk=512)  # all_gather inside
            keep["vals"] = vals
            return vals

        units_per_step, unit, scaling = B, "DRRs/s", "strong"
        metric = f"DRRs/sec forward + per-pose NCC, {D}^3 vol -> {H}^2 det, {B} candidate poses sharded"
        workload = (f"{D}^3 fp32 noise volume -> {H}x{H} detector, Siddon forward + NCC of {B} candidate "
                    f"poses sharded over the ranks, all_gather of the per-pose similarities")
        dominant = "ddrr_siddon_forward_bricks"

    # The collector stays out of the timed region, as in `timeit`: a generation-2 pass of a process
    # that has imported torch takes ~50 ms -- 40 steps' worth -- and WHERE it falls depends on the
    # allocation count of everything before it (the driver's 20-step run of round 6's first build
    # read 3.38 ms per step where 400 steps read 1.50: profiles/r06/gc_pause_in_the_timed_region.txt).
    # Collected HERE, in front of the priming steps: the pause idles the board, and an idle board
    # needs ~0.1 s under load to clock up again (collected directly in front of the timed region, the
    # same 20-step run read kernels of 1.56 instead of 1.41 ms).
    gc.collect()
    gc.disable()
    if on_gpu:
        # set-up, not measurement: the first launches on a fresh box allocate (caching allocator,
        # the 16-bit bricks of the volume) and run at boot clocks; the driver's own --warmup may be
        # as short as 3 steps (60 steps = 0.1 s: the clocks of an idle board need about that long
        # under load; measured 1.576 vs 1.540 ms per step with 10)
        for _ in range(prime):
            step()
        fence(pending)
        pending.clear()
    for _ in range(warmup):
        step()
    fence(pending)
    pending.clear()
    timer.enabled, timer.only = True, (dominant if cfg != "4" else None)
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    fence(pending)
    dt = time.perf_counter() - t0
    timer.enabled, timer.only = False, None
    # (the collector stays off for the legs that follow -- forward-only launches between HIP events,
    # the other configs: every run_config collects once, up front)
    assert torch.isfinite(last).all()
    # the step's other kernels: a few steps of their own, every launch but the dominant one timed
    census_steps = 0
    if cfg != "4":
        census_steps = min(steps, 30)
        timer.enabled, timer.skip = True, dominant
        for _ in range(census_steps):
            step()
        fence(pending)
        pending.clear()
        timer.enabled, timer.skip = False, None
    if cfg in ("headline", "2"):
        assert torch.isfinite(rot.grad).all() and torch.isfinite(xyz.grad).all()
        with torch.no_grad():  # (the fused step writes no image: the same poses, rendered once more)
            images = drr(rot0, xyz0, parameterization="euler_angles", convention="ZXY")
        fused_used = not args.unfused and B <= drr.FUSED_NCC_MAX_POSES
        extra["step"] = ("DRR.ncc, fused (at most %d poses per call): ddrr_pose_raygen_forward, the brick kernel "
                         "with its record, ddrr_siddon_ncc_forward; backward: ddrr_siddon_ncc_backward_pose"
                         % drr.FUSED_NCC_MAX_POSES if fused_used else
                         "DRR.forward + NormalizedCrossCorrelation2d through autograd (what DRR.ncc composes "
                         "for more than %d poses per call)" % drr.FUSED_NCC_MAX_POSES)
    if cfg == "4":
        extra["registration"].update(
            ncc_after=float(last.item()), iterations=warmup + steps,
            rot_error_rad=float((reg.rotation.detach() - true_rot).abs().max().item()),
            xyz_error_mm=float((reg.translation.detach() - true_xyz).abs().max().item()),
            start="U(+-0.2 rad, +-30 mm) from the true pose")

    # The north star's target kernel, forward ONLY, on the step's own poses: directly after the
    # timed region (same clocks, same caches), primed like the step -- 60 launches, then 30 timed
    # (config 5's 512-pose launches: 4 + 6).
    f_ms, f_n, f_poses = None, 0, 0
    if on_gpu and cfg in ("headline", "2", "5"):
        nposes5 = min(512, -(-B // world))
        with torch.no_grad():
            fr, fx = (rot0, xyz0) if cfg != "5" else (rot0[:nposes5], xyz0[:nposes5])
            n_prime, n_timed = ((60, 30) if not short else (40, 20)) if cfg != "5" else (4, 6)
            for _ in range(n_prime):
                drr(fr, fx, parameterization="euler_angles", convention="ZXY")
            before = len(timer.events.get(dominant, []))
            timer.enabled, timer.only = True, dominant
            for _ in range(n_timed):
                drr(fr, fx, parameterization="euler_angles", convention="ZXY")
            torch.cuda.synchronize()
            timer.enabled, timer.only = False, None
            ev = timer.events[dominant][before:]
            f_ms, f_n, f_poses = sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev), int(fr.shape[0])
            del timer.events[dominant][before:]

    # The same launches from the volume's OWN fp32 values (brick_storage = "f32": what the reference
    # gathers, renderers.py:159-164): the value-faithful figure next to the 16-bit default's.
    f32_ms, f32_n = None, 0
    if on_gpu and cfg == "headline" and hasattr(drr.renderer, "brick_storage") and args.storage is None:
        had = drr.renderer.brick_storage
        drr.renderer.brick_storage = "f32"
        try:
            with torch.no_grad():
                for _ in range(40):
                    drr(rot0, xyz0, parameterization="euler_angles", convention="ZXY")
                before = len(timer.events.get(dominant, []))
                timer.enabled, timer.only = True, dominant
                for _ in range(20):
                    drr(rot0, xyz0, parameterization="euler_angles", convention="ZXY")
                torch.cuda.synchronize()
                timer.enabled, timer.only = False, None
                ev = timer.events[dominant][before:]
                f32_ms, f32_n = sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev)
                del timer.events[dominant][before:]
        finally:
            drr.renderer.brick_storage = had

    t_max = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        # per-rank picture for the scaling run: timed region, dominant kernel, waits inside fence()
        k_tot, k_cnt = timer.total_ms(dominant)
        mine = torch.tensor([dt, k_tot / max(1, k_cnt), fence_wait["all_gather_s"], fence_wait["barrier_s"]],
                            device=device, dtype=torch.float64)
        every = torch.empty(world * 4, device=device, dtype=torch.float64)
        dist.all_gather_into_tensor(every, mine)
        every = every.reshape(world, 4).cpu()
        extra["ranks"] = {
            "timed_region_s": {"min": every[:, 0].min().item(), "max": every[:, 0].max().item()},
            "kernel_ms": {"min": every[:, 1].min().item(), "max": every[:, 1].max().item()},
            "final_fence_wait_own_work_and_all_gather_s": {"min": every[:, 2].min().item(),
                                                           "max": every[:, 2].max().item()},
            "final_fence_wait_barrier_s": {"min": every[:, 3].min().item(), "max": every[:, 3].max().item()},
        }
        if rank == 0:
            log(f"[bench] ranks: {extra['ranks']}")
    dt = t_max.item()
    if rank != 0:
        return None

    ms_per_step = dt / steps * 1e3
    kernel_timing = ("HIP events around every launch of the dominant kernel inside the timed region (the "
                     f"step's other kernels: around every launch in {census_steps} further steps)")
    if cfg in ("headline", "2") and "fused (" in str(extra.get("step", "")):
        kernel_timing += ("; in the fused step the record and the brick counter are cleared by the launch in "
                          "front of this one (ddrr_pose_raygen_forward), not by this entry")
    if cfg == "4" and not timer.events:
        # the timed region replayed a HIP graph: no launches went through the timer.  Time the
        # same forward (+ record) launches eagerly, after the fact
        kernel_timing = ("HIP events around 50 eager launches of the same call after the timed "
                         "region (the region itself replays a HIP graph)")
        timer.enabled = True
        for _ in range(50):
            r_, x_ = reg._rotation.detach().clone().requires_grad_(), reg._translation.detach().clone()
            ncc(gt, drr(r_, x_, parameterization="euler_angles", convention="ZXY")).sum().backward()
        torch.cuda.synchronize()
        timer.enabled = False
    steps_k = steps if kernel_timing.startswith("HIP events around every") else 50
    steps_other = census_steps if kernel_timing.startswith("HIP events around every") else 50
    # the dominant kernel: every launch of it in the timed region, HIP events on its stream
    names = [n for n in timer.events if n == dominant] or \
        [max(timer.events, key=lambda n: timer.total_ms(n)[0])]
    k_name = names[0]
    k_total, k_n = timer.total_ms(k_name)
    k_ms = k_total / max(1, k_n)   # per launch
    per_step = k_n / steps_k
    bwd_ms = sum(timer.total_ms(n)[0] for n in timer.events if "backward" in n and n != k_name) / max(1, steps_other)
    # algorithmic bytes of ONE launch
    with torch.no_grad():
        if cfg == "3":
            from diffdrr_amd.renderers import get_alpha_minmax

            s_v, t_v, L_v = voxel_rays(drr, rot0, xyz0)
            lo, hi = get_alpha_minmax(s_v, t_v, torch.tensor(drr.density.shape, device=device), 0.5, 1e-8)
            a0, a1 = lo.min(), hi.max()
            n_in = 0
            d = t_v - s_v + 1e-8
            for m0 in range(0, P, 32):  # samples whose 8-cell touches the volume
                al = a0 + (torch.arange(m0, min(P, m0 + 32), device=device) / (P - 1)) * (a1 - a0)
                x = s_v[:, :, None, :] + al[None, None, :, None] * d[:, :, None, :]
                inside = ((x > -1) & (x < D)).all(-1)
                n_in += int(inside.sum().item())
            launch_units = B
            alg_bytes = 32 * n_in + B * H * H * 20
            per_unit = f"{n_in / (B * H * H):.1f} samples in the volume per ray x 32 B + 20 B per ray"
            traffic_kind = "trilinear_forward"
        else:
            nposes = {"headline": B, "2": B, "4": 1, "5": min(512, -(-B // world))}[cfg]
            if cfg == "4":
                s_v, t_v, L_v = voxel_rays(drr, r0, x0)
            elif cfg == "5":
                s_v, t_v, L_v = voxel_rays(drr, rot0[:nposes], xyz0[:nposes])
            else:
                s_v, t_v, L_v = voxel_rays(drr, rot0, xyz0)
            nv_total = 0
            for a in range(0, nposes, 64):
                _, _, nvox = ops.siddon_forward(drr.density.detach(), s_v[a:a + 64], t_v[a:a + 64],
                                                L_v[a:a + 64], count_voxels=True, det=(H, H))
                nv_total += int(nvox.sum().item())
            launch_units = nposes
            alg_bytes = 4 * nv_total + nposes * H * H * 20 + 12 * nposes
            per_unit = (f"{nv_total / (nposes * H * H):.1f} voxels per ray x 4 B + 20 B per ray "
                        f"(target 12 + img 4 + out 4) + 12 B per source")
            traffic_kind = "forward_record" if cfg in ("headline", "2", "4") else "forward"
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic = traffic_record(traffic_kind) if (D, H) == (512, 256) and cfg in ("headline", "5") else None
    at_headline_size = (D, H) == (512, 256)
    # every renderer kernel of the step with its own algorithmic bytes
    kernels = []
    for name in timer.events:
        tot, cnt = timer.total_ms(name)
        if not cnt:
            continue
        ent = {"kernel": name, "kernel_ms": tot / cnt,
               "launches_per_step": cnt / (steps_k if name == k_name else max(1, steps_other))}
        if name == k_name:
            ent.update(algorithmic_bytes_per_launch=alg_bytes, frac=achieved / HBM_PEAK_GBS)
            if cfg == "headline" and at_headline_size and B == 32:  # (the workload the counters were taken on)
                ib = issue_bound_record("forward_record", tot / cnt, nv_total)
                if ib:
                    ent["issue_bound"] = ib
        elif cfg == "3" and name == "ddrr_trilinear_backward_volume_bricks":
            vb = (32 + 64) * n_in  # 8 corner reads + 8 corner read-modify-writes per sample
            ent.update(algorithmic_bytes_per_launch=vb, bound="LDS atomics (VALU issue)",
                       algorithmic_bytes_per_unit="(32 + 64) B per sample in the volume",
                       work_rate=vb / (tot / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       work_rate_note="algorithmic bytes over time in units of 8 TB/s: NOT a distance "
                                      "to a wall -- the 8 corner updates go to the brick in LDS, not "
                                      "to HBM, so this number can exceed 1")
            ib = issue_bound_record("trilinear_volume_gradient", tot / cnt)
            if ib:
                ent["issue_bound"] = ib
        kernels.append(ent)
    kernels.sort(key=lambda e: -e["kernel_ms"] * e["launches_per_step"])
    forward = None
    if f_ms is not None:
        forward = {"kernel": dominant + " (aux = NULL: forward only)", "kernel_ms": f_ms,
                   "launches_timed": f_n, "poses_per_launch": f_poses,
                   "primed_with": "60 launches, directly after the timed region" if cfg != "5"
                   else "4 launches of 512 poses, directly after the timed region",
                   "algorithmic_bytes_per_launch": alg_bytes,
                   "achieved": alg_bytes / (f_ms * 1e-3) / 1e9,
                   "frac": alg_bytes / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "target_frac": 0.70}
        tr = traffic_record("forward") if at_headline_size and cfg == "headline" and f_poses == 32 else None
        if tr:
            forward["traffic"], forward["traffic_source"] = tr
        if at_headline_size and cfg in ("headline", "5") and f_poses in (32, 512):
            ib = issue_bound_record("forward" if f_poses == 32 else "forward_sweep", f_ms, nv_total)
            if ib:
                forward["issue_bound"] = ib
        log(f"[bench] config {cfg} forward only: {f_ms:.3f} ms per launch of {f_poses} poses = "
            f"{forward['frac'] * 100:.1f} % of the 8 TB/s roofline")
    forward_f32 = None
    if f32_ms is not None:
        forward_f32 = {"kernel": dominant + " (aux = NULL: forward only), brick_storage = \"f32\"",
                       "what": "the same 32-pose launches on 32^3 bricks of the volume's own fp32 values "
                               "(what the reference gathers, renderers.py:159-164): no quantisation anywhere",
                       "kernel_ms": f32_ms, "launches_timed": f32_n, "poses_per_launch": int(rot0.shape[0]),
                       "primed_with": "40 launches, directly after the 16-bit forward leg",
                       "algorithmic_bytes_per_launch": alg_bytes,
                       "achieved": alg_bytes / (f32_ms * 1e-3) / 1e9,
                       "frac": alg_bytes / (f32_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "target_frac": 0.70}
        tr = traffic_record("forward_f32") if at_headline_size else None
        if tr:
            forward_f32["traffic"], forward_f32["traffic_source"] = tr
        log(f"[bench] config {cfg} forward only, fp32 bricks: {f32_ms:.3f} ms = "
            f"{forward_f32['frac'] * 100:.1f} % of the 8 TB/s roofline")
    log(f"[bench] config {cfg}: step {ms_per_step:.3f} ms | {k_name} {k_ms:.3f} ms per launch, "
        f"{per_step:.1f} launch(es) per step | backward kernels {bwd_ms:.3f} ms per step | "
        f"{per_unit} | {alg_bytes / launch_units / 1e6:.1f} MB algorithmic per DRR")
    result = {
        "metric": metric,
        "value": units_per_step * steps / dt,
        "unit": unit,
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "baseline_config": cfg,
            "volume": f"{D}x{D}x{D} f32 ({D ** 3 * 4 / 2 ** 20:.0f} MiB, replicated per GPU)",
            "detector": f"{H}x{H}",
            "batch_per_gpu": B if scaling == "weak" else -(-B // world),
            "global_batch": B * world if scaling == "weak" else B,
            "parallelism": (f"pose-sharded x{world}, all_gather of per-pose values (RCCL)"
                            if world > 1 else "single GPU"),
        },
        "roofline": {
            "kernel": k_name,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_is": "the contract's A / P: algorithmic (SURVEY 8d) bytes per launch / kernel time / "
                       "8 TB/s.  `bound` is the wall the contract prices the kernel against; what BINDS "
                       "it is `limiter` (the bricks are read from LDS: HBM carries `hbm_traffic_frac` "
                       "of its peak) and `issue_bound` is the distance to that wall",
            "limiter": "valu_issue",
            "hbm_traffic_frac": (traffic[0] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and k_ms else None,
            "traffic": traffic[0] if traffic else None,
            "traffic_source": (traffic[1] + " (separate rocprofv3 --pmc passes of this command, "
                               "committed; not measured in this run)") if traffic else None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes_per_unit": per_unit,
            "units_per_launch": launch_units,
            "kernel_ms": k_ms,
            "kernel_timing": kernel_timing,
            "launches_per_step": per_step,
            "launches_timed": k_n,
            "forward": forward,
            "forward_f32": forward_f32,
            "kernels": kernels,
        },
    }
    storage = getattr(drr.renderer, "brick_storage", None)
    if storage is not None:
        from diffdrr_amd.renderers import _brick_storage

        used = _brick_storage(drr.density, {"storage": storage})
        q16 = ("bricks staged as 16-bit block-quantised voxels, one (min, step) per 32x32x64 "
               "brick, where the brick's range is <= 12x its level (else the brick is rendered from "
               "its fp32 values: brick_storage_fallbacks), |error| <= brick range / 131070 per "
               "voxel, fp32 arithmetic (parity block: measured in this run)")
        result["config"]["brick_storage"] = {
            "q16": "q16: " + q16,
            "q16p": "q16p: " + q16 + "; staged from the volume's packed 16-bit bricks, a "
                    "per-volume layout copy (+52 % of the volume's bytes) built by the first "
                    "render after the volume changed (here: in the warm-up, +0.35 ms once), "
                    "like the bricks' (min, max) table",
        }.get(used, "f32: the volume's own values" + ("" if used == storage else
              f" (the module's {storage} applies to volumes with >= 4 double bricks per CU)"))
        fb = ops.brick_fallbacks(drr.density, used) if used != "f32" else None
        if fb is not None:
            result["config"]["brick_storage_fallbacks"] = fb[0]
            result["config"]["bricks"] = fb[1]
    if world == 1 and cfg in ("headline", "2") and not (top and args.no_cpu_baseline):
        if top:
            result["cpu_baseline"], result["parity"] = cpu_baseline_and_parity(
                drr, rot0, xyz0, images, rot.grad, xyz.grad, base, H)
        else:
            result["parity"] = pose_parity(drr, rot0, xyz0, images, rot.grad, xyz.grad, base, 1)
        result["parity"]["oracle"] = (
            "oracle/drr_oracle.c (C restatement of diffdrr/renderers.py:34-183, pinned to the "
            "reference's fixtures): fp32 = the reference's arithmetic, fp64 = exact; pose "
            "gradients: the oracle's analytic ray gradients chained through float64 ray "
            "generation to (rot, xyz[mm] x 100) of the timed step's own backward")
        result["parity"]["tolerance"] = (
            "fwd_rel_err <= 1e-4 (at the pixels where the fp32 reference is itself within 1e-4 of "
            "fp64; ref_off_pixels counts the others; fwd_rel_err_all_pixels: no mask), "
            "fwd_rel_err_vs_fp64 <= 1e-4 everywhere; "
            "pose_grad_rel_err_vs_fp64 <= 2 x the reference's own fp32 error + 1e-3")
    if world == 1 and cfg == "headline" and top and not args.no_cpu_baseline:
        # the same check where gradients are not tie-breaking noise: a phantom volume, absolute bound
        ph = set_storage(DRR(synthetic_subject(D, kind="phantom", seed=0), sdd=1020.0, height=H,
                             delx=delx, renderer="siddon").to(device))
        n_ph = 2
        r_ph = rot0[:n_ph].clone().requires_grad_()
        x_ph = xyz0[:n_ph].clone().requires_grad_()
        with torch.no_grad():
            base_ph = ph(torch.zeros(1, 3, device=device), torch.tensor([[0.0, 850.0, 0.0]], device=device),
                         parameterization="euler_angles", convention="ZXY")
        img_ph = ph(r_ph, x_ph, parameterization="euler_angles", convention="ZXY")
        ncc(base_ph.expand(n_ph, -1, -1, -1), img_ph).sum().backward()
        pp = pose_parity(ph, rot0[:n_ph], xyz0[:n_ph], img_ph.detach(), r_ph.grad, x_ph.grad, base_ph, n_ph)
        pp["volume"] = f"{D}^3 phantom (smooth ellipsoids), same detector and poses"
        pp["tolerance"] = "pose_grad_rel_err_vs_fp64 <= 1e-3 (absolute bound), fwd_rel_err <= 1e-4"
        used_ph = _brick_storage(ph.density, {"storage": getattr(ph.renderer, "brick_storage", "f32")})
        fb = ops.brick_fallbacks(ph.density, used_ph) if used_ph != "f32" else None
        if fb is not None:
            pp["brick_storage_fallbacks"], pp["bricks"] = fb
        result["parity"]["phantom"] = pp
        del ph
    if world == 1 and cfg == "3" and on_gpu and parity and not (top and args.no_cpu_baseline):
        result["parity"] = trilinear_parity(drr, rot0, xyz0, keep["img"].detach(), go, P, H)
    if world == 1 and cfg == "5" and on_gpu and not (top and args.no_cpu_baseline):
        with torch.no_grad():
            imgs = drr(rot0[:nposes], xyz0[:nposes], parameterization="euler_angles", convention="ZXY")
        result["parity"] = sweep_parity(drr, fixed, rot0, xyz0, imgs, keep["vals"],
                                        (0, nposes // 2, nposes - 1))
        result["parity"]["tolerance"] = "fwd_rel_err <= 1e-4; ncc_abs_err <= 1e-4"
        del imgs
    if cfg in ("headline", "2", "5") and D in REFERENCE_CPU and top:
        fwd_only = cfg == "5"
        result["reference_cpu"] = {
            "value": REFERENCE_CPU[D]["forward_only" if fwd_only else "value"], "unit": "DRRs/s",
            "cores": 8, "kind": "reference",
            "what": ("the UNMODIFIED reference (diffdrr.drr.DRR, CPU torch, 8 threads) on the same "
                     f"{D}^3 -> 256^2 scene, " + ("forward" if fwd_only else "forward + backward to the pose")
                     + ", one pose per call; measured in the build container: /root/reference does "
                     "not exist on the GPU box"),
            "source": "profiles/r02/ref_cpu_baseline.txt (tools/ref_cpu_baseline.py)"}
    result.update(extra)
    return result


def trilinear_parity(drr, rot, xyz, image, go, P, H, rows=4):
    """Config 3 against the oracle on a ray subset: `rows` detector rows of the timed step's image,
    and the volume gradient of exactly those rays (one more launch of the volume-gradient kernel
    with grad_out zero elsewhere) against the oracle's, which accumulates in double."""
    import numpy as np

    import oracle
    from diffdrr_amd import ops
    from diffdrr_amd.renderers import get_alpha_minmax

    dev = image.device
    s, t, L = voxel_rays(drr, rot[:1], xyz[:1])
    V = drr.density.detach()
    lo, hi = get_alpha_minmax(s, t, torch.tensor(V.shape, device=dev), 0.5, 1e-8)
    amin, amax = lo.min().reshape(1).contiguous(), hi.max().reshape(1).contiguous()
    picks = [int(r) for r in np.linspace(H // 8, H - 1 - H // 8, rows)]
    idx = torch.cat([torch.arange(r * H, (r + 1) * H) for r in picks]).to(dev)
    g_sub = torch.zeros(1, H * H, device=dev)
    g_sub[0, idx] = go.reshape(-1, H * H)[0, idx]
    gv = ops.trilinear_backward_volume_bricks(V.shape, s, t, L, g_sub, amin, amax, (H, H), n_points=P)
    kw = dict(n_points=P, alphamin=float(amin.item()), alphamax=float(amax.item()))
    a32 = (V.cpu().numpy(), s[:, :1].cpu().numpy(), t[:, idx].cpu().numpy(), L[:, idx].cpu().numpy())
    ref = oracle.trilinear(*a32, grad_out=g_sub[:, idx].cpu().numpy(), want_volume_grad=True, **kw)
    mine = image.reshape(-1, H * H)[0, idx].cpu().numpy()
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-300))  # noqa
    return {"rays": int(idx.numel()), "detector_rows": picks,
            "fwd_rel_err": rel(mine, ref["out"].reshape(-1)),
            "volume_grad_rel_err": rel(gv.cpu().numpy(), ref["g_volume"]),
            "oracle": "oracle/drr_oracle.c trilinear (restatement of diffdrr/renderers.py:205-254), fp32 "
                      "arithmetic, volume gradient accumulated in double",
            "tolerance": "fwd_rel_err <= 1e-4, volume_grad_rel_err <= 1e-3"}


def ct_config(rt, poses=(1, 8, 32), det=200):
    """`configs.ct`: the reference's example geometry (README.md:67-87: 512 x 512 x 133 CT at
    0.703 x 0.703 x 2.5 mm, sdd 1020, 200 x 200 detector at delx 2.0, source 850 mm away) on a
    CT-LIKE volume -- the CT itself is not shipped -- that went through the package's own
    `transform_hu_to_density` (reference data.py:214-227): exact-zero air, dim lung texture among
    zeros, a partial-volume skin, bone near 0.5, a metal marker at 1.0.  Kernel-only timings of the
    forward and forward + record launches at 1 / 8 / 32 perturbed poses on the default storage
    (guarded 16-bit bricks, with how many bricks the guard sent to the fp32 path) and on fp32
    bricks, the algorithmic rate of each, and parity of a rendered pose against the oracle."""
    import numpy as np

    import oracle
    from diffdrr_amd import DRR, ops
    from diffdrr_amd.data import ct_like_hu_volume, make_subject, transform_hu_to_density

    device, timer = rt.device, rt.timer
    dims, spacing = (512, 512, 133), (0.703, 0.703, 2.5)
    density = transform_hu_to_density(ct_like_hu_volume(dims, seed=0))
    drr = DRR(make_subject(density, spacing=spacing, orientation="AP"), sdd=1020.0, height=det, delx=2.0,
              renderer="siddon").to(device)
    V = drr.density
    name = "ddrr_siddon_forward_bricks"
    out = {"workload": "512x512x133 CT-like volume (HU phantom -> transform_hu_to_density) at 0.703 x 0.703 x "
                       f"2.5 mm -> {det}x{det} detector (sdd 1020, delx 2.0), Siddon, perturbed poses: the "
                       "reference's example geometry (README.md:67-87)",
           "volume": {"shape": list(dims), "zero_fraction": float((density == 0).float().mean()),
                      "max": float(density.max()), "soft_tissue": float(density[256, 256, 66])},
           "kernel": name, "poses": {},
           "work_rate_is": "algorithmic bytes (4 B per voxel a ray crosses, air included, + 20 B per ray) per "
                           "launch / kernel time / 8 TB/s: NOT a roofline fraction -- two thirds of this "
                           "volume is exact-zero air, whose bricks are skipped, so the figure can exceed 1"}
    from diffdrr_amd.renderers import _brick_storage
    out["module_default_storage"] = {
        "storage": _brick_storage(V, {"storage": drr.renderer.brick_storage}),
        "why": "Siddon.brick_storage = \"q16p\" applies to volumes with >= 4 double bricks per CU; this shape has "
               "768 (3 per CU): the module decides by the poses of the launch from a measured table "
               "(renderers._FEW_BRICKS_POLICY, profiles/r06/storage_table.txt) -- 16-bit bricks at 8 ... 12 "
               "poses, fp32 bricks (`forward_f32` / `forward_record_f32`) otherwise",
        "by_poses": {str(B): _brick_storage(V, {"storage": drr.renderer.brick_storage}, B) for B in poses}}

    def timed(fn, n_prime, n_timed):
        for _ in range(n_prime):
            fn()
        before = len(timer.events.get(name, []))
        timer.enabled, timer.only = True, name
        for _ in range(n_timed):
            fn()
        torch.cuda.synchronize()
        timer.enabled, timer.only = False, None
        ev = timer.events[name][before:]
        ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        del timer.events[name][before:]
        return ms

    with torch.no_grad():
        for B in poses:
            rot, xyz = perturbed_poses(B, seed=2, device=device)
            s, t, L = voxel_rays(drr, rot, xyz)
            _, _, nvox = ops.siddon_forward(V, s, t, L, count_voxels=True, det=(det, det))
            visits = int(nvox.sum().item())
            alg = 4 * visits + B * det * det * 20 + 12 * B
            ent = {"voxels_per_ray": visits / (B * det * det), "algorithmic_bytes_per_launch": alg}
            n_prime, n_timed = (60, 40) if B <= 8 else (30, 20)
            for key, storage, aux in (("forward", "q16p", False), ("forward_record", "q16p", True),
                                      ("forward_f32", "f32", False), ("forward_record_f32", "f32", True)):
                ms = timed(lambda: ops.siddon_forward_bricks(V, s, t, L, (det, det), want_aux=aux,
                                                             storage=storage), n_prime, n_timed)
                ent[key] = {"kernel_ms": ms, "work_rate": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "drrs_per_s": B / (ms * 1e-3)}
            out["poses"][str(B)] = ent
            log(f"[bench] config ct, {B} pose(s): forward {ent['forward']['kernel_ms']:.3f} ms (fp32 bricks "
                f"{ent['forward_f32']['kernel_ms']:.3f}), + record {ent['forward_record']['kernel_ms']:.3f} ms "
                f"(fp32 bricks {ent['forward_record_f32']['kernel_ms']:.3f})")
        fb = ops.brick_fallbacks(V, "q16p")
        if fb is not None:
            out["brick_storage_fallbacks"], out["bricks"] = fb
        # parity: pose 0 of the 8-pose batch on both storages against the oracle's fp32 / fp64 renders
        rot, xyz = perturbed_poses(8, seed=2, device=device)
        s, t, L = voxel_rays(drr, rot[:1], xyz[:1])
        a32 = (V.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy(), L.cpu().numpy())
        r32 = oracle.siddon(*a32)["out"].reshape(-1)
        r64 = oracle.siddon(*(a.astype(np.float64) for a in a32))["out"].reshape(-1)
        rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())  # noqa: E731
        par = {"pose": 0, "ref_fp32_fwd_rel_err_vs_fp64": rel(r32, r64),
               "tolerance": "fwd_rel_err <= 1e-4 (image-normalised), every storage"}
        for storage in ("q16p", "f32"):
            mine = ops.siddon_forward_bricks(V, s, t, L, (det, det), storage=storage)[0].reshape(-1).cpu().numpy()
            par[storage] = {"fwd_rel_err": rel(mine, r32), "fwd_rel_err_vs_fp64": rel(mine, r64),
                            # per pixel, relative to the pixel's own value (where it is not in air)
                            "max_pixel_rel_err_vs_fp64": float(
                                (np.abs(mine - r64) / np.maximum(np.abs(r64), 1e-3 * np.abs(r64).max())).max())}
        out["parity"] = par
    return out


def sparse_config(rt, poses=(1, 8), det=200, p=0.1):
    """`configs.sparse`: the reference's own speed lever on its own example geometry -- the CT-like
    512 x 512 x 133 volume of `configs.ct` -> 200 x 200 with `p_subsample = 0.1` (reference
    drr.py:36-39, 142-147; published: 5.15 ms per forward call on an RTX 2080 Ti,
    notebooks/tutorials/introduction.ipynb:611) at 1 and 8 poses: the MODULE call, forward (no
    grad) and forward + backward to the Euler pose, wall time per call between HIP events --
    on the volume-stationary kernels (the subsample's grid is rendered by the fused entries and
    gathered: `DRR._render_sparse`), next to the same module on the per-ray kernels (where every
    subsample went until round 5) and to the dense 200 x 200 render; parity of the subsample
    against the oracle on exactly the listed rays."""
    import numpy as np

    import oracle
    from diffdrr_amd import DRR
    from diffdrr_amd.data import ct_like_hu_volume, make_subject, transform_hu_to_density

    device = rt.device
    density = transform_hu_to_density(ct_like_hu_volume((512, 512, 133), seed=0))
    subject = make_subject(density, spacing=(0.703, 0.703, 2.5), orientation="AP")
    torch.manual_seed(0)
    sub = DRR(subject, sdd=1020.0, height=det, delx=2.0, renderer="siddon", p_subsample=p).to(device)
    dense = DRR(subject, sdd=1020.0, height=det, delx=2.0, renderer="siddon").to(device)
    out = {"workload": f"512x512x133 CT-like volume -> {det}x{det}, Siddon, p_subsample = {p} "
                       f"({sub.detector.n_subsample} rays per pose), module calls",
           "p_subsample": p, "unit": "ms per module call (HIP events around 50 calls)",
           "reference_published_ms": 5.15,
           "reference_published": "one pose, forward, RTX 2080 Ti (notebooks/tutorials/introduction.ipynb:611); "
                                  "dense 200 x 200: 25.2 ms (README.md:85-87)",
           "poses": {}}

    def timed(fn, warm=20, n=50):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def call(drr, rot0, xyz0, grad):
        if not grad:
            def fn():
                with torch.no_grad():
                    drr(rot0, xyz0, parameterization="euler_angles", convention="ZXY")
            return fn
        rot, xyz = rot0.clone().requires_grad_(), xyz0.clone().requires_grad_()

        def fn():
            rot.grad = xyz.grad = None
            drr(rot, xyz, parameterization="euler_angles", convention="ZXY").sum().backward()
        return fn

    for B in poses:
        rot0, xyz0 = perturbed_poses(B, seed=2, device=device)
        ent = {}
        for key, drr, fused in (("bricks", sub, True), ("per_ray", sub, False), ("dense", dense, True)):
            drr.fuse_ray_generation = fused
            for grad in (False, True):
                ent[f"{key}_{'forward_backward' if grad else 'forward'}"] = {"ms": timed(call(drr, rot0, xyz0, grad))}
            drr.fuse_ray_generation = True
        out["poses"][str(B)] = ent
        log(f"[bench] config sparse, {B} pose(s), p_subsample {p}: forward {ent['bricks_forward']['ms']:.3f} ms "
            f"(per-ray kernels {ent['per_ray_forward']['ms']:.3f}, dense {ent['dense_forward']['ms']:.3f}), "
            f"forward + backward {ent['bricks_forward_backward']['ms']:.3f} ms "
            f"(per-ray {ent['per_ray_forward_backward']['ms']:.3f}, dense {ent['dense_forward_backward']['ms']:.3f})")
    # parity: the subsample of pose 0 against the oracle on exactly those rays
    rot0, xyz0 = perturbed_poses(1, seed=2, device=device)
    sub.reshape = False
    with torch.no_grad():
        mine = sub(rot0, xyz0, parameterization="euler_angles", convention="ZXY").reshape(-1).cpu().numpy()
        from diffdrr_amd.pose import convert
        src, tgt = sub.detector(convert(rot0, xyz0, parameterization="euler_angles", convention="ZXY"), None)
        L = (tgt - src).norm(dim=-1)
        s, t = sub.affine_inverse(src), sub.affine_inverse(tgt)
    a32 = (sub.density.cpu().numpy(), s.cpu().numpy(), t.cpu().numpy(), L.cpu().numpy())
    r32 = oracle.siddon(*a32)["out"].reshape(-1)
    r64 = oracle.siddon(*(a.astype(np.float64) for a in a32))["out"].reshape(-1)
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())  # noqa: E731
    out["parity"] = {"rays": int(mine.size), "fwd_rel_err": rel(mine, r32), "fwd_rel_err_vs_fp64": rel(mine, r64),
                     "ref_fp32_fwd_rel_err_vs_fp64": rel(r32, r64),
                     "tolerance": "fwd_rel_err <= 1e-4 (image-normalised)"}
    return out


def few_poses_config(rt, poses=(1, 2, 8), det=256, D=512):
    """`configs.few_poses`: kernel-only timings of ddrr_siddon_forward_bricks at the headline size with
    1 / 2 / 8 poses per launch -- what a registration step or a small sweep launches -- on the bench's
    noise volume (every brick quantised) and on the phantom (a third of its bricks on the fp32 path),
    default storage; forward and forward + record.  A launch of one pose is ~71 % per-brick fixed
    cost (profiles/r05/one_pose_stage_stamps.txt)."""
    from diffdrr_amd import DRR, ops
    from diffdrr_amd.data import make_subject, noise_volume, phantom_volume

    device, timer = rt.device, rt.timer
    name = "ddrr_siddon_forward_bricks"
    out = {"kernel": name, "detector": f"{det}x{det}", "storage": "q16p",
           "frac_is": "algorithmic bytes per launch / kernel time / 8 TB/s: a work rate (see roofline.frac_is)"}

    def timed(fn, n_prime, n_timed):
        for _ in range(n_prime):
            fn()
        before = len(timer.events.get(name, []))
        timer.enabled, timer.only = True, name
        for _ in range(n_timed):
            fn()
        torch.cuda.synchronize()
        timer.enabled, timer.only = False, None
        ev = timer.events[name][before:]
        ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        del timer.events[name][before:]
        return ms

    with torch.no_grad():
        for kind, vol in (("noise", noise_volume(D, seed=0)), ("phantom", phantom_volume(D, seed=0))):
            drr = DRR(make_subject(vol, spacing=(1.0, 1.0, 1.0), orientation="AP"), sdd=1020.0, height=det,
                      delx=2.4 * (256 / det) * (D / 512), renderer="siddon").to(device)
            V = drr.density
            ent = {}
            for B in poses:
                rot, xyz = perturbed_poses(B, seed=2, device=device)
                s, t, L = voxel_rays(drr, rot, xyz)
                _, _, nvox = ops.siddon_forward(V, s, t, L, count_voxels=True, det=(det, det))
                alg = 4 * int(nvox.sum().item()) + B * det * det * 20 + 12 * B
                e = {"algorithmic_bytes_per_launch": alg}
                for key, aux in (("forward", False), ("forward_record", True)):
                    ms = timed(lambda: ops.siddon_forward_bricks(V, s, t, L, (det, det), want_aux=aux,
                                                                 storage="q16p"), 60, 40)
                    e[key] = {"kernel_ms": ms, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "drrs_per_s": B / (ms * 1e-3)}
                    assert e[key]["frac"] < 1.0  # (every voxel of these volumes is read: a fraction)
                ent[str(B)] = e
            fb = ops.brick_fallbacks(V, "q16p")
            if fb is not None:
                ent["brick_storage_fallbacks"], ent["bricks"] = fb
            out[kind] = ent
            log(f"[bench] few poses, {kind}: " + ", ".join(
                f"{B}: {ent[str(B)]['forward']['kernel_ms']:.3f} / {ent[str(B)]['forward_record']['kernel_ms']:.3f} ms"
                for B in poses) + " (forward / forward + record)")
            del drr, V
    return out


def guarded_run(world, name, fn):
    """One GPU: a side run that fails must not cost the driver its headline line (the failure is
    recorded in its place and on stderr).  Several ranks: exceptions propagate -- a rank that
    skipped a collective would hang the others."""
    t0 = time.perf_counter()
    if world > 1:
        out = fn()
    else:
        try:
            out = fn()
        except Exception as exc:  # noqa: BLE001
            import traceback

            log(f"[bench] configs.{name} FAILED: {type(exc).__name__}: {exc}\n{traceback.format_exc()}")
            out = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    if isinstance(out, dict):
        out["wall_s"] = time.perf_counter() - t0
    return out


def summary_of(res):
    """What the default line keeps of a short config run."""
    rf = res["roofline"]
    out = {"metric": res["metric"], "value": res["value"], "unit": res["unit"], "steps": res["steps"],
           "ms_per_step": res["ms_per_step"], "scaling": res["scaling"],
           "workload": res["config"]["workload"],
           "dominant_kernel": {"kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"], "frac": rf["frac"],
                               "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"]},
           "kernels": rf["kernels"], "parity": res.get("parity")}
    if rf.get("forward"):
        out["forward"] = rf["forward"]
    for k in ("brick_storage", "brick_storage_fallbacks", "bricks"):
        if k in res["config"]:
            out[k] = res["config"][k]
    if "registration" in res:
        out["registration"] = res["registration"]
    return out


LINE_LIMIT = 6144  # bytes of the final stdout line (the driver parses the tail of stdout; the
#                    26.6 KB line of round 5 came back `parsed: null`)


def _sig(x, n=5):
    """Floats of the printed line: n significant digits (the full record keeps every digit)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x == 0.0 or not math.isfinite(x):
        return x if math.isfinite(x) else None
    return float(f"{x:.{n}g}")


def _sig_tree(o):
    if isinstance(o, dict):
        return {k: _sig_tree(v) for k, v in o.items() if v is not None or k in ("vs_baseline", "traffic")}
    if isinstance(o, (list, tuple)):
        return [_sig_tree(v) for v in o]
    return _sig(o)


def _pick(d, *keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


def _cap(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def compact_line(full, limit=LINE_LIMIT):
    """The ONE line rank 0 prints on stdout, from the full record (`bench_full.json`, also on
    stderr): the driver's contract keys, `roofline`, `cpu_baseline`, `reference_cpu`, the headline
    parity numbers and a few numbers per config -- at most `limit` bytes whatever the run (sections
    are dropped, least important first, until it fits; tests/test_bench_line.py).  Names: `frac` is
    the contract's A / P (algorithmic bytes over kernel time over the HBM peak) and appears only
    where it cannot exceed 1; the same quotient for kernels that skip work the count includes
    (all-zero bricks of a CT, bytes served by LDS) is called `work_rate`."""
    rf = full["roofline"]
    kern0 = (rf.get("kernels") or [{}])[0]
    ib = kern0.get("issue_bound") or {}
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                 "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in full}
    cfg = dict(_pick(full["config"], "baseline_config", "volume", "detector", "batch_per_gpu", "global_batch",
                     "parallelism", "brick_storage_fallbacks", "bricks"))
    cfg["workload"] = _cap(full["config"]["workload"], 200)
    if "brick_storage" in full["config"]:
        cfg["brick_storage"] = full["config"]["brick_storage"].split(":")[0]
    if "step" in full:
        cfg["step"] = _cap(full["step"], 60)
    line["config"] = {"workload": cfg.pop("workload"), **cfg}
    r = _pick(rf, "kernel", "bound", "achieved", "peak", "unit", "frac")
    r["traffic"] = rf.get("traffic")
    r.update(_pick(rf, "kernel_ms", "algorithmic_bytes_per_launch", "units_per_launch", "launches_timed"))
    r["algorithmic_bytes_per_unit"] = _cap(rf.get("algorithmic_bytes_per_unit"), 100)
    if rf.get("traffic") and rf.get("kernel_ms"):
        r["hbm_traffic_frac"] = rf["traffic"] / (rf["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    r.update(_pick(rf, "limiter"))
    if ib:
        # what binds the kernel is not the `bound` the contract prices it against
        r["issue_bound"] = _pick(ib, "issue_frac", "useful_frac")
    r["step_minus_kernel_ms"] = full["ms_per_step"] - rf["kernel_ms"] * rf.get("launches_per_step", 1.0)
    for key in ("forward", "forward_f32", "forward_sweep"):
        f = rf.get(key)
        if f:
            e = _pick(f, "kernel_ms", "frac", "poses_per_launch", "traffic")
            if f.get("issue_bound"):
                e["issue_bound"] = _pick(f["issue_bound"], "issue_frac", "useful_frac")
            if f.get("parity"):
                e.update(_pick(f["parity"], "fwd_rel_err", "fwd_rel_err_vs_fp64"))
            r[key] = e
    if rf.get("forward"):
        r["forward"]["target_frac"] = 0.70
    r["other_kernels_ms_per_step"] = {
        k["kernel"].replace("ddrr_", ""): k["kernel_ms"] * k["launches_per_step"]
        for k in (rf.get("kernels") or [])[1:6]}
    line["roofline"] = r
    if "cpu_baseline" in full:
        cb = dict(full["cpu_baseline"])
        cb["sample"] = _cap(cb.get("sample"), 160)
        line["cpu_baseline"] = cb
    if "reference_cpu" in full:
        line["reference_cpu"] = _pick(full["reference_cpu"], "value", "unit", "cores", "kind", "source")
    pkeys = ("fwd_rel_err", "fwd_rel_err_vs_fp64", "ref_fp32_fwd_rel_err_vs_fp64", "pose_grad_rel_err_vs_fp64",
             "ref_fp32_pose_grad_rel_err_vs_fp64", "fwd_rel_err_all_pixels", "ref_off_pixels", "poses")
    if full.get("parity"):
        par = _pick(full["parity"], *pkeys)
        par["tolerance"] = "fwd 1e-4 (vs fp32 ref where it is within 1e-4 of fp64, vs fp64 everywhere); " \
                           "pose grad vs fp64 <= 2x the reference's own fp32 error + 1e-3"
        if full["parity"].get("phantom"):
            par["phantom"] = _pick(full["parity"]["phantom"], "fwd_rel_err", "fwd_rel_err_vs_fp64",
                                   "pose_grad_rel_err_vs_fp64", "ref_fp32_pose_grad_rel_err_vs_fp64",
                                   "brick_storage_fallbacks")
        line["parity"] = par
    if "sweep" in full:
        line["sweep"] = _pick(full["sweep"], "value", "unit", "n_gpus", "scaling", "ms_per_step", "steps",
                              "poses", "poses_per_launch")
    if "ranks" in full:
        line["ranks"] = full["ranks"]
    for k in ("rccl_ranks", "devices_visible", "registration"):
        if k in full:
            line[k] = full[k]
    configs = {}
    for name, c in (full.get("configs") or {}).items():
        if name == "sparse":
            e = {"unit": "ms per module call [forward, forward + backward]", "poses": {}}
            for B, ent in c.get("poses", {}).items():
                e["poses"][B] = {k: [_sig(ent[f"{k}_forward"]["ms"], 4), _sig(ent[f"{k}_forward_backward"]["ms"], 4)]
                                 for k in ("bricks", "per_ray", "dense") if f"{k}_forward" in ent}
            e.update(_pick(c, "p_subsample", "reference_published_ms"))
            if c.get("parity"):
                e["parity"] = _pick(c["parity"], "fwd_rel_err", "fwd_rel_err_vs_fp64")
        elif name == "ct":
            e = {"poses": {}}
            for B, ent in c.get("poses", {}).items():
                e["poses"][B] = {k: _sig(v["kernel_ms"], 4) if isinstance(v, dict) and "kernel_ms" in v else
                                 (_sig(v["ms"], 4) if isinstance(v, dict) and "ms" in v else v)
                                 for k, v in ent.items()
                                 if isinstance(v, dict) and ("kernel_ms" in v or "ms" in v)}
            e["unit"] = "ms per launch"
            e.update(_pick(c, "brick_storage_fallbacks", "bricks", "p_subsample", "reference_published_ms"))
            if c.get("parity"):
                e["parity"] = {k: (_pick(v, "fwd_rel_err", "fwd_rel_err_vs_fp64") if isinstance(v, dict) else v)
                               for k, v in c["parity"].items() if k not in ("tolerance", "oracle", "pose", "rays")}
        elif name == "few_poses":
            e = {"unit": "ms per launch [forward, forward + record]"}
            for kind in ("noise", "phantom"):
                if kind in c:
                    e[kind] = {B: [_sig(v["forward"]["kernel_ms"], 4), _sig(v["forward_record"]["kernel_ms"], 4)]
                               for B, v in c[kind].items() if isinstance(v, dict)}
        else:
            dk = c.get("dominant_kernel") or {}
            e = _pick(c, "value", "unit", "ms_per_step")
            e["kernel_ms"] = dk.get("kernel_ms")
            e["frac"] = dk.get("frac")
            if c.get("forward"):
                e["forward_frac"] = c["forward"].get("frac")
            e.update(_pick(c.get("parity") or {}, "fwd_rel_err", "fwd_rel_err_vs_fp64", "pose_grad_rel_err_vs_fp64",
                           "ref_fp32_pose_grad_rel_err_vs_fp64", "volume_grad_rel_err", "ncc_abs_err"))
            e.update(_pick(c, "brick_storage_fallbacks"))
            if "b4" in c:
                e["b4"] = _pick(c["b4"], "value", "ms_per_step")
            if "registration" in c:
                e["registration"] = _pick(c["registration"], "hip_graph", "ncc_after", "iterations",
                                          "rot_error_rad", "xyz_error_mm")
        if isinstance(c, dict) and "error" in c:
            e = {"error": _cap(c["error"], 120)}
        configs[name] = e
    if configs:
        line["configs"] = configs
    line["full_record"] = "bench_full.json (beside bench.py) and stderr"
    exact = {k: line[k] for k in ("value", "ms_per_step") if k in line}  # (the driver's consistency check)
    line = _sig_tree(line)
    line.update(exact)
    # by construction, not by luck: drop the least important sections until the line fits
    for drop in (("configs", "few_poses"), ("configs", "ct"), ("configs", "sparse"), ("roofline", "other_kernels_ms_per_step"),
                 ("configs",), ("reference_cpu",), ("ranks",), ("sweep",)):
        if len(json.dumps(line)) <= limit:
            break
        tgt = line
        for k in drop[:-1]:
            tgt = tgt.get(k, {})
        if tgt.pop(drop[-1], None) is not None:
            line.setdefault("dropped_for_size", []).append(".".join(drop))
    assert len(json.dumps(line)) <= limit, "bench line over its size limit"
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="headline", choices=["headline", "2", "3", "4", "5"])
    ap.add_argument("--batch", type=int, default=None, help="poses per GPU per step")
    ap.add_argument("--size", type=int, default=None, help="volume edge (voxels)")
    ap.add_argument("--det", type=int, default=None, help="detector edge (pixels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="headline only: skip the short runs of configs 2, 3, 5 (`configs`, `sweep`)")
    ap.add_argument("--sweep-poses", type=int, default=None,
                    help="candidate poses of the `sweep` sub-run (default 4096; given explicitly, the "
                         "sub-run also happens at non-default sizes / on the cpu harness)")
    ap.add_argument("--storage", default=None, choices=["q16p", "q16", "f32"],
                    help="Siddon.brick_storage (default: the module's default, q16p)")
    ap.add_argument("--unfused", action="store_true",
                    help="headline / config 2: the step as DRR.forward + the NCC module through autograd "
                         "(six small launches around the brick kernel; nine before the render became one autograd node) "
                         "instead of DRR.ncc (three)")
    ap.add_argument("--fused-max-poses", type=int, default=None,
                    help="DRR.FUSED_NCC_MAX_POSES for this run (measurement: where the fused step stops paying)")
    ap.add_argument("--torch-adam", action="store_true",
                    help="config 4: torch.optim.Adam(fused, capturable) instead of diffdrr_amd.PoseAdam")
    ap.add_argument("--criterion", default="ncc", choices=["ncc", "multiscale", "gradient", "gradient_patch9"],
                    help="config 4: the similarity of the registration loop -- NormalizedCrossCorrelation2d (the fused "
                         "step), MultiscaleNormalizedCrossCorrelation2d([13, None], [0.5, 0.5]) (metrics.ipynb:94), "
                         "GradientNormalizedCrossCorrelation2d() or GradientNormalizedCrossCorrelation2d(patch_size=9, "
                         "sigma=1) (a timing run: it does not converge on the phantom): composed from the renderer "
                         "and the metric kernels inside the same HIP graph")
    ap.add_argument("--packed-record", action="store_true",
                    help="Siddon.packed_record = True (the opt-in fixed-point backward record)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: harness test only (gloo ranks; the kernels are whatever "
                         "DDRR_BENCH_HOOK routes diffdrr_amd.ops to, see tests/test_dist_gloo.py)")
    args = ap.parse_args()

    if args.device == "cuda":
        # pre-flight, before anything is spawned: more ranks than devices would die one by one in
        # torch.cuda.set_device under a torchrun traceback -- one JSON line and rc 2 instead
        visible = torch.cuda.device_count()
        want = max(args.gpus, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        if want > visible or visible == 0:
            if int(os.environ.get("RANK", "0")) == 0:
                print(json.dumps({"error": f"--gpus {args.gpus}: {want} rank(s) on this node need {want} "
                                           f"GPU(s), torch.cuda.device_count() = {visible}",
                                  "n_gpus_requested": args.gpus, "n_gpus_visible": visible}), flush=True)
            sys.exit(2)
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: start one rank per GPU ourselves (RCCL needs one process per device)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        os.execvpe(cmd[0], cmd, env)

    rt = Runtime(args)
    if rt.world != args.gpus and rt.rank == 0:
        log(f"note: --gpus {args.gpus}, process group has {rt.world} ranks; reporting n_gpus={rt.world}")

    result = run_config(args.config, args, rt)
    plain_headline = (args.config == "headline" and rt.on_gpu and not args.no_configs
                      and args.size is None and args.det is None and args.batch is None)
    sweep_only = (args.config == "headline" and not plain_headline and not args.no_configs
                  and args.sweep_poses is not None)
    if plain_headline or sweep_only:
        # The other BASELINE configs, short, in the driver's own record.  N = 1: configs 2, 3 and 5
        # with their parity; N > 1: the sweep only (config 5 -- 4096 poses over the N ranks, the
        # informative strong-scaling curve; the headline above is N independent 32-pose steps).
        configs = {}

        def guarded(name, fn):
            return guarded_run(rt.world, name, fn)

        for cfg in (("2", "3", "4", "5") if rt.world == 1 and plain_headline else ("5",)):
            res = guarded(cfg, lambda: summary_of(run_config(cfg, args, rt, short=True)) if rt.rank == 0 or rt.world == 1
                          else run_config(cfg, args, rt, short=True))
            if rt.rank == 0:
                configs[cfg] = res
        if rt.world == 1 and plain_headline:
            # config 3 also at B = 4 (timing only; the parity block above is B = 1's)
            def b4():
                res = run_config("3", args, rt, short=True, batch=4, parity=False)
                out = {k: res[k] for k in ("value", "unit", "ms_per_step", "steps")}
                out["kernels"] = res["roofline"]["kernels"]
                return out
            if "error" not in configs["3"]:
                configs["3"]["b4"] = guarded("3.b4", b4)
            configs["ct"] = guarded("ct", lambda: ct_config(rt))
            configs["few_poses"] = guarded("few_poses", lambda: few_poses_config(rt))
            configs["sparse"] = guarded("sparse", lambda: sparse_config(rt))
        if rt.rank == 0 and "error" not in configs["5"]:
            c5 = configs["5"]
            result["sweep"] = {"metric": c5["metric"], "value": c5["value"], "unit": c5["unit"],
                               "n_gpus": rt.world, "scaling": "strong", "ms_per_step": c5["ms_per_step"],
                               "steps": c5["steps"], "poses": args.sweep_poses or 4096, "poses_per_launch": 512,
                               "what": "bench.py --config 5, short: the candidate sweep of BASELINE "
                                       "configs[4], pose-sharded over the ranks"}
            if rt.world == 1 and plain_headline and c5.get("forward"):
                # the north star's figure at the batch size it is met at: 512 poses per launch
                fs = dict(c5["forward"])
                fs["parity"] = c5["parity"]
                fs["what"] = ("the forward-only kernel on 512 of config 5's candidate poses per launch "
                              "(the sweep's launch size), same volume and detector as the headline")
                result["roofline"]["forward_sweep"] = fs
        if rt.rank == 0 and rt.world == 1 and plain_headline:
            result["configs"] = configs
    if rt.rank == 0:
        result["rccl_ranks"] = rt.rccl_ranks  # (ranks counted by an all_reduce of ones on the devices)
        result["devices_visible"] = torch.cuda.device_count() if rt.on_gpu else 0
        # the full record: a file beside bench.py and stderr, BEFORE the line the driver parses
        full = json.dumps(result)
        try:
            with open(os.path.join(ROOT, "bench_full.json"), "w") as f:
                f.write(full + "\n")
        except OSError as exc:
            log(f"[bench] bench_full.json not written: {exc}")
        log("[bench] full record: " + full)
        sys.stderr.flush()
        print(json.dumps(compact_line(result)), flush=True)

    if rt.world > 1:
        rt.dist.barrier()
        rt.dist.destroy_process_group()


if __name__ == "__main__":
    main()
