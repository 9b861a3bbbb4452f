"""World-size-2 run of the pose-sharded sweep (diffdrr_amd/dist.py) on CPU over gloo:
the N>1 path of bench.py / SURVEY.md section 8(e) without GPUs.  Each rank renders its
slice of the pose batch (through the host emulation of the kernels, test-only) and the
per-pose similarities are all_gathered; the result must equal the single-process sweep."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(renderer="siddon", P=7, det=(20, 16)):
    from diffdrr_amd import DRR, NormalizedCrossCorrelation2d
    from diffdrr_amd.data import synthetic_subject

    drr = DRR(synthetic_subject(24, kind="phantom", seed=0), sdd=300.0, height=det[0], width=det[1],
              delx=2.0 * 16 / det[1], renderer=renderer)
    g = torch.Generator().manual_seed(5)
    # (P = 7: ragged over 2 ranks, 4 + 3)
    rot = (torch.rand(P, 3, generator=g) - 0.5) * 0.6
    xyz = torch.tensor([0.0, 150.0, 0.0]) + (torch.rand(P, 3, generator=g) - 0.5) * 10.0
    return drr, NormalizedCrossCorrelation2d(), rot, xyz


def _patch_ops():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import build_emu
    from diffdrr_amd import ops
    from diffdrr_amd._lib import DdrrLibrary

    emu = DdrrLibrary(build_emu())
    ops._require_gpu = lambda volume: None
    ops.on_device = lambda t: True
    ops._launch = lambda name, device, *a: emu.call(name, *a, None)


def _worker(rank, world, port, q, renderer="siddon", P=7, det=(20, 16), chunk=3):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    _patch_ops()
    from diffdrr_amd import dist as ddist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        drr, ncc, rot, xyz = _scene(renderer, P, det)
        kw = {"n_points": 60} if renderer == "trilinear" else {}
        with torch.no_grad():
            fixed = drr(torch.zeros(1, 3), torch.tensor([[0.0, 150.0, 0.0]]),
                        parameterization="euler_angles", convention="ZXY", **kw)
        vals = ddist.sweep(drr, ncc, fixed, rot, xyz, chunk=chunk, **kw)
        lo, hi = ddist.shard_bounds(rot.shape[0], rank, world)
        q.put((rank, vals.tolist(), (lo, hi)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_bounds_partition():
    from diffdrr_amd.dist import shard_bounds

    for n in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_pose_sharded_sweep_world2_matches_single_process(emulated_ops):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference of the same sweep (world = 1 path of the same function)
    from diffdrr_amd import dist as ddist

    drr, ncc, rot, xyz = _scene()
    with torch.no_grad():
        fixed = drr(torch.zeros(1, 3), torch.tensor([[0.0, 150.0, 0.0]]),
                    parameterization="euler_angles", convention="ZXY")
    ref = ddist.sweep(drr, ncc, fixed, rot, xyz, chunk=3)
    assert got[0][2] == (0, 4) and got[1][2] == (4, 7)
    for _, vals, _ in got:  # every rank holds the full, identical result
        assert torch.allclose(torch.tensor(vals), ref, rtol=0, atol=1e-6)
    assert ref.shape == (7,) and torch.isfinite(ref).all()


@pytest.mark.timeout(600)
def test_pose_sharded_sweep_world8_ragged_4099_candidates(emulated_ops):
    """BASELINE config 5's partitioning at the node's real world size, with a candidate count that
    does not divide: 4099 poses over 8 ranks = 513 + 513 + 513 + 512 x 5, 512 poses per launch as in
    the bench -- so three ranks end on a 1-pose tail launch -- gathered with one padded all_gather.
    Every rank must hold the single-process result.  (No 8-GPU node was available to any round:
    this is what stands in for the first real SCALE run tripping on a tail shard.)"""
    P, world, det = 4099, 8, (6, 8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "siddon", P, det, 512))
             for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from diffdrr_amd import dist as ddist

    drr, ncc, rot, xyz = _scene("siddon", P, det)
    with torch.no_grad():
        fixed = drr(torch.zeros(1, 3), torch.tensor([[0.0, 150.0, 0.0]]),
                    parameterization="euler_angles", convention="ZXY")
    ref = ddist.sweep(drr, ncc, fixed, rot, xyz, chunk=512)
    sizes = [hi - lo for _, _, (lo, hi) in got]
    assert sizes == [513, 513, 513, 512, 512, 512, 512, 512] and got[-1][2][1] == P
    assert ref.shape == (P,) and torch.isfinite(ref).all() and float(ref.std()) > 1e-3
    for _, vals, _ in got:
        assert torch.allclose(torch.tensor(vals), ref, rtol=0, atol=1e-6)


@pytest.mark.timeout(300)
def test_trilinear_sweep_does_not_depend_on_world_size_or_chunking(emulated_ops):
    """The marcher's sample positions depend on a marching range taken over the batch of a call
    (reference renderers.py:220-223).  `dist.sweep` pins one range for the whole candidate
    list (local reduction + all_reduce MIN / MAX), so a sharded sweep equals the single-process
    one and neither depends on the chunk size -- unlike rendering the chunks naively."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "trilinear")) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from diffdrr_amd import dist as ddist

    drr, ncc, rot, xyz = _scene("trilinear")
    kw = dict(parameterization="euler_angles", convention="ZXY", n_points=60)
    with torch.no_grad():
        fixed = drr(torch.zeros(1, 3), torch.tensor([[0.0, 150.0, 0.0]]), **kw)
        one = ddist.sweep(drr, ncc, fixed, rot, xyz, chunk=3, n_points=60)
        other_chunks = ddist.sweep(drr, ncc, fixed, rot, xyz, chunk=7, n_points=60)
        # what the pinning avoids: every chunk on its own range
        naive = torch.cat([ncc(fixed.expand(b - a, -1, -1, -1), drr(rot[a:b], xyz[a:b], **kw))
                           for a, b in ((0, 3), (3, 6), (6, 7))])
    for _, vals, _ in got:
        assert torch.allclose(torch.tensor(vals), one, rtol=0, atol=1e-6)
    assert torch.allclose(other_chunks, one, rtol=0, atol=1e-6)
    assert (naive - one).abs().max() > 1e-6  # (the ranges really differ between chunks)
    # the range the sweep used is the range of the whole list
    r0, r1 = drr.marching_range(rot, xyz, parameterization="euler_angles", convention="ZXY")
    parts = [drr.marching_range(rot[a:b], xyz[a:b], parameterization="euler_angles",
                                convention="ZXY") for a, b in ((0, 4), (4, 7))]
    assert r0 == min(p[0] for p in parts) and r1 == max(p[1] for p in parts)


@pytest.mark.parametrize("config,extra,world", [("headline", ["--batch", "3", "--sweep-poses", "5"], 2),
                                                ("5", ["--batch", "7"], 2),
                                                ("headline", ["--batch", "2", "--sweep-poses", "4099"], 8)])
def test_bench_harness_spawns_its_ranks(config, extra, world):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself (torch.distributed.run,
    127.0.0.1), runs its step on both and rank 0 prints the contract's JSON line with the world
    size the process group really has.  CPU stand-in: gloo + the host emulation of the kernels
    (bench.py --device cpu, tests/bench_emu_hook.py); tiny sizes."""
    import json
    import subprocess

    env = dict(os.environ, DDRR_BENCH_HOOK="bench_emu_hook",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT,
                                           os.environ.get("PYTHONPATH", "")]))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    det = "16" if world == 2 else "8"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--device", "cpu",
           "--config", config, "--size", "24", "--det", det, "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", *extra]
    if world > 2:
        env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 only
    assert res.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 8192  # (what the driver parses)
    out = json.loads(lines[0])
    assert out["rccl_ranks"] == world  # (counted by an all_reduce of ones, not read from the environment)
    full = [ln for ln in res.stderr.splitlines() if ln.startswith("[bench] full record: ")]
    assert len(full) == 1 and json.loads(full[0].split(": ", 1)[1])["roofline"]["kernels"]
    assert out["n_gpus"] == world and out["steps"] == 2 and out["warmup"] == 1
    assert out["value"] > 0 and out["unit"] == "DRRs/s" and out["higher_is_better"] is True
    if config == "headline":
        gb = int(extra[1]) * world
        assert out["scaling"] == "weak" and out["config"]["global_batch"] == gb
        assert abs(out["value"] - gb * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
        # ... and the strong-scaling figure next to it: config 5's sweep over the same ranks (world 8:
        # 4099 candidates, a ragged split with 1-pose tail launches)
        sw = out["sweep"]
        assert sw["n_gpus"] == world and sw["scaling"] == "strong" and sw["poses"] == int(extra[3])
        assert sw["value"] > 0
    else:
        assert out["scaling"] == "strong" and out["config"]["global_batch"] == 7
    assert out["roofline"]["kernel"] == "ddrr_siddon_forward_bricks" and out["roofline"]["frac"] > 0
