"""Pose-batch sharding across the GPUs of one node.

The render path shards by pose with no data-path exchange: every rank holds
the whole volume in its own HBM, renders a contiguous slice of the pose batch
and only the per-pose similarity values (4 B / pose) are gathered -- one
``all_gather`` over RCCL / xGMI (``backend="nccl"`` is RCCL on ROCm; the CPU
tests run the same code over gloo).  The reference has no multi-device code at
all; the closest thing is its one-pose-per-call sweep loop
(notebooks/tutorials/metrics.ipynb:97-175), which this batches and shards.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_poses: int, rank: int, world: int) -> tuple[int, int]:
    """[lo, hi) of rank's contiguous slice; slices differ in size by at most one."""
    q, r = divmod(n_poses, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def all_gather_ragged(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate every rank's (n_r, ...) slice in rank order -> (n_total, ...)."""
    rank, world = _world()
    if world == 1:
        return local
    q = -(-n_total // world)  # padded slice length so that one fixed-size all_gather suffices
    pad = torch.zeros((q,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * q,) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    dist.all_gather_into_tensor(out, pad)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        pieces.append(out[r * q: r * q + (hi - lo)])
    return torch.cat(pieces, dim=0)


@torch.no_grad()
def sweep(drr, metric, fixed, rotations, translations, *, parameterization="euler_angles",
          convention="ZXY", chunk=512, **render_kwargs) -> torch.Tensor:
    """Similarity of ``fixed`` (1,C,H,W) to the DRR at each of the P candidate poses,
    sharded over the ranks of the default process group.  Every rank receives the full
    (P,) result.  ``rotations`` / ``translations`` are the full (P, ...) candidate lists
    (identical on every rank); a rank renders only its slice, ``chunk`` poses per launch."""
    rank, world = _world()
    P = rotations.shape[0]
    lo, hi = shard_bounds(P, rank, world)
    dev = drr.density.device
    if getattr(drr.renderer, "batch_global_range", False) and "alphamin" not in render_kwargs:
        # The marcher's sample positions depend on a marching range it takes over the whole batch
        # of a call (reference renderers.py:220-223): left alone, a sweep's values would depend
        # on how the candidates are cut into chunks and ranks.  One range for the whole
        # candidate list instead: every rank reduces its slice, one all_reduce each way.
        pose_kw = dict(parameterization=parameterization, convention=convention)
        big = torch.finfo(torch.float32).max
        amin = torch.full((), big, device=dev)
        amax = torch.full((), -big, device=dev)
        for a in range(lo, hi, chunk):
            b = min(hi, a + chunk)
            r0, r1 = drr.marching_range(rotations[a:b].to(dev), translations[a:b].to(dev), **pose_kw)
            amin, amax = torch.minimum(amin, r0), torch.maximum(amax, r1)
        if world > 1:
            dist.all_reduce(amin, op=dist.ReduceOp.MIN)
            dist.all_reduce(amax, op=dist.ReduceOp.MAX)
        render_kwargs = dict(render_kwargs, alphamin=amin, alphamax=amax)
    vals = []
    for a in range(lo, hi, chunk):
        b = min(hi, a + chunk)
        img = drr(rotations[a:b].to(dev), translations[a:b].to(dev),
                  parameterization=parameterization, convention=convention, **render_kwargs)
        vals.append(metric(fixed.to(dev).expand(b - a, -1, -1, -1), img))
    local = torch.cat(vals) if vals else torch.empty(0, device=dev)
    return all_gather_ragged(local, P)
