"""``Siddon`` and ``Trilinear``: the reference's renderer modules, on MI355X.

Same constructor arguments and ``forward`` signatures as reference
``diffdrr/renderers.py:11-91`` (Siddon) and ``:186-254`` (Trilinear), same
``(B, 1 or C, N)`` result; the tensor programs behind them are replaced by the
HIP kernels of ``libdiffdrr_hip.so`` through ``torch.autograd.Function``s, so a
``DRR`` module (ours or the reference's) differentiates
w.r.t. pose, ray endpoints and the volume exactly as before.

Every keyword combination the reference renders is rendered and differentiated here, in
float32 and -- a module moved ``.to(torch.float64)``, reference drr.py:71-75 -- float64:

* the fused kernels take what the reference's defaults and tutorials use: sum / max, ``mode``
  nearest / bilinear, any ``align_corners``, ``mask`` channels, per-ray sources, a callable
  ``reducefn`` over the materialised per-segment / per-sample tensor (float32), and the default
  sums in float64 (csrc/f64_rays.hip);
* every other combination -- a mask, a callable, or gradients of ``reducefn="max"`` /
  ``stop_gradients_through_grid_sample`` together with a midpoint lookup (``align_corners=True``,
  Siddon ``mode="bilinear"``); the marcher's ``mode="nearest"`` with a mask; float64 beyond the
  default sums -- goes through the general path (csrc/general_core.h): the tensor the reference
  itself materialises just before it reduces, reduced by ordinary tensor ops.  As in the
  reference, a mask makes the ``reducefn`` irrelevant (renderers.py:73-89).

What raises instead of silently diverging: CPU tensors (there is no CPU fallback) and dtypes
other than float32 / float64.
"""
from __future__ import annotations

import weakref

import torch

from . import ops


def _grid_or_none(renderer, source, target):
    """``renderer.detector_shape`` if the rays really are that row-major affine grid, else None.
    The shape is a CONTRACT (the volume-stationary kernels cull rays with an affine model of
    the grid): ``diffdrr_amd.DRR`` sets it only for rays it generated or checked itself and
    marks it trusted; set by anyone else (a renderer swapped into the reference's ``DRR``,
    INTEGRATION.md) it is checked here, once per call (one reduction, one host sync), and the
    per-ray kernels render whatever does not satisfy it."""
    det = renderer.detector_shape
    if det is None or renderer.trust_detector_shape:
        return det
    return det if ops.rays_form_detector_grid(source, target, int(det[0]), int(det[1])) else None


_label_cache = {}  # id(mask) -> (weakref to the mask, its version, uint8 labels, C)


def _labels_u8(mask: torch.Tensor):
    """Label map as uint8 + channel count (reference: ``C = int(mask.max()) + 1``,
    renderers.py:81 -- a host sync per call there; cached per mask TENSOR here: the entry is
    tied to the live object by a weak reference and dropped with it, so that a new mask that
    happens to land at a freed mask's address can never be served the old labels).
    -> list of (uint8 labels, channels, first channel kept) chunks: the kernels take one byte
    per label, so a map with more than 256 labels is rendered ~255 labels at a time and the
    channel blocks are concatenated (the reference takes any ``mask.max()``).  The first chunk
    holds labels 0 .. 254 as they are (255 = a label of a later chunk: no channel); later chunks
    hold their labels as 1 .. n with 0 = everything else -- including the marcher's samples
    outside the volume, which the lookup's zero padding gives label 0 -- and drop channel 0."""
    ent = _label_cache.get(id(mask))
    if ent is not None and ent[0]() is mask and ent[1] == mask._version:
        return ent[2]
    C = int(mask.max().item()) + 1
    if C <= 256:
        chunks = [(mask.to(torch.uint8).contiguous(), C, 0)]
    else:
        lab = mask.to(torch.int64)
        chunks = [(torch.where(lab < 255, lab, torch.full_like(lab, 255)).to(torch.uint8).contiguous(),
                   255, 0)]
        for c0 in range(255, C, 254):
            n = min(254, C - c0)
            inside = (lab >= c0) & (lab < c0 + n)
            chunks.append((torch.where(inside, lab - c0 + 1, torch.zeros_like(lab))
                           .to(torch.uint8).contiguous(), n + 1, 1))
    key = id(mask)
    _label_cache[key] = (weakref.ref(mask, lambda _, k=key: _label_cache.pop(k, None)),
                         mask._version, chunks)
    return chunks


def _volume_gradient(volume, source, target, img, grad_out, cfg):
    """dLoss/dvolume: the volume-stationary brick kernel for a detector-grid render
    (LDS accumulation, no global atomics), the re-walk with global atomics otherwise."""
    N = target.shape[1]
    grid = (cfg["lookup"] == "step" and cfg["reducefn"] == "sum" and cfg["det"] is not None
            and cfg["det"][0] * cfg["det"][1] == N and source.shape[1] == 1
            and min(cfg["det"]) >= 2)
    if grid and cfg["path"] == "bricks":
        return ops.siddon_backward_volume_bricks(
            volume.shape, source, target, img, grad_out, cfg["det"],
            voxel_shift=cfg["voxel_shift"], eps=cfg["eps"])
    return ops.siddon_backward_volume(
        volume, source, target, img, grad_out, voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
        reducefn=cfg["reducefn"], det=cfg["det"], tile=cfg["tile"])


def _record_vmax(volume, want_aux, cfg):
    """Scale of the packed backward record (csrc/record_pack.h): max |volume|, cached while the
    volume buffer does not change; 0 (= float record) when the volume is being optimised (it
    changes every step) or the packed form is switched off."""
    if not want_aux or not cfg.get("packed_record", False) or volume.requires_grad:
        return 0.0
    return ops.volume_absmax(volume)


_cu_count = {}


# Which storage renders a volume with FEW double bricks (32 x 32 x 64) per CU faster, by poses per
# launch: measured, not one constant (tools/storage_table.py, profiles/r06/storage_table.txt; the two
# storages are within ~10 % of each other everywhere in this regime, forward and with the record).
#   (double bricks per CU below, poses per launch from, to): "q16p" inside, "f32" outside
#   < 1.5  (256^3: 1.00)             q16p at <= 5 poses: half as many work items, and a few poses pay per
#                                    item; from 8 poses on the 512 fp32 bricks balance better (-15 ... -24 %)
#   < 2.5  (384 x 384 x 256: 2.25)   q16p throughout (-3 ... -10 %)
#   < 4    (512 x 512 x 133: 3.00,   q16p at 8 ... 12 poses (-5 ... -10 % with the record, a tie without);
#           the example CT's shape)  f32 below (a one-pose launch: 0.066 against 0.077 ms) and beyond
#                                    (32 poses: 0.298 against 0.335 ms)
#   >= 4                             q16p (512^3: 8 per CU)
_FEW_BRICKS_POLICY = ((1.5, 1, 5), (2.5, 1, 1 << 30), (4.0, 8, 12))


def _few_bricks_take_q16(per_cu: float, poses) -> bool:
    """The table above: a volume with ``per_cu`` < 4 double bricks per CU, ``poses`` per launch."""
    if poses is None:
        return False
    lo, hi = next((a, b) for r, a, b in _FEW_BRICKS_POLICY if per_cu < r)
    return lo <= int(poses) <= hi


def _brick_storage(volume, cfg, poses=None):
    """How the brick kernel stages the volume (Siddon.brick_storage).  fp32 bricks whatever the
    setting for a volume that is being optimised (it changes every step: its 16-bit ranges would
    be recomputed per launch, one more pass over the volume, and its gradient is taken w.r.t. the
    exact values); for a volume with fewer than 4 double bricks per CU -- too few to balance over
    the persistent workgroups whatever the batch -- the measured table above decides by the poses
    of the launch (``poses`` = None: unknown, fp32 bricks as until round 5)."""
    storage = cfg.get("storage", "f32")
    if storage not in ("q16", "q16p") or volume.requires_grad:
        return "f32"
    if not ops.brick_storage_applies(volume):
        return "f32"
    dev = volume.device
    if (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
            and not cfg.get("static_volume", False)):
        # A captured graph would bake in the workspace's address and "already built": replays
        # after an in-place update of the volume would render the OLD bricks.  fp32 bricks read
        # the live volume.  (Siddon.static_volume = True promises that the volume is not edited
        # between replays -- registration.GraphedIteration sets it.)
        return "f32"
    if dev.type == "cuda":
        if dev not in _cu_count:
            _cu_count[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
        dx, dy, dz = volume.shape
        per_cu = (-(-dx // 32)) * (-(-dy // 32)) * (-(-dz // 64)) / _cu_count[dev]
        if per_cu < 4 and not _few_bricks_take_q16(per_cu, poses):
            return "f32"
    if ops.workspace_churn(volume, storage) >= 3:
        # edited in place between renders again and again (a reconstruction loop on a plain
        # tensor): every render would pay the pass over the volume that builds the workspace
        return "f32"
    return storage


class _SiddonFn(torch.autograd.Function):
    """out (B,N) = img * sum_k V_k dalpha_k (or max_k).  Inputs: volume, source,
    target, img.  Backward: ddrr_siddon_backward_rays from the 8-float forward
    record (pose / ray gradients, elementwise) and ddrr_siddon_backward_volume
    (atomic scatter), replacing SortBackward + grid_sampler_3d_backward."""

    @staticmethod
    def forward(ctx, volume, source, target, img, cfg):
        need_rays = any(ctx.needs_input_grad[1:4])
        want_aux = bool(need_rays and cfg["lookup"] == "step")
        N = target.shape[1]
        grid = (cfg["lookup"] == "step" and cfg["reducefn"] == "sum" and cfg["det"] is not None
                and cfg["det"][0] * cfg["det"][1] == N and source.shape[1] == 1
                and min(cfg["det"]) >= 2)
        if grid and cfg["path"] == "bricks":
            # detector-grid fast path: volume-stationary LDS bricks (volume read once)
            out, aux = ops.siddon_forward_bricks(
                volume, source, target, img, cfg["det"], voxel_shift=cfg["voxel_shift"],
                eps=cfg["eps"], want_aux=want_aux, record_vmax=_record_vmax(volume, want_aux, cfg),
                storage=_brick_storage(volume, cfg, source.shape[0]))
        else:
            out, aux, _ = ops.siddon_forward(
                volume, source, target, img, voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
                reducefn=cfg["reducefn"], lookup=cfg["lookup"],
                align_corners=cfg["align_corners"], want_aux=want_aux, det=cfg["det"],
                tile=cfg["tile"])
        ctx.cfg = cfg
        ctx.has_aux = want_aux
        ctx.save_for_backward(volume, source, target, img, aux if want_aux else None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        volume, source, target, img, aux = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_s, need_t, need_i = ctx.needs_input_grad[:4]
        stop = cfg["stop_gradients"]
        if cfg["lookup"] != "step":
            # midpoint lookups (align_corners=True, mode="bilinear"): one more walk
            if cfg["reducefn"] != "sum":
                raise NotImplementedError("gradients of the midpoint lookups need reducefn='sum'")
            if stop:
                raise NotImplementedError("stop_gradients_through_grid_sample is implemented for "
                                          "mode='nearest', align_corners=False")
            gs, gt, gi, gv = ops.siddon_backward_midpoint(
                volume, source, target, img, grad_out, voxel_shift=cfg["voxel_shift"],
                eps=cfg["eps"], lookup=cfg["lookup"], align_corners=cfg["align_corners"],
                want_rays=bool(need_s or need_t), want_img=bool(need_i),
                want_volume=bool(need_vol))
            g_s = None
            if need_s:
                g_s = gs.sum(dim=1, keepdim=True) if source.shape[1] == 1 else gs
            return gv, g_s, (gt if need_t else None), \
                (gi.view_as(img) if need_i else None), None
        g_vol = g_s = g_t = g_i = None
        grad_out = grad_out.contiguous()
        if need_s or need_t or need_i:
            gs, gt, gi = ops.siddon_backward_rays(
                aux, grad_out, source, target, img, eps=cfg["eps"], reducefn=cfg["reducefn"],
                want_img_grad=bool(need_i and not stop))
            if need_s:
                g_s = gs.sum(dim=1, keepdim=True) if source.shape[1] == 1 else gs
            if need_t:
                g_t = gt
            if need_i and not stop:
                g_i = gi.view_as(img)
        if need_vol and not stop:
            g_vol = _volume_gradient(volume, source, target, img, grad_out, cfg)
        return g_vol, g_s, g_t, g_i, None


class _SiddonF64Fn(torch.autograd.Function):
    """The Siddon renderer in double precision (csrc/f64_rays.hip): what a reference module
    moved `.to(torch.float64)` computes (drr.py:71-75).  mode="nearest", align_corners=False;
    reducefn sum (differentiable) or max (forward)."""

    @staticmethod
    def forward(ctx, volume, source, target, img, cfg):
        need_rays = any(ctx.needs_input_grad[1:4])
        out, aux = ops.siddon_forward_f64(
            volume, source, target, img, voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
            reducefn=cfg["reducefn"], want_aux=bool(need_rays and cfg["reducefn"] == "sum"))
        ctx.cfg = cfg
        ctx.save_for_backward(volume, source, target, img, aux)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        volume, source, target, img, aux = ctx.saved_tensors
        cfg = ctx.cfg
        if cfg["reducefn"] != "sum":
            raise NotImplementedError("float64 gradients are implemented for reducefn='sum'")
        need_vol, need_s, need_t, need_i = ctx.needs_input_grad[:4]
        stop = cfg["stop_gradients"]
        gs, gt, gi, gv = ops.siddon_backward_f64(
            volume.shape, source, target, img, grad_out.contiguous(), aux,
            voxel_shift=cfg["voxel_shift"], eps=cfg["eps"], want_rays=bool(need_s or need_t),
            want_img=bool(need_i and not stop), want_volume=bool(need_vol and not stop))
        g_s = None
        if need_s:
            g_s = gs.sum(dim=1, keepdim=True) if source.shape[1] == 1 else gs
        return gv, g_s, (gt if need_t else None), (gi.view_as(img) if gi is not None else None), None


class _TrilinearF64Fn(torch.autograd.Function):
    """The marcher in double precision (csrc/f64_rays.hip): mode="bilinear", reducefn sum,
    align_corners=False."""

    @staticmethod
    def forward(ctx, volume, source, target, img, alphamin, alphamax, cfg):
        out = ops.trilinear_forward_f64(volume, source, target, img, alphamin, alphamax,
                                        n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"],
                                        eps=cfg["eps"])
        ctx.cfg = cfg
        ctx.save_for_backward(volume, source, target, img, alphamin, alphamax)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        volume, source, target, img, alphamin, alphamax = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_s, need_t, need_i, need_a0, need_a1 = ctx.needs_input_grad[:6]
        r = ops.trilinear_backward_f64(
            volume, source, target, img, grad_out, alphamin, alphamax, n_points=cfg["n_points"],
            voxel_shift=cfg["voxel_shift"], eps=cfg["eps"], want_rays=bool(need_s or need_t),
            want_img=bool(need_i), want_alpha=bool(need_a0 or need_a1), want_volume=bool(need_vol))
        g_s = g_t = g_a0 = g_a1 = g_i = None
        if need_s:
            g_s = r["g_source"].sum(dim=1, keepdim=True) if source.shape[1] == 1 \
                else r["g_source"]
        if need_t:
            g_t = r["g_target"]
        if need_i:
            g_i = r["g_img"].view_as(img)
        if need_a0 or need_a1:
            ga = r["g_alpha"].sum(dim=(0, 1))
            g_a0 = ga[0].reshape(alphamin.shape) if need_a0 else None
            g_a1 = ga[1].reshape(alphamax.shape) if need_a1 else None
        return r["g_volume"], g_s, g_t, g_i, g_a0, g_a1, None


class _SiddonChannelsFn(torch.autograd.Function):
    """mask_to_channels (renderers.py:77-89): out (B,C,N), channel c = the line integral over
    the voxels labelled c.  Backward: one ddrr_siddon_backward_channels launch (the ray is
    re-walked over the volume weighted by the incoming gradient of each voxel's channel),
    replacing ScatterAddBackward + SortBackward + grid_sampler_3d_backward."""

    @staticmethod
    def forward(ctx, volume, source, target, img, labels, C, cfg):
        out = _channels_forward(volume, labels, C, source, target, img, cfg)
        ctx.cfg = cfg
        ctx.save_for_backward(volume, source, target, img, labels)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        volume, source, target, img, labels = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_s, need_t, need_i = ctx.needs_input_grad[:4]
        stop = cfg["stop_gradients"]
        want_vol = bool(need_vol and not stop)
        B, C, N = grad_out.shape
        if _channels_use_bricks(cfg, source, N) and ops.channels_fit_bricks(B, C, N):
            # the DRR case: the record of the gradient-weighted volume on the bricks
            # (ddrr_siddon_backward_channels_bricks), 4-6x faster than the per-ray re-walk, and the
            # volume gradient with the brick in LDS as its accumulator
            gs = gt = gi = gv = None
            if need_s or need_t or (need_i and not stop):
                gs, gt, gi = ops.siddon_backward_channels_bricks(
                    volume, labels, source, target, img, grad_out, cfg["det"],
                    voxel_shift=cfg["voxel_shift"], eps=cfg["eps"], want_img=bool(need_i and not stop))
            if want_vol:
                gv = ops.siddon_backward_channels_volume_bricks(
                    labels, source, target, img, grad_out, cfg["det"],
                    voxel_shift=cfg["voxel_shift"], eps=cfg["eps"]).to(volume.dtype)
        else:
            gs, gt, gi, gv = ops.siddon_backward_channels(
                volume, labels, source, target, img, grad_out, voxel_shift=cfg["voxel_shift"],
                eps=cfg["eps"], want_rays=bool(need_s or need_t), want_img=bool(need_i and not stop),
                want_volume=want_vol, det=cfg["det"], tile=cfg["tile"])
        g_s = g_t = None
        if need_s:
            g_s = gs.sum(dim=1, keepdim=True) if source.shape[1] == 1 else gs
        if need_t:
            g_t = gt
        g_i = gi.view_as(img) if gi is not None else None
        return gv, g_s, g_t, g_i, None, None, None


class _SiddonSegmentsFn(torch.autograd.Function):
    """The (B, N, M-1) per-segment tensor the reference hands to a callable ``reducefn``
    (renderers.py:70-71, 175-183), materialised by ddrr_siddon_segments; backward =
    ddrr_siddon_segments_backward (one more walk weighted by the incoming gradient)."""

    @staticmethod
    def forward(ctx, volume, source, target, img, cfg):
        terms = ops.siddon_segments(volume, source, target, img, voxel_shift=cfg["voxel_shift"],
                                    eps=cfg["eps"])
        ctx.cfg = cfg
        ctx.save_for_backward(volume, source, target, img)
        return terms.transpose(1, 2)  # (B, N, M-1), the reference's layout (a view)

    @staticmethod
    def backward(ctx, grad):
        volume, source, target, img = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_s, need_t, need_i = ctx.needs_input_grad[:4]
        stop = cfg["stop_gradients"]
        gs, gt, gi, gv = ops.siddon_segments_backward(
            volume, source, target, img, grad.transpose(1, 2), voxel_shift=cfg["voxel_shift"],
            eps=cfg["eps"], want_rays=bool(need_s or need_t), want_img=bool(need_i and not stop),
            want_volume=bool(need_vol and not stop))
        g_s = g_t = None
        if need_s:
            g_s = gs.sum(dim=1, keepdim=True) if source.shape[1] == 1 else gs
        if need_t:
            g_t = gt
        g_i = gi.view_as(img) if gi is not None else None
        return gv, g_s, g_t, g_i, None


class _SiddonTermsFn(torch.autograd.Function):
    """The general path: the (B, N, M-1) per-segment tensor `img * value * interval`
    (renderers.py:66-71) for ANY lookup (mode, align_corners) in float32 or float64, with its
    autograd (csrc/general_core.h).  What is not one of the fused kernels' cases reduces this
    tensor with ordinary tensor ops, exactly as the reference does."""

    @staticmethod
    def forward(ctx, volume, source, target, img, cfg):
        terms = ops.siddon_segments_general(
            volume, source, target, img, voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
            lookup=cfg["lookup"], align_corners=cfg["align_corners"])
        ctx.cfg = cfg
        ctx.save_for_backward(volume, source, target, img)
        return terms.transpose(1, 2)

    @staticmethod
    def backward(ctx, grad):
        volume, source, target, img = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_s, need_t, need_i = ctx.needs_input_grad[:4]
        through = not cfg["stop_gradients"]
        gs, gt, gi, gv = ops.siddon_segments_general_backward(
            volume, source, target, img, grad.transpose(1, 2), voxel_shift=cfg["voxel_shift"],
            eps=cfg["eps"], lookup=cfg["lookup"], align_corners=cfg["align_corners"],
            through_lookup=through, want_rays=bool(need_s or need_t), want_img=bool(need_i),
            want_volume=bool(need_vol))
        g_s = None
        if need_s:
            g_s = gs.sum(dim=1, keepdim=True) if source.shape[1] == 1 else gs
        return gv, g_s, (gt if need_t else None), \
            (gi.view_as(img) if gi is not None and need_i else None), None


def _scatter_channels(terms, labels, mask):
    """mask_to_channels (renderers.py:77-89, 242-252) on materialised tensors: `terms` and the
    integer `labels` are (B, K, N); channel c collects the terms whose label is c."""
    B, _, N = terms.shape
    C = int(mask.max().item()) + 1
    return torch.zeros(B, C, N, dtype=terms.dtype, device=terms.device).scatter_add_(1, labels, terms)


def _reduce_terms(terms, reducefn):
    """reference `reduce` (renderers.py:175-183) over the last dim of a materialised tensor"""
    if reducefn == "sum":
        return terms.sum(dim=-1)
    if reducefn == "max":
        return terms.max(dim=-1).values
    if callable(reducefn):
        return reducefn(terms)
    raise ValueError(f"Only supports reducefn 'sum' or 'max', not {reducefn}")


class _SiddonPoseFn(torch.autograd.Function):
    """The DRR case end to end: world pose per DRR -> image.  Inputs: volume, Mw (B,3,4)
    (extrinsic o reorient), P (N,3) calibrated detector points, Ainv (3,4) world -> voxel.
    Forward = fused ray generation (ddrr_raygen_forward: detector.py:151-153 +
    drr.py:201-205) + the Siddon kernel; backward = ddrr_siddon_backward_pose (ray
    gradients chained through the ray generation and reduced to dLoss/dMw in one kernel)
    and, if asked for, the volume-gradient scatter."""

    @staticmethod
    def forward(ctx, volume, Mw, P, Ainv, cfg):
        source, target, img = ops.raygen_forward(Mw, Ainv, P)
        want_aux = bool(ctx.needs_input_grad[1])
        if cfg["path"] == "bricks":
            out, aux = ops.siddon_forward_bricks(
                volume, source, target, img, cfg["det"], voxel_shift=cfg["voxel_shift"],
                eps=cfg["eps"], want_aux=want_aux, record_vmax=_record_vmax(volume, want_aux, cfg),
                storage=_brick_storage(volume, cfg, source.shape[0]), pixel_mask=cfg.get("pixel_mask"))
        else:
            out, aux, _ = ops.siddon_forward(
                volume, source, target, img, voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
                want_aux=want_aux, det=cfg["det"], tile=cfg["tile"])
        ctx.cfg = cfg
        ctx.save_for_backward(volume, Mw, P, Ainv, source, target, img, aux)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        volume, Mw, P, Ainv, source, target, img, aux = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_M = ctx.needs_input_grad[:2]
        stop = cfg["stop_gradients"]
        g_vol = g_M = None
        grad_out = grad_out.contiguous()
        if need_M:
            g_M = ops.siddon_backward_pose(aux, grad_out, source, target, img, Mw, Ainv, P,
                                           eps=cfg["eps"], with_img_path=not stop)
        if need_vol and not stop:
            g_vol = _volume_gradient(volume, source, target, img, grad_out, cfg)
        return g_vol, g_M, None, None, None


class _EulerSiddonNccFn(torch.autograd.Function):
    """A registration step's forward and backward around the brick kernel in three launches
    instead of nine: (rot, xyz) Euler pose parameters -> per-pose NCC of the DRR with a fixed
    image.  Forward: ddrr_pose_raygen_forward (pose -> matrix -> rays), the brick kernel with its
    record (no image is written), ddrr_siddon_ncc_forward (image from the record + NCC).
    Backward: ddrr_siddon_ncc_backward_pose (NCC backward, the record's ray gradients chained
    through the ray generation, matrix -> pose parameters).  The arithmetic per element is that
    of ``NormalizedCrossCorrelation2d()(fixed, drr(rot, xyz, parameterization="euler_angles"))``
    (reference registration.py:32-42, metrics.py:21-44); only the order of the sums differs."""

    @staticmethod
    def forward(ctx, rot, xyz, volume, reorient34, P, Ainv, fixed, axes, cfg, ncc_eps, reduce_sum=False):
        # (the record and the brick counter of the render are cleared by the launch in front of it)
        B, N = rot.shape[0], P.shape[0]
        aux = ops.brick_record_buffer(B, N, rot.device)
        launch_ws = ops.launch_workspace(volume.shape, volume.device)
        some = B > 0 and N > 0  # (an empty batch launches nothing)
        Mw, source, target, img = ops.pose_raygen_forward(rot, xyz, axes, reorient34, Ainv, P,
                                                          clear=aux if some else None,
                                                          clear_launch_ws=launch_ws if some else None)
        ops.siddon_forward_bricks(
            volume, source, target, img, cfg["det"], voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
            want_aux=True, storage=_brick_storage(volume, cfg, B), want_image=False, aux=aux,
            launch_ws=launch_ws, cleared=some)
        ncc, stats, _, total = ops.siddon_ncc_forward(aux, img, fixed, ncc_eps, want_sum=True) if reduce_sum \
            else (*ops.siddon_ncc_forward(aux, img, fixed, ncc_eps), None)
        ctx.axes, ctx.cfg = axes, cfg
        ctx.save_for_backward(rot, xyz, reorient34, P, Ainv, fixed, Mw, source, target, img, aux, stats)
        # (reduce_sum: the batch's objective sum_b ncc_b, put together by the epilogue's own launch; its
        # gradient arrives as ONE value, which the backward epilogue reads with stride 0; the per-pose
        # values ride along, not differentiable)
        if reduce_sum:
            # (the per-pose values take no gradient -- and autograd is not to fill one with zeros
            # for them: one launch per step)
            ctx.mark_non_differentiable(ncc)
            ctx.set_materialize_grads(False)
            return total, ncc
        return ncc

    @staticmethod
    def backward(ctx, g, _g_values=None):
        if g is None:  # (reduce_sum with nothing flowing back)
            return (None,) * 11
        rot, xyz, reorient34, P, Ainv, fixed, Mw, source, target, img, aux, stats = ctx.saved_tensors
        g_rot, g_xyz = ops.siddon_ncc_backward_pose(
            aux, img, fixed, stats, g, source, target, Mw, Ainv, P, rot, xyz, ctx.axes, reorient34,
            eps=ctx.cfg["eps"], with_img_path=not ctx.cfg["stop_gradients"])
        return g_rot, g_xyz, None, None, None, None, None, None, None, None, None


class _EulerSiddonImageFn(torch.autograd.Function):
    """``drr(rot, xyz, parameterization="euler_angles")`` for pose parameters that take a gradient, when the
    similarity is computed OUTSIDE the render (reference registration.py:32-42 with any criterion of
    metrics.py, or a user's own loss): forward ddrr_pose_raygen_forward (pose -> matrix -> rays, and the
    clears of the render behind it) + the brick kernel with its record + the image from the record -- three
    launches for five; backward ddrr_siddon_backward_pose_euler -- one for three (a fill, the record's ray
    gradients reduced to dL/dMw, matrix -> pose parameters).  Same arithmetic per element as
    ``_PoseEulerFn`` + ``_SiddonPoseFn``; (rot, xyz) -> image (B, N)."""

    @staticmethod
    def forward(ctx, rot, xyz, volume, reorient34, P, Ainv, axes, cfg):
        B, N = rot.shape[0], P.shape[0]
        aux = ops.brick_record_buffer(B, N, rot.device)
        launch_ws = ops.launch_workspace(volume.shape, volume.device)
        Mw, source, target, img = ops.pose_raygen_forward(rot, xyz, axes, reorient34, Ainv, P,
                                                          clear=aux, clear_launch_ws=launch_ws)
        out, _ = ops.siddon_forward_bricks(
            volume, source, target, img, cfg["det"], voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
            want_aux=True, storage=_brick_storage(volume, cfg, B), aux=aux, launch_ws=launch_ws, cleared=True)
        ctx.axes, ctx.cfg = axes, cfg
        ctx.save_for_backward(rot, xyz, reorient34, P, Ainv, Mw, source, aux)
        return out

    @staticmethod
    def backward(ctx, g):
        rot, xyz, reorient34, P, Ainv, Mw, source, aux = ctx.saved_tensors
        g_rot, g_xyz = ops.siddon_backward_pose_euler(
            aux, g, source, Mw, Ainv, P, rot, xyz, ctx.axes, reorient34, eps=ctx.cfg["eps"],
            with_img_path=not ctx.cfg["stop_gradients"])
        return g_rot, g_xyz, None, None, None, None, None, None


def _cat_channels(blocks):
    """Channel blocks of the label chunks side by side (one chunk -- up to 256 labels -- is the
    result itself: no copy of the (B, C, N) tensor)."""
    blocks = list(blocks)
    return blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=1)


def _chunk(block, k0):
    """A later label chunk's block without its placeholder channels below k0."""
    return block if k0 == 0 else block[:, k0:]


def _channels_use_bricks(cfg, source, N):
    """Whether a channel render / its ray backward may take the volume-stationary kernels: a
    detector grid, one source per pose, the brick path switched on."""
    grid = (cfg["det"] is not None and cfg["det"][0] * cfg["det"][1] == N
            and source.shape[1] == 1 and min(cfg["det"]) >= 2)
    return bool(grid and cfg["path"] == "bricks" and cfg.get("channels_on_bricks", True))


def _channels_forward(volume, labels, C, source, target, img, cfg):
    """(B, C, N) channel render: the volume-stationary kernel for a detector grid (the label
    rides in the staged voxel word), the per-ray channel kernel otherwise."""
    B, N = target.shape[:2]
    if _channels_use_bricks(cfg, source, N) and ops.channels_fit_bricks(B, C, N):
        # a (volume, label map) pair that is rendered again: from its ready-packed words (cached, +100 %
        # of the volume's bytes; kept in step with both by the call itself, ops.channel_words)
        words = None
        if (cfg.get("channel_words", True) and not volume.requires_grad and volume.is_contiguous()
                and labels.is_contiguous() and B > 0
                and not (volume.device.type == "cuda" and torch.cuda.is_current_stream_capturing())):
            words = ops.channel_words(volume, labels, C, build=False)
        return ops.siddon_forward_channels_bricks(
            volume, labels, C, source, target, img, cfg["det"],
            voxel_shift=cfg["voxel_shift"], eps=cfg["eps"], words=words)
    return ops.siddon_forward_channels(
        volume, labels, C, source.contiguous(), target.contiguous(), img.contiguous(),
        voxel_shift=cfg["voxel_shift"], eps=cfg["eps"], det=cfg["det"], tile=cfg["tile"])


class _RaygenFn(torch.autograd.Function):
    """Fused ray generation as a differentiable op: world pose per DRR ``Mw`` (B,3,4), calibrated
    detector points ``P`` (N,3), world -> voxel ``Ainv`` (3,4)  ->  voxel-space source (B,1,3),
    target (B,N,3) and world ray length (B,N) in one kernel (ddrr_raygen_forward: reference
    detector.py:151-153, drr.py:201-205) instead of ~80 small tensor ops.  Backward: the adjoint
    of raygen_core.h's raygen_ray_adjoint restated on the (B, N, 3) tensors -> dLoss/dMw.
    (``_SiddonPoseFn`` fuses that adjoint into its pose-gradient kernel; this op serves the
    renderers whose backward yields per-ray endpoint gradients: channels, the marcher.)"""

    @staticmethod
    def forward(ctx, Mw, P, Ainv):
        source, target, img = ops.raygen_forward(Mw, Ainv, P)
        ctx.save_for_backward(Mw, P, Ainv, img)
        ctx.set_materialize_grads(False)  # an unused output's gradient arrives as None
        return source, target, img

    @staticmethod
    def backward(ctx, gs, gt, gi):
        Mw, P, Ainv, img = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        A, R = Ainv[:, :3], Mw[:, :, :3]
        B, N = img.shape
        g_tw = gt @ A if gt is not None else Mw.new_zeros(B, N, 3)  # transpose of Ainv's 3x3 block
        g_sw = gs.sum(dim=1) @ A if gs is not None else Mw.new_zeros(B, 3)
        if gi is not None:                                  # img = ||tw - sw||
            ku = (gi / img.clamp_min(1e-30)).unsqueeze(-1) * (P @ R.transpose(1, 2))
            g_tw = g_tw + ku
            g_sw = g_sw - ku.sum(dim=1)
        g_R = torch.einsum("bna,nj->baj", g_tw, P)
        g_T = g_tw.sum(dim=1) + g_sw                        # tw and sw both carry the translation
        return torch.cat([g_R, g_T.unsqueeze(-1)], dim=-1), None, None


class Siddon(torch.nn.Module):
    """Differentiable X-ray renderer: Siddon's exact ray tracing (reference
    renderers.py:11-91) as one fused gfx950 kernel per call."""

    def __init__(
        self,
        voxel_shift: float = 0.5,
        mode: str = "nearest",
        stop_gradients_through_grid_sample: bool = False,
        filter_intersections_outside_volume: bool = False,
        reducefn: str = "sum",
        eps: float = 1e-8,
    ):
        super().__init__()
        if mode not in ("nearest", "bilinear"):
            raise ValueError(f"mode must be 'nearest' or 'bilinear', not {mode}")
        self.mode = mode
        self.stop_gradients_through_grid_sample = stop_gradients_through_grid_sample
        # The reference's branch (renderers.py:116-121) cannot run: it calls _get_alpha_minmax
        # without `voxel_shift` (:118 vs :124), a TypeError on every call.  Its INTENDED semantics
        # -- drop the sorted-crossing columns that lie outside [alphamin, alphamax] for every ray --
        # remove only segments outside the volume, which add nothing (zero padding): image and
        # every gradient equal the default render's.  Pinned by tests/golden/
        # siddon_filter_outside.npz (the reference with that one call completed, made by
        # tests/golden/make_golden_filter.py).  The fused walks never visit crossings outside
        # the volume, so the flag changes nothing here; with a ray endpoint INSIDE the volume the
        # intended filter would also clamp to alpha in [0, 1] where every ray agrees -- the
        # product integrates the whole line, as without the flag.
        self.filter_intersections_outside_volume = filter_intersections_outside_volume
        self.reducefn = reducefn
        self.voxel_shift = voxel_shift
        self.eps = eps
        # set by DRR: the detector grid the rays form (a contract, see _grid_or_none) and the
        # wave tile shape (a hint)
        self.detector_shape = None
        self.trust_detector_shape = False
        self.tile = None
        # which kernel renders a detector-grid call (same results to ~1e-6):
        # "bricks" (volume-stationary, brick_core.h / brick_step.h) or "generic" (per-ray walk)
        self.grid_path = "bricks"
        # Opt-in: the brick kernel's backward record in 32-bit fixed point (csrc/record_pack.h):
        # 3 atomics per ray and brick instead of 5 (it was 7 % faster than the float record of round 2;
        # since the blocked float record of round 3 it is 11 % slower: 1.57 vs 1.41 ms) and exact,
        # order-independent sums, i.e. bit-reproducible pose gradients; per brick piece it
        # resolves 2 max|V| (Dx+Dy+Dz+3) / 2^30, ~10x coarser than fp32 accumulation, hence off
        # by default: the default record is fp32 like the reference's arithmetic.
        self.packed_record = False
        # How the brick kernel holds a brick in LDS (include/diffdrr_hip.h DDRR_BRICKS_*):
        # "q16": 16-bit block quantisation, one (min, step) pair per 32 x 32 x 64 brick -- |error|
        # per voxel <= the brick's value range / 131070, all arithmetic fp32; measured image
        # error against the fp64 oracle at 512^3: 5e-6 of the image scale, the same as with fp32
        # bricks (the reference's own fp32 arithmetic: 6e-5) -- in exchange for bricks of twice
        # the volume: 6-7 % faster.  "f32": the volume's own values in 32^3 bricks (always used
        # for a volume that requires grad).  "q16p": the same bricks, staged from a packed copy that
        # is kept with the cached per-volume workspace (ops.brick_workspace: +52 % of the volume's
        # bytes, built on the first render after the volume changed, +0.35 ms at 512^3): same
        # results; one pose per launch -23 %, 32 poses -3 %.  (fp32 bricks from a packed copy with
        # the same look-ahead were built and measured in round 5: within 1-3 % of "f32" at 512^3
        # and 256^3, 45 % SLOWER on the 133-slice example shape -- not kept,
        # profiles/r05/f32_packed_lookahead_experiment.txt.)
        self.brick_storage = "q16p"
        # Under HIP-graph capture the 16-bit storages are only used if the caller promises that
        # the volume is not edited in place between replays (the graph bakes in the cached
        # workspace): otherwise captured renders use fp32 bricks, which read the live volume.
        self.static_volume = False
        # mask_to_channels of a detector-grid call on the brick kernel: the label rides in the low
        # byte of the staged voxel word, the value keeps a 16-bit mantissa (2^-17 relative per
        # voxel; channel sums agree with the plain render to 1e-5 of the image scale,
        # tests/test_gpu_parity.py).  False: the per-ray channel kernel on exact fp32 values.
        self.channels_on_bricks = True
        # ... from the volume's ready-packed words once a (volume, label map) pair is rendered a second
        # time (ops.channel_words: +100 % of the volume's bytes per pair, self-healing on the device;
        # one pose 0.089 -> 0.079 ms, 8 poses 0.280 -> 0.262 = 1.50x the plain render on the reference's
        # example shape and label map).  False: staged from volume and label map every time.
        self.channel_words = True

    def dims(self, volume):
        return torch.tensor(volume.shape).to(volume)

    @staticmethod
    def volume_changed(volume):
        """Tell the renderer that ``volume`` was edited in a way PyTorch does not track
        (``volume.data[...] = x``): its cached 16-bit bricks are rebuilt by the next render.
        Tracked edits (any in-place op on the tensor itself, ``volume.data = other``) need no call;
        an untracked edit without this call is still rendered from the live values -- the launch
        checks a fingerprint of the volume and falls back to fp32 bricks (``ops.brick_workspace``)
        -- only slower, until this is called."""
        ops.invalidate_brick_workspace(volume)

    def _cfg(self, align_corners, det="unchecked", reducefn=None):
        if self.mode == "bilinear":
            lookup = "mid_trilinear"
        elif align_corners:
            lookup = "mid_nearest"
        else:
            lookup = "step"
        reducefn = self.reducefn if reducefn is None else reducefn
        ops.reduce_code(reducefn)  # validates / raises like reference `reduce`
        return {"voxel_shift": self.voxel_shift, "eps": self.eps, "reducefn": reducefn,
                "lookup": lookup, "align_corners": bool(align_corners),
                "stop_gradients": self.stop_gradients_through_grid_sample,
                "det": self.detector_shape if det == "unchecked" else det, "tile": self.tile,
                "path": self.grid_path,
                "packed_record": self.packed_record, "storage": self.brick_storage,
                "static_volume": self.static_volume,
                "channels_on_bricks": self.channels_on_bricks, "channel_words": self.channel_words}

    def supports_pose_entry(self):
        """Whether ``render_poses`` (the fused DRR entry) computes what ``forward`` would."""
        return self.mode == "nearest" and self.reducefn == "sum"

    def render_poses(self, volume, Mw, P, Ainv, mask=None, pixel_mask=None):
        """The DRR case without materialising the ray tensors in PyTorch: ``Mw`` (B,3,4)
        world pose per DRR (extrinsic o reorient), ``P`` (N,3) calibrated detector points,
        ``Ainv`` (3,4) world -> voxel.  Equals ``forward(volume, *rays(Mw, P, Ainv), mask=mask)``;
        -> (B, 1, N), or (B, C, N) with a mask.  ``pixel_mask`` (``Detector.subsample_mask``, plain
        render on the bricks only): the pixels rendered, zeros at the others."""
        cfg = self._cfg(False)
        if pixel_mask is not None and mask is None and cfg["path"] == "bricks":
            cfg["pixel_mask"] = pixel_mask
        if mask is not None:
            source, target, img = _RaygenFn.apply(Mw, P, Ainv)
            return _cat_channels(_chunk(_SiddonChannelsFn.apply(volume, source, target, img, labels, C, cfg), k0)
                                 for labels, C, k0 in _labels_u8(mask))
        return _SiddonPoseFn.apply(volume, Mw, P, Ainv, cfg).unsqueeze(1)

    def _general(self, volume, source, target, img, lookup, align_corners, mask):
        """Every other keyword combination the reference renders (csrc/general_core.h): the
        materialised per-segment tensor, reduced as the reference reduces it."""
        B, N, _ = target.shape
        dt = volume.dtype
        source, target = source.to(dt), target.to(dt)
        gcfg = {"voxel_shift": self.voxel_shift, "eps": self.eps, "lookup": lookup,
                "align_corners": bool(align_corners),
                "stop_gradients": self.stop_gradients_through_grid_sample}
        terms = _SiddonTermsFn.apply(volume, source, target, img.reshape(B, N).to(dt), gcfg)
        if mask is None:
            return _reduce_terms(terms, self.reducefn).unsqueeze(1)
        # the label of every segment: the same lookup on the label map, `.long()` (:82-84)
        labels = ops.siddon_segments_general(
            mask.to(dt), source.detach(), target.detach(), None, voxel_shift=self.voxel_shift,
            eps=self.eps, lookup=lookup, align_corners=align_corners, raw=True).long()
        return _scatter_channels(terms.transpose(1, 2), labels, mask)

    def forward(self, volume, source, target, img, align_corners=False, mask=None):
        B, N, _ = target.shape
        user_reduce = callable(self.reducefn) and not isinstance(self.reducefn, str)
        f64 = volume.dtype == torch.float64
        if self.mode == "bilinear":
            lookup = "mid_trilinear"
        else:
            lookup = "mid_nearest" if align_corners else "step"
        if mask is not None:
            # mask_to_channels (renderers.py:77-89): the reducefn plays no part
            if f64 or lookup != "step":
                return self._general(volume, source, target, img, lookup, align_corners, mask)
            cfg = self._cfg(align_corners, _grid_or_none(self, source, target), reducefn="sum")
            return _cat_channels(_chunk(_SiddonChannelsFn.apply(volume, source, target, img.reshape(B, N),
                                                                labels, C, cfg), k0)
                                 for labels, C, k0 in _labels_u8(mask))
        if user_reduce:
            # a user reduction over the per-segment tensor (renderers.py:175-183,
            # introduction.ipynb:506-529): the tensor is materialised for it
            if f64 or lookup != "step":
                return self._general(volume, source, target, img, lookup, align_corners, None)
            cfg = {"voxel_shift": self.voxel_shift, "eps": self.eps,
                   "stop_gradients": self.stop_gradients_through_grid_sample}
            terms = _SiddonSegmentsFn.apply(volume, source, target, img.reshape(B, N), cfg)
            return self.reducefn(terms).unsqueeze(1)
        cfg = self._cfg(align_corners, _grid_or_none(self, source, target))
        wants_grad = torch.is_grad_enabled() and any(
            t.requires_grad for t in (volume, source, target, img))
        if f64:
            # a module moved .to(torch.float64) (reference drr.py:71-75): the fused fp64 kernels
            # take the default lookup (reducefn sum, or max without gradients)
            if lookup != "step" or (wants_grad and self.reducefn != "sum"):
                return self._general(volume, source, target, img, lookup, align_corners, None)
            out = _SiddonF64Fn.apply(volume, source.to(volume), target.to(volume),
                                     img.reshape(B, N).to(volume), cfg)
            return out.unsqueeze(1)
        if lookup != "step" and wants_grad and (
                self.reducefn != "sum" or self.stop_gradients_through_grid_sample):
            # gradients of a midpoint lookup with reducefn="max" or under stop_gradients
            return self._general(volume, source, target, img, lookup, align_corners, None)
        out = _SiddonFn.apply(volume, source, target, img.reshape(B, N), cfg)
        return out.unsqueeze(1)


def get_alpha_minmax(source, target, dims, voxel_shift, eps):
    """First / last intersection of each ray with the (one voxel enlarged) volume,
    clipped to [0, 1]: the reference's ``_get_alpha_minmax`` (renderers.py:124-140),
    including its far plane at ``dims + 1 - voxel_shift``.  Plain torch so that
    autograd routes d/d alphamin, d/d alphamax to the arg-min / arg-max ray."""
    sdd = target - source + eps
    lo_plane = -voxel_shift
    hi_plane = dims.to(source) + 1 - voxel_shift
    a0 = (lo_plane - source) / sdd
    a1 = (hi_plane - source) / sdd
    alphamin = torch.minimum(a0, a1).amax(dim=-1, keepdim=True).clamp_min(0.0)
    alphamax = torch.maximum(a0, a1).amin(dim=-1, keepdim=True).clamp_max(1.0)
    return alphamin, alphamax


class _TrilinearFn(torch.autograd.Function):
    """out (B,N) = img * step * sum_m T(V, x(alpha_m)).  Inputs: volume, source,
    target, img, alphamin, alphamax (0-dim device tensors).  Backward: one
    ddrr_trilinear_backward launch (ray gradients, d/d alphamin, d/d alphamax and
    the 8-corner atomic scatter), replacing grid_sampler_3d_backward + chain."""

    @staticmethod
    def _grid(cfg, source, target):
        """The volume-stationary brick kernels apply: detector grid, bilinear, sum."""
        N = target.shape[1]
        return (cfg["bricks"] and cfg["mode"] == "bilinear" and cfg["reducefn"] == "sum"
                and not cfg["align_corners"] and cfg["det"] is not None
                and cfg["det"][0] * cfg["det"][1] == N and source.shape[1] == 1
                and min(cfg["det"]) >= 2)

    @staticmethod
    def forward(ctx, volume, source, target, img, alphamin, alphamax, cfg):
        aux = None
        if _TrilinearFn._grid(cfg, source, target):
            # with ray / range gradients to come, the brick kernel also leaves the backward
            # record (sum dT, sum alpha dT per ray): backward is then elementwise
            want_aux = any(ctx.needs_input_grad[1:6])
            res = ops.trilinear_forward_bricks(
                volume, source, target, img, alphamin, alphamax, cfg["det"],
                n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
                want_aux=want_aux)
            out, aux = res if want_aux else (res, None)
        else:
            out = ops.trilinear_forward(
                volume, source.contiguous(), target.contiguous(), img.contiguous(), alphamin,
                alphamax, n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"],
                eps=cfg["eps"], reducefn=cfg["reducefn"], mode=cfg["mode"],
                align_corners=cfg["align_corners"], det=cfg["det"], tile=cfg["tile"])
        ctx.cfg = cfg
        ctx.save_for_backward(volume, source, target, img, alphamin, alphamax, aux)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        volume, source, target, img, alphamin, alphamax, aux = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_s, need_t, need_i, need_a0, need_a1 = ctx.needs_input_grad[:6]
        need_rays = need_s or need_t or need_i or need_a0 or need_a1
        g_vol_bricks = None
        if need_vol and _TrilinearFn._grid(cfg, source, target):
            # volume gradient: LDS accumulation per brick instead of 8 global atomics per
            # sample; the per-ray kernel below then only produces the ray gradients
            g_vol_bricks = ops.trilinear_backward_volume_bricks(
                volume.shape, source, target, img, grad_out, alphamin, alphamax, cfg["det"],
                n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"], eps=cfg["eps"])
            need_vol = False
            if not need_rays:
                return g_vol_bricks, None, None, None, None, None, None
        if aux is not None and not need_vol:
            r = ops.trilinear_backward_rays(
                aux, grad_out, source, target, img, alphamin, alphamax, n_points=cfg["n_points"],
                eps=cfg["eps"], want_rays=bool(need_s or need_t), want_img=bool(need_i),
                want_alpha=bool(need_a0 or need_a1))
        else:
            r = ops.trilinear_backward(
                volume, source.contiguous(), target.contiguous(), img.contiguous(), grad_out,
                alphamin, alphamax, n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"],
                eps=cfg["eps"], mode=cfg["mode"], align_corners=cfg["align_corners"],
                want_rays=bool(need_s or need_t), want_img=bool(need_i),
                want_alpha=bool(need_a0 or need_a1), want_volume=bool(need_vol), det=cfg["det"],
                tile=cfg["tile"], reducefn=cfg["reducefn"])
        g_s = g_t = g_a0 = g_a1 = g_i = None
        if need_s:
            g_s = r["g_source"].sum(dim=1, keepdim=True) if source.shape[1] == 1 \
                else r["g_source"]
        if need_t:
            g_t = r["g_target"]
        if need_i:
            g_i = r["g_img"].view_as(img)
        if need_a0 or need_a1:
            ga = r["g_alpha"].sum(dim=(0, 1))
            g_a0 = ga[0].reshape(alphamin.shape) if need_a0 else None
            g_a1 = ga[1].reshape(alphamax.shape) if need_a1 else None
        return (g_vol_bricks if g_vol_bricks is not None else r["g_volume"]), g_s, g_t, g_i, \
            g_a0, g_a1, None


class _TrilinearChannelsFn(torch.autograd.Function):
    """The marcher's mask_to_channels (renderers.py:242-252): out (B,C,N); backward = every sample
    weighted by the incoming gradient of the channel its nearest label selects -- the weighted
    record on the bricks for a detector grid without a volume gradient, one
    ddrr_trilinear_backward_channels launch otherwise."""

    @staticmethod
    def forward(ctx, volume, source, target, img, alphamin, alphamax, labels, C, cfg):
        source, target, img = source.contiguous(), target.contiguous(), img.contiguous()
        B, N = target.shape[:2]
        det = cfg["det"]
        grid = (det is not None and det[0] * det[1] == N and source.shape[1] == 1
                and min(det) >= 2 and not cfg["align_corners"])
        if grid and cfg.get("bricks", True) and ops.channels_fit_bricks(B, C, N):
            # detector grid: the volume-stationary kernel (label in the staged voxel word)
            out = ops.trilinear_forward_channels_bricks(
                volume, labels, C, source, target, img, alphamin.reshape(1), alphamax.reshape(1),
                det, n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"], eps=cfg["eps"])
        else:
            out = ops.trilinear_forward_channels(
                volume, labels, C, source, target, img, alphamin.reshape(1), alphamax.reshape(1),
                n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
                align_corners=cfg["align_corners"], det=cfg["det"], tile=cfg["tile"])
        ctx.cfg = cfg
        ctx.save_for_backward(volume, source, target, img, alphamin, alphamax, labels)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        volume, source, target, img, alphamin, alphamax, labels = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_s, need_t, need_i, need_a0, need_a1 = ctx.needs_input_grad[:6]
        B, C, N = grad_out.shape
        det = cfg["det"]
        grid = (det is not None and det[0] * det[1] == N and source.shape[1] == 1
                and min(det) >= 2 and not cfg["align_corners"])
        if grid and cfg.get("bricks", True) and ops.channels_fit_bricks(B, C, N):
            # the DRR case: the weighted record on the bricks (ddrr_trilinear_backward_channels_bricks)
            # instead of the per-ray re-march, the volume gradient on the owner bricks
            r = {"g_source": None, "g_target": None, "g_img": None, "g_alpha": None, "g_volume": None}
            if need_s or need_t or need_i or need_a0 or need_a1:
                r = ops.trilinear_backward_channels_bricks(
                    volume, labels, source, target, img, grad_out, alphamin.reshape(1),
                    alphamax.reshape(1), det, n_points=cfg["n_points"],
                    voxel_shift=cfg["voxel_shift"], eps=cfg["eps"], want_rays=bool(need_s or need_t),
                    want_img=bool(need_i), want_alpha=bool(need_a0 or need_a1))
            if need_vol:
                r["g_volume"] = ops.trilinear_backward_channels_volume_bricks(
                    labels, source, target, img, grad_out, alphamin.reshape(1), alphamax.reshape(1),
                    det, n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"],
                    eps=cfg["eps"]).to(volume.dtype)
        else:
            r = ops.trilinear_backward_channels(
                volume, labels, source, target, img, grad_out, alphamin.reshape(1),
                alphamax.reshape(1), n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"],
                eps=cfg["eps"], align_corners=cfg["align_corners"],
                want_rays=bool(need_s or need_t), want_img=bool(need_i),
                want_alpha=bool(need_a0 or need_a1), want_volume=bool(need_vol), det=cfg["det"],
                tile=cfg["tile"])
        g_s = g_t = g_a0 = g_a1 = g_i = None
        if need_s:
            g_s = r["g_source"].sum(dim=1, keepdim=True) if source.shape[1] == 1 \
                else r["g_source"]
        if need_t:
            g_t = r["g_target"]
        if need_i:
            g_i = r["g_img"].view_as(img)
        if need_a0 or need_a1:
            ga = r["g_alpha"].sum(dim=(0, 1))
            g_a0 = ga[0].reshape(alphamin.shape) if need_a0 else None
            g_a1 = ga[1].reshape(alphamax.shape) if need_a1 else None
        return r["g_volume"], g_s, g_t, g_i, g_a0, g_a1, None, None, None


class _TrilinearSamplesFn(torch.autograd.Function):
    """The (B, N, P) per-sample tensor the reference hands to a callable ``reducefn`` of the
    marcher (renderers.py:226-238): ddrr_trilinear_samples / _backward."""

    @staticmethod
    def forward(ctx, volume, source, target, img, alphamin, alphamax, cfg):
        out = ops.trilinear_samples(
            volume, source, target, img, alphamin.reshape(1), alphamax.reshape(1),
            n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
            mode=cfg["mode"], align_corners=cfg["align_corners"])
        ctx.cfg = cfg
        ctx.save_for_backward(volume, source, target, img, alphamin, alphamax)
        return out.transpose(1, 2)  # (B, N, P), the reference's layout (a view)

    @staticmethod
    def backward(ctx, grad):
        volume, source, target, img, alphamin, alphamax = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_s, need_t, need_i, need_a0, need_a1 = ctx.needs_input_grad[:6]
        r = ops.trilinear_samples_backward(
            volume, source, target, img, grad.transpose(1, 2), alphamin.reshape(1),
            alphamax.reshape(1), n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"],
            eps=cfg["eps"], mode=cfg["mode"], align_corners=cfg["align_corners"],
            want_rays=bool(need_s or need_t), want_img=bool(need_i),
            want_alpha=bool(need_a0 or need_a1), want_volume=bool(need_vol))
        g_s = g_t = g_a0 = g_a1 = g_i = None
        if need_s:
            g_s = r["g_source"].sum(dim=1, keepdim=True) if source.shape[1] == 1 \
                else r["g_source"]
        if need_t:
            g_t = r["g_target"]
        if need_i:
            g_i = r["g_img"].view_as(img)
        if need_a0 or need_a1:
            ga = r["g_alpha"].sum(dim=(0, 1))
            g_a0 = ga[0].reshape(alphamin.shape) if need_a0 else None
            g_a1 = ga[1].reshape(alphamax.shape) if need_a1 else None
        return r["g_volume"], g_s, g_t, g_i, g_a0, g_a1, None


class _MarchSamplesGeneralFn(torch.autograd.Function):
    """The general path of the marcher: the (B, N, P) per-sample tensor `img * step * value`
    (renderers.py:224-236) for any mode / align_corners in float32 or float64, with its autograd
    (csrc/general_core.h)."""

    @staticmethod
    def forward(ctx, volume, source, target, img, alphamin, alphamax, cfg):
        out = ops.trilinear_samples_general(
            volume, source, target, img, alphamin, alphamax, n_points=cfg["n_points"],
            voxel_shift=cfg["voxel_shift"], eps=cfg["eps"], mode=cfg["mode"],
            align_corners=cfg["align_corners"])
        ctx.cfg = cfg
        ctx.save_for_backward(volume, source, target, img, alphamin, alphamax)
        return out.transpose(1, 2)

    @staticmethod
    def backward(ctx, grad):
        volume, source, target, img, alphamin, alphamax = ctx.saved_tensors
        cfg = ctx.cfg
        need_vol, need_s, need_t, need_i, need_a0, need_a1 = ctx.needs_input_grad[:6]
        r = ops.trilinear_samples_general_backward(
            volume, source, target, img, grad.transpose(1, 2), alphamin, alphamax,
            n_points=cfg["n_points"], voxel_shift=cfg["voxel_shift"], eps=cfg["eps"],
            mode=cfg["mode"], align_corners=cfg["align_corners"],
            want_rays=bool(need_s or need_t), want_img=bool(need_i),
            want_alpha=bool(need_a0 or need_a1), want_volume=bool(need_vol))
        g_s = g_t = g_a0 = g_a1 = g_i = None
        if need_s:
            g_s = r["g_source"].sum(dim=1, keepdim=True) if source.shape[1] == 1 \
                else r["g_source"]
        if need_t:
            g_t = r["g_target"]
        if need_i:
            g_i = r["g_img"].view_as(img)
        if need_a0 or need_a1:
            ga = r["g_alpha"].sum(dim=(0, 1))
            g_a0 = ga[0].reshape(alphamin.shape) if need_a0 else None
            g_a1 = ga[1].reshape(alphamax.shape) if need_a1 else None
        return r["g_volume"], g_s, g_t, g_i, g_a0, g_a1, None


class Trilinear(torch.nn.Module):
    """Differentiable X-ray renderer: trilinear ray marching (reference
    renderers.py:186-254) as one fused gfx950 kernel per call."""

    def __init__(self, voxel_shift: float = 0.5, mode: str = "bilinear", reducefn: str = "sum",
                 eps: float = 1e-8):
        super().__init__()
        if mode not in ("nearest", "bilinear"):
            raise ValueError(f"mode must be 'nearest' or 'bilinear', not {mode}")
        self.mode = mode
        self.reducefn = reducefn
        self.voxel_shift = voxel_shift
        self.eps = eps
        self.detector_shape = None
        self.trust_detector_shape = False  # see _grid_or_none
        self.tile = None
        self.use_bricks = True  # detector-grid calls: volume-stationary kernels (tri_brick.h)
        # mask_to_channels of a detector-grid call on the brick kernel (the label rides in the low
        # byte of the staged voxel word, the value keeps a 16-bit mantissa: as Siddon's)
        self.channels_on_bricks = True

    def dims(self, volume):
        return torch.tensor(volume.shape).to(volume)

    # the marching range is taken over the whole batch of a call unless it is passed in
    # (reference renderers.py:220-223); diffdrr_amd.dist.sweep pins it for a sharded sweep
    batch_global_range = True

    def supports_pose_entry(self):
        """Whether ``render_poses`` (the fused DRR entry) computes what ``forward`` would."""
        return self.mode == "bilinear" and self.reducefn == "sum"

    def render_poses(self, volume, Mw, P, Ainv, mask=None, n_points=500, alphamin=None,
                     alphamax=None):
        """The DRR case with the rays generated by one kernel (see ``Siddon.render_poses``);
        equals ``forward(volume, *rays(Mw, P, Ainv), n_points=n_points, mask=mask, ...)``."""
        source, target, img = _RaygenFn.apply(Mw, P, Ainv)
        return self.forward(volume, source, target, img, n_points=n_points, mask=mask,
                            alphamin=alphamin, alphamax=alphamax)

    def forward(self, volume, source, target, img, n_points=500, align_corners=False, mask=None,
                alphamin=None, alphamax=None):
        B, N, _ = target.shape
        det = _grid_or_none(self, source, target)
        user_reduce = callable(self.reducefn) and not isinstance(self.reducefn, str)
        if not user_reduce:
            ops.reduce_code(self.reducefn)
        if alphamin is None or alphamax is None:
            # the reference's batch-global marching range (renderers.py:220-223): one kernel
            # when nothing differentiates through it, tensor ops otherwise (autograd routes
            # d/d alphamin, d/d alphamax to the arg-min / arg-max ray)
            need_grad = torch.is_grad_enabled() and (source.requires_grad or target.requires_grad)
            if (not need_grad and volume.dtype == torch.float32 and B > 0 and N > 0
                    and ops.on_device(target)):
                alphamin, alphamax = ops.trilinear_alpha_range(
                    source, target, volume.shape, voxel_shift=self.voxel_shift, eps=self.eps)
            else:
                lo, hi = get_alpha_minmax(source, target, self.dims(volume), self.voxel_shift,
                                          self.eps)
                alphamin, alphamax = lo.min(), hi.max()
        alphamin = torch.as_tensor(alphamin, dtype=volume.dtype, device=volume.device)
        alphamax = torch.as_tensor(alphamax, dtype=volume.dtype, device=volume.device)
        f64 = volume.dtype == torch.float64
        scfg = {"n_points": int(n_points), "voxel_shift": self.voxel_shift, "eps": self.eps,
                "mode": self.mode, "align_corners": bool(align_corners)}

        def samples():
            """the materialised (B, N, P) tensor: the fp32 kernels that follow the reference's
            fp32 rounding, or the general path in float64"""
            if f64:
                return _MarchSamplesGeneralFn.apply(
                    volume, source.to(volume), target.to(volume), img.reshape(B, N).to(volume),
                    alphamin, alphamax, scfg)
            return _TrilinearSamplesFn.apply(volume, source, target, img.reshape(B, N), alphamin,
                                             alphamax, scfg)

        if mask is not None:
            # mask_to_channels (renderers.py:242-252): the reducefn plays no part, the labels
            # are looked up with mode "nearest" whatever the volume's mode
            if not f64 and self.mode == "bilinear":
                ccfg = {"n_points": int(n_points), "voxel_shift": self.voxel_shift,
                        "eps": self.eps, "align_corners": bool(align_corners), "det": det,
                        "tile": self.tile, "bricks": self.use_bricks and self.channels_on_bricks}
                return _cat_channels(_chunk(_TrilinearChannelsFn.apply(
                    volume, source, target, img.reshape(B, N), alphamin, alphamax, labels, C,
                    ccfg), k0) for labels, C, k0 in _labels_u8(mask))
            labels = ops.trilinear_samples_general(
                mask.to(volume), source.detach().to(volume), target.detach().to(volume), None,
                alphamin.detach(), alphamax.detach(), n_points=int(n_points),
                voxel_shift=self.voxel_shift, eps=self.eps, mode="nearest",
                align_corners=align_corners, raw=True).long()
            return _scatter_channels(samples().transpose(1, 2), labels, mask)
        if user_reduce:
            # a user reduction over the per-sample tensor (renderers.py:236-240)
            return self.reducefn(samples()).unsqueeze(1)
        if f64:
            if self.mode != "bilinear" or align_corners or self.reducefn != "sum":
                return _reduce_terms(samples(), self.reducefn).unsqueeze(1)
            fcfg = {"n_points": int(n_points), "voxel_shift": self.voxel_shift, "eps": self.eps}
            out = _TrilinearF64Fn.apply(volume, source.to(volume), target.to(volume),
                                        img.reshape(B, N).to(volume), alphamin.reshape(1),
                                        alphamax.reshape(1), fcfg)
            return out.unsqueeze(1)
        cfg = {"n_points": int(n_points), "voxel_shift": self.voxel_shift, "eps": self.eps,
               "reducefn": self.reducefn, "mode": self.mode,
               "align_corners": bool(align_corners), "det": det, "tile": self.tile,
               "bricks": self.use_bricks}
        out = _TrilinearFn.apply(volume, source, target, img.reshape(B, N), alphamin, alphamax,
                                 cfg)
        return out.unsqueeze(1)
