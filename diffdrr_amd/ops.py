"""Tensor-level entry points over the C ABI: validate, allocate, launch.

Every function takes CUDA(HIP) fp32 tensors, launches on torch's current
stream and returns without synchronising.  CPU tensors are rejected loudly --
this package renders on the MI355X only.
"""
from __future__ import annotations

import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import REDUCE_MAX, REDUCE_SUM, SIDDON_AUX

_REDUCE = {"sum": REDUCE_SUM, "max": REDUCE_MAX}
_LOOKUP = {"step": _lib.LOOKUP_STEP, "mid_nearest": _lib.LOOKUP_MID_NEAREST,
           "mid_trilinear": _lib.LOOKUP_MID_TRILINEAR}


def reduce_code(reducefn) -> int:
    if isinstance(reducefn, str) and reducefn in _REDUCE:
        return _REDUCE[reducefn]
    if callable(reducefn):
        raise NotImplementedError(
            "the fused kernels reduce with 'sum' or 'max'; a callable reducefn is served by the "
            "renderer modules through the materialised per-segment / per-sample tensors "
            "(ops.siddon_segments, ops.trilinear_samples)")
    raise ValueError(f"Only supports reducefn 'sum' or 'max', not {reducefn}")


def default_tile() -> tuple[int, int]:
    """(tile_h, tile_w) of the 64 detector pixels one wavefront renders; override
    with DDRR_TILE=HxW.  Rows map to the volume's fastest axis in the usual AP /
    lateral geometry, so the default tile is tall."""
    env = os.environ.get("DDRR_TILE")
    if env:
        h, w = (int(v) for v in env.lower().split("x"))
        if h * w != 64:
            raise ValueError("DDRR_TILE must multiply to 64")
        return h, w
    return 16, 4


def _ptr(t):
    return None if t is None else t.data_ptr()


def on_device(t) -> bool:
    """Whether `t` lives where the kernels run (the tests' host emulation patches this)."""
    return t.is_cuda


def _require_gpu(volume):
    if not volume.is_cuda:
        raise RuntimeError(
            "diffdrr_amd renders on the MI355X only: got a CPU volume tensor (there is no CPU "
            "fallback; move the module with .to('cuda'))."
        )


def _launch(name, device, *args):
    """One asynchronous C-ABI call on torch's current stream of `device`.  (The raw stream handle and the
    device switch only where the current device is another one: `torch.cuda.device(...)` +
    `torch.cuda.current_stream()` cost ~10 us of host time per call -- a third of what an eager
    registration iteration's six launches spend on the host, tools/eager_registration_loop.py.)"""
    index = device.index
    current = torch.cuda.current_device()
    if index is None or index == current:
        _lib.get_lib().call(name, *args, torch._C._cuda_getCurrentRawStream(current))
        return
    with torch.cuda.device(index):
        _lib.get_lib().call(name, *args, torch._C._cuda_getCurrentRawStream(index))


def _query(name, *args):
    """A C-ABI entry that returns a size (no launch, no stream)."""
    return _lib.get_lib().query(name, *args)


def launch_workspace(shape, device):
    """This launch's own device scratch for a ``*_bricks`` entry point (include/diffdrr_hip.h
    ``launch_ws``: the brick counter and the hand-out order of the bricks).  From torch's caching
    allocator, so it is recycled in stream order, a captured graph gets a block of its private
    pool for as long as the graph lives, and launches on different streams never share one."""
    n = int(_query("ddrr_brick_launch_workspace_bytes", *(int(d) for d in shape)))
    return torch.empty((n + 3) // 4, dtype=torch.int32, device=device)


def _check_rays(volume, source, target, img, dtype=torch.float32):
    _require_gpu(volume)
    for name, t in (("volume", volume), ("source", source), ("target", target), ("img", img)):
        if t is None:
            continue
        if t.dtype != dtype:
            raise NotImplementedError(
                f"{name} must be {dtype} here (got {t.dtype}): the tuned kernels are fp32 like the "
                "reference's default; float64 modules render through the *_f64 entry points")
        if t.device != volume.device:
            raise RuntimeError(f"{name} is on {t.device}, volume on {volume.device}")
    if volume.dim() != 3:
        raise ValueError(f"volume must be (Dx, Dy, Dz), got {tuple(volume.shape)}")
    if target.dim() != 3 or target.shape[-1] != 3:
        raise ValueError(f"target must be (B, N, 3), got {tuple(target.shape)}")
    B, N, _ = target.shape
    if source.dim() != 3 or source.shape[0] != B or source.shape[2] != 3 or \
            source.shape[1] not in (1, N):
        raise ValueError(f"source must be (B, 1, 3) or (B, N, 3), got {tuple(source.shape)}")
    if img is not None and img.numel() != B * N:
        raise ValueError(f"img must have B*N = {B * N} elements, got {tuple(img.shape)}")
    return B, N


def _empty(B, N):
    """Nothing to render (an empty batch has no valid device pointers to hand over)."""
    return B == 0 or N == 0


def rays_form_detector_grid(source, target, det_h, det_w, tol=2e-3) -> bool:
    """Whether ``target`` (B, det_h * det_w, 3) is, per pose, the row-major affine grid
    ``t00 + i e_i + j e_j`` the volume-stationary kernels assume (they cull candidate rays with
    that model; rays that do not follow it would be dropped silently), within ``tol`` voxels.
    One fused reduction and ONE host sync: meant for ray lists of unknown provenance
    (``DRR.render`` called directly); rays that come out of ``Detector.forward`` are such a grid
    by construction (reference detector.py:126, 147-153) and are not checked."""
    B, N, _ = target.shape
    if N != det_h * det_w or min(det_h, det_w) < 2 or source.shape[1] != 1:
        return False
    if target.numel() == 0:
        return False  # (an empty batch: the per-ray entry points return empty results)
    t = target.detach().reshape(B, det_h, det_w, 3)
    t00 = t[:, :1, :1]
    ei = (t[:, -1:, :1] - t00) / (det_h - 1)
    ej = (t[:, :1, -1:] - t00) / (det_w - 1)
    i = torch.arange(det_h, device=t.device, dtype=t.dtype).view(1, det_h, 1, 1)
    j = torch.arange(det_w, device=t.device, dtype=t.dtype).view(1, 1, det_w, 1)
    dev = (t00 + i * ei + j * ej - t).abs().amax()
    # a grid, not a line or a point: the pixel steps span a plane (|e_i x e_j| well above the
    # tolerance; all targets equal, or all on one line, is not a detector)
    area = torch.linalg.cross(ei.reshape(B, 3), ej.reshape(B, 3)).norm(dim=-1).amin()
    return bool(((dev <= tol) & (area > tol * tol)).item())


def _hints(det, tile, N):
    if det is None or det[0] * det[1] != N:
        return 0, 0, 1, 64
    th, tw = default_tile() if tile is None else tile
    return int(det[0]), int(det[1]), int(th), int(tw)


def siddon_forward(volume, source, target, img, *, voxel_shift=0.5, eps=1e-8, reducefn="sum",
                   lookup="step", align_corners=False, want_aux=False, count_voxels=False,
                   det=None, tile=None):
    """-> (out (B,N), aux (B,N,8) | None, n_vox (B,N) int32 | None)"""
    B, N = _check_rays(volume, source, target, img)
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    out = torch.empty(B, N, dtype=torch.float32, device=volume.device)
    aux = torch.empty(B, N, SIDDON_AUX, dtype=torch.float32, device=volume.device) \
        if want_aux else None
    nvox = torch.empty(B, N, dtype=torch.int32, device=volume.device) if count_voxels else None
    dh, dw, th, tw = _hints(det, tile, N)
    if _empty(B, N):
        return out, aux, nvox
    _launch(
        "ddrr_siddon_forward", volume.device, volume.data_ptr(), *volume.shape, source.data_ptr(),
        source.shape[1], target.data_ptr(), _ptr(img), B, N, float(voxel_shift), float(eps),
        reduce_code(reducefn), _LOOKUP[lookup], int(bool(align_corners)), dh, dw, th, tw,
        out.data_ptr(), _ptr(aux), _ptr(nvox))
    return out, aux, nvox


_vmax_cache = {}  # id(volume) -> (weakref to the volume, its version, max |volume|)


def volume_absmax(volume) -> float:
    """max |volume| as a host float, cached per volume TENSOR (weak reference + version, so a
    different volume at a recycled address is never confused with it): the scale of the
    packed backward record.  One reduction and one host sync when the volume changes."""
    ent = _vmax_cache.get(id(volume))
    if ent is not None and ent[0]() is volume and ent[1] == volume._version:
        return ent[2]
    value = float(volume.detach().abs().max().item())
    key = id(volume)
    _vmax_cache[key] = (weakref.ref(volume, lambda _, k=key: _vmax_cache.pop(k, None)),
                        volume._version, value)
    return value


def record_blocks(B, N) -> int:
    """16-ray blocks of the brick kernel's float backward record (csrc/record_layout.h)."""
    return -(-(B * N) // _lib.REC_BLOCK_RAYS)


def _aux_layout(aux, B, N):
    if aux.shape == (B, N, SIDDON_AUX):
        return _lib.AUX_INTERLEAVED
    if aux.shape == (record_blocks(B, N), _lib.REC_BLOCK_FLOATS):
        return _lib.AUX_BLOCKED
    if aux.shape == (_lib.PACKED_AUX_PLANES, B, N):
        return _lib.AUX_PACKED
    raise ValueError(f"aux has shape {tuple(aux.shape)}: neither (B,N,8), the blocked record "
                     f"(ceil(B N / 16), 80) nor the packed one (7,B,N)")


def record_planes(aux, B, N):
    """The blocked float record of :func:`siddon_forward_bricks` as (5, B, N) planes
    I, S0x, S0z, S1x, S1z (a copy; for inspection and tests)."""
    R = B * N
    blk = aux.reshape(-1, 5, 16)  # lines: [g0 p0|p1] [g0 p2|p3] [g1 p0|p1] [g1 p2|p3] [p4 x16]
    pairs = blk[:, :4].reshape(-1, 2, 2, 2, 8)        # block, group, line of group, plane in line, ray
    p03 = pairs.permute(2, 3, 0, 1, 4).reshape(4, -1)  # plane = 2 * line + plane-in-line
    return torch.cat([p03, blk[:, 4].reshape(1, -1)])[:, :R].reshape(5, B, N)


_BRICK_STORAGE = {"f32": _lib.BRICKS_F32, "q16": _lib.BRICKS_Q16, "q16p": _lib.BRICKS_Q16_PACKED}
_range_cache = {}  # (id(volume), storage) -> _Workspace


class _Workspace:
    """Cache entry of :func:`brick_workspace`: the buffer and what it was built from."""
    __slots__ = ("ref", "buf", "built_version", "built_ptr", "churn", "event", "stream")

    def __init__(self, ref, buf):
        self.ref, self.buf = ref, buf
        self.built_version = None  # volume._version the buffer holds the bricks of (None: nothing)
        self.built_ptr = None      # ... and the address of the storage they were read from
        self.churn = 0             # rebuilds because the volume had changed
        self.event = self.stream = None  # end of the building launch, and the stream it ran on


def _workspace_entry(volume, storage):
    ent = _range_cache.get((id(volume), storage))
    if ent is not None and ent.ref() is volume and ent.buf.device == volume.device:
        return ent
    return None


def brick_workspace(volume, storage="q16"):
    """Workspace of the 16-bit brick staging (include/diffdrr_hip.h DDRR_BRICKS_Q16 /
    _PACKED): a header, the bricks' (min, max) and fp32-fallback flags and, for "q16p", the
    bricks themselves as they lie in LDS (+52 % of the volume's bytes, held as long as the volume
    tensor lives).  Cached per volume TENSOR and storage.  The kernel fills it on the first
    launch after the volume changed (one pass over the volume); the launch that filled it
    reports so with :func:`brick_workspace_commit`, and only then do later calls get
    ``valid`` = 1 -- a call that returns early (empty batch) or fails leaves it unbuilt.
    "Changed" is what PyTorch itself tracks: the tensor's version counter (any in-place op,
    ``no_grad`` or not) and the address of its storage (``volume.data = other``).  In-place edits
    that bypass the version counter -- ``volume.data[...] = x``, ``volume.data.copy_(x)`` -- are
    caught ON THE DEVICE: the workspace carries a fingerprint of the volume it was built from (1024
    voxels spread over it, csrc/brick_core.h) and a launch that finds it changed renders every
    brick from the volume's own fp32 values -- slower, never stale
    (:func:`brick_workspace_stale` reports it; :func:`invalidate_brick_workspace` /
    ``Siddon.volume_changed`` rebuild).  Only an edit of a few voxels that misses all 1024 samples
    needs the explicit call.
    -> (tensor, valid)"""
    key = (id(volume), storage)
    ent = _workspace_entry(volume, storage)
    n = (int(_query("ddrr_brick_workspace_bytes", *(int(d) for d in volume.shape),
                    _BRICK_STORAGE[storage])) + 3) // 4
    if ent is None or ent.buf.numel() != n:
        # (a volume edited in place keeps its buffer: a captured graph may hold the address)
        ent = _Workspace(weakref.ref(volume, lambda _, k=key: _range_cache.pop(k, None)),
                         torch.empty(n, dtype=torch.float32, device=volume.device))
        _range_cache[key] = ent
    valid = int(ent.built_version is not None and ent.built_version == volume._version
                and ent.built_ptr == volume.data_ptr())
    if valid and ent.event is not None and not torch.cuda.is_current_stream_capturing():
        # built on another stream: this stream's launches must not overtake the build
        if ent.event.query():
            ent.event = ent.stream = None
        elif torch.cuda.current_stream(volume.device) != ent.stream:
            torch.cuda.current_stream(volume.device).wait_event(ent.event)
    return ent.buf, valid


def brick_workspace_commit(volume, storage):
    """The launch that was handed ``valid`` = 0 has been enqueued: the workspace now holds (in
    stream order) the bricks of the volume's current version."""
    ent = _workspace_entry(volume, storage)
    if ent is None or (ent.built_version == volume._version and ent.built_ptr == volume.data_ptr()):
        return
    if volume.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        # a captured launch has not run: the workspace stays unbuilt for eager launches (which
        # rebuild into the same buffer); the captured graph rebuilds it on every replay
        return
    if ent.built_version is not None:
        ent.churn += 1  # workspace_churn() lets the renderer stop paying for rebuilds
    ent.built_version = volume._version
    ent.built_ptr = volume.data_ptr()
    ent.event = ent.stream = None
    if volume.device.type == "cuda":
        ent.stream = torch.cuda.current_stream(volume.device)
        ent.event = torch.cuda.Event()
        ent.event.record(ent.stream)


def invalidate_brick_workspace(volume):
    """The volume was edited in a way PyTorch does not track (``volume.data[...] = x``): the next
    render rebuilds its cached 16-bit bricks (every storage).  Without this call such an edit is
    still rendered correctly -- the launch notices (:func:`brick_workspace`) and takes the fp32
    path for every brick -- but at the fp32 bricks' speed until the workspace is rebuilt."""
    for storage in ("q16", "q16p"):
        ent = _workspace_entry(volume, storage)
        if ent is not None:
            ent.built_version = ent.built_ptr = None


def brick_workspace_stale(volume, storage):
    """How many launches found the volume changed under this built workspace (its fingerprint did
    not match: they rendered from the fp32 values); 0 for a workspace in step with its volume, None
    if none is built.  Reads one word from the device (a host sync)."""
    ent = _workspace_entry(volume, storage)
    if ent is None or ent.built_version is None:
        return None
    return int(ent.buf[2:3].view(torch.int32).item())


def workspace_churn(volume, storage):
    """How many times the workspace of this volume was rebuilt because the volume had changed
    (0: built once): a volume that changes between renders gains nothing from a packed copy."""
    ent = _workspace_entry(volume, storage)
    return ent.churn if ent is not None else 0


def brick_fallbacks(volume, storage):
    """(bricks rendered from their own fp32 values, bricks) of the built workspace of this
    volume -- the 16-bit block quantisation is only used for bricks whose range is small against
    their level (csrc/brick_step.h q16_usable) -- or None if no launch has built it yet.
    Reads two words from the device (a host sync)."""
    ent = _workspace_entry(volume, storage)
    if ent is None or ent.built_version is None:
        return None
    head = ent.buf[:2].view(torch.int32).tolist()
    return head[0], head[1]


def brick_storage_applies(volume) -> bool:
    """Whether the 16-bit brick storages can serve this volume tensor at all: a non-contiguous
    volume is a new temporary on every call (its workspace would be rebuilt -- and its memory
    churned -- per render).  Any shape does (the reference's example CT has 133 slices: the
    staging reads quads of four voxels from dword-aligned addresses)."""
    return volume.dim() == 3 and volume.is_contiguous() and volume.numel() >= 4


def brick_ranges(volume):
    """(kept name) the "q16" workspace of :func:`brick_workspace`"""
    return brick_workspace(volume, "q16")


def brick_record_buffer(B, N, device):
    """The (uninitialised) blocked float record of a brick launch of B x N rays."""
    return torch.empty(record_blocks(B, N), _lib.REC_BLOCK_FLOATS, dtype=torch.float32, device=device)


def siddon_forward_bricks(volume, source, target, img, det, *, voxel_shift=0.5, eps=1e-8,
                          want_aux=False, record_vmax=0.0, storage="f32", want_image=True,
                          aux=None, out=None, launch_ws=None, cleared=False, pixel_mask=None):
    """Detector-grid Siddon (sum) through the volume-stationary brick kernel: every 32^3
    brick is staged in LDS once and all rays of all poses are traced through it.
    Requires the targets to be the affine detector grid DRR builds.
    -> (out (B,N), aux | None); aux is the blocked float record (ceil(B N / 16), 80)
    (csrc/record_layout.h; :func:`record_planes` unpacks it), or with ``record_vmax`` =
    max |volume| > 0 the (7,B,N) packed fixed-point record (3 atomics per ray and brick
    instead of 5, bit-reproducible; csrc/record_pack.h).
    ``storage``: "f32" stages the volume's own values in 32^3 bricks, "q16" a 16-bit block
    quantisation in 32 x 32 x 64 bricks, "q16p" the same bricks from a packed copy kept in the
    cached workspace (include/diffdrr_hip.h DDRR_BRICKS_*).
    ``aux`` / ``out`` / ``launch_ws``: a record (:func:`brick_record_buffer`) or an image (B, N) and a
    launch workspace (:func:`launch_workspace`) the caller brings along; ``cleared``: what the
    launch's atomics add to (the record if one is wanted, else the image) and the workspace's
    counter are zero already (:func:`pose_raygen_forward` did it in its launch) -- the call then
    clears nothing.  ``pixel_mask``: an int32 tensor of ceil(N / 32) words, one bit per pixel of the
    grid (:func:`pixel_mask_of`): only the pixels whose bit is set are rendered, image and record
    hold zeros at the others (``p_subsample``: reference drr.py:36-39, 142-147)."""
    B, N = _check_rays(volume, source, target, img)
    H, W = int(det[0]), int(det[1])
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    if storage != "f32" and not brick_storage_applies(volume):
        storage = "f32"
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    # (want_image = False with want_aux: the record alone -- siddon_ncc_forward forms the image)
    need_out = want_image or not want_aux
    if out is not None and (not need_out or out.shape != (B, N) or out.dtype != torch.float32
                            or not out.is_contiguous() or out.device != volume.device):
        raise ValueError("out: a contiguous float32 (B, N) image on the volume's device, where one is written")
    if need_out and out is None:
        out = torch.empty(B, N, dtype=torch.float32, device=volume.device)
    ranges, valid = brick_workspace(volume, storage) if storage != "f32" else (None, 0)
    packed = bool(want_aux and record_vmax and record_vmax > 0.0)
    if cleared and (launch_ws is None or packed or (want_aux and aux is None)):
        raise ValueError("cleared=True: the float record (an image with it is formed from the record afterwards) "
                         "or the image, and the launch workspace, as the caller cleared them")
    if want_aux and aux is None:
        shape = (_lib.PACKED_AUX_PLANES, B, N) if packed else \
            (record_blocks(B, N), _lib.REC_BLOCK_FLOATS)
        aux = torch.empty(*shape, dtype=torch.float32, device=volume.device)
    elif not want_aux:
        aux = None
    if _empty(B, N):
        return out, aux
    if launch_ws is None:
        launch_ws = launch_workspace(volume.shape, volume.device)
    if pixel_mask is not None and (pixel_mask.dtype != torch.int32 or pixel_mask.numel() != (N + 31) // 32
                                   or not pixel_mask.is_contiguous() or pixel_mask.device != volume.device):
        raise ValueError("pixel_mask: a contiguous int32 tensor of ceil(N / 32) words on the volume's device")
    _launch(
        "ddrr_siddon_forward_bricks" if pixel_mask is None else "ddrr_siddon_forward_bricks_masked",
        volume.device, volume.data_ptr(), *volume.shape,
        source.data_ptr(), target.data_ptr(), _ptr(img), B, H, W, float(voxel_shift), float(eps),
        _ptr(out), _ptr(aux), float(record_vmax) if packed else 0.0,
        _BRICK_STORAGE[storage], _ptr(ranges), int(valid) | (_lib.BRICKS_CLEARED if cleared else 0),
        launch_ws.data_ptr(), *(() if pixel_mask is None else (pixel_mask.data_ptr(),)))
    if storage != "f32" and not valid:
        brick_workspace_commit(volume, storage)
    return out, aux


def pixel_mask_of(index, n_pixels):
    """One bit per pixel of a detector grid of ``n_pixels`` pixels, set for the pixels listed in
    ``index`` (int64 tensor): the ``pixel_mask`` of :func:`siddon_forward_bricks`, int32 words on
    ``index``'s device."""
    words = torch.zeros((n_pixels + 31) // 32 * 32, dtype=torch.bool, device=index.device)
    words[index] = True
    weights = (1 << torch.arange(32, dtype=torch.int64, device=index.device))
    packed = (words.view(-1, 32).to(torch.int64) * weights).sum(dim=1)
    return (packed - ((packed >> 31) << 32)).to(torch.int32).contiguous()  # (two's complement words)


def siddon_backward_rays(aux, grad_out, source, target, img, *, eps=1e-8, reducefn="sum",
                         want_img_grad=True):
    """aux: (B,N,8) interleaved record (generic forward) or the blocked / packed record of
    the brick forward.  -> (g_source (B,N,3) per ray, g_target (B,N,3), g_img (B,N) | None)"""
    B, N, _ = target.shape
    layout = _aux_layout(aux, B, N)
    grad_out = grad_out.contiguous()
    g_source = torch.empty(B, N, 3, dtype=torch.float32, device=target.device)
    g_target = torch.empty(B, N, 3, dtype=torch.float32, device=target.device)
    g_img = torch.empty(B, N, dtype=torch.float32, device=target.device) if want_img_grad else None
    if _empty(B, N):
        return g_source, g_target, g_img
    _launch(
        "ddrr_siddon_backward_rays", target.device, aux.data_ptr(), layout, grad_out.data_ptr(), source.data_ptr(),
        source.shape[1], target.data_ptr(), _ptr(img), B, N, float(eps),
        reduce_code(reducefn), g_source.data_ptr(), g_target.data_ptr(), _ptr(g_img))
    return g_source, g_target, g_img


def pose_euler_forward(rot, xyz, axes, reorient34):
    """rot, xyz (B,3) radians / world units; axes: 3 ints in {0,1,2}; reorient34 (3,4).
    -> Mw (B,3,4) = [R | R xyz] @ reorient"""
    _require_gpu(rot)
    B = rot.shape[0]
    rot, xyz, reorient34 = rot.contiguous(), xyz.contiguous(), reorient34.contiguous()
    Mw = torch.empty(B, 3, 4, dtype=torch.float32, device=rot.device)
    if B:
        _launch("ddrr_pose_euler_forward", rot.device, rot.data_ptr(), xyz.data_ptr(), *axes,
                reorient34.data_ptr(), B, Mw.data_ptr())
    return Mw


def pose_euler_backward(rot, xyz, axes, reorient34, gMw):
    B = rot.shape[0]
    rot, xyz, reorient34, gMw = (t.contiguous() for t in (rot, xyz, reorient34, gMw))
    g_rot, g_xyz = torch.empty_like(rot), torch.empty_like(xyz)
    if B:
        _launch("ddrr_pose_euler_backward", rot.device, rot.data_ptr(), xyz.data_ptr(), *axes,
                reorient34.data_ptr(), gMw.data_ptr(), B, g_rot.data_ptr(), g_xyz.data_ptr())
    return g_rot, g_xyz


def ncc_forward(x1, x2, eps):
    """x2 (B,N); x1 (B,N) or (1,N) shared by the batch.  -> (ncc (B), stats (B,5))"""
    _require_gpu(x2)
    B, N = x2.shape
    shared = x1.shape[0] == 1 and B != 1
    x1, x2 = x1.contiguous(), x2.contiguous()
    out = torch.empty(B, dtype=torch.float32, device=x2.device)
    stats = torch.empty(B, 5, dtype=torch.float32, device=x2.device)
    if B:
        _launch("ddrr_ncc_forward", x2.device, x1.data_ptr(), 0 if shared else N, x2.data_ptr(), B,
                N, float(eps), out.data_ptr(), stats.data_ptr())
    return out, stats


def ncc_backward(x1, x2, stats, g_out, want_x1, want_x2):
    B, N = x2.shape
    shared = x1.shape[0] == 1 and B != 1
    x1, x2 = x1.contiguous(), x2.contiguous()
    # (the gradient of `.sum()` / `.mean()` arrives as an expanded scalar: read in place, stride 0)
    # (one value for the whole batch: the gradient of a summed objective -- a 0-dim tensor, or what
    # autograd expands it to)
    g_stride = 0 if (g_out.dim() == 0 or (g_out.dim() == 1 and B > 1 and g_out.stride(0) == 0)) else 1
    if g_stride:
        g_out = g_out.contiguous()
    g_x2 = torch.empty_like(x2) if want_x2 else None
    g_x1 = torch.empty_like(x2) if (want_x1 and not shared) else None
    if B:
        _launch("ddrr_ncc_backward", x2.device, x1.data_ptr(), 0 if shared else N, x2.data_ptr(),
                stats.data_ptr(), g_out.data_ptr(), g_stride, B, N, _ptr(g_x1), _ptr(g_x2))
    return g_x1, g_x2


def ncc_patch_forward(x1, x2, p, eps, want_coef=True):
    """Patch-wise NCC (reference metrics.py:16-44, ``patch_size = p``) of image pairs: x2 (B, H, W);
    x1 (B, H, W) or (1, H, W) shared by the batch.  -> (ncc (B), coef (B, H-p+1, W-p+1, 4) | None: what
    :func:`ncc_patch_backward` needs)"""
    _require_gpu(x2)
    B, H, W = x2.shape
    shared = x1.shape[0] == 1 and B != 1
    x1, x2 = x1.contiguous(), x2.contiguous()
    out = torch.empty(B, dtype=torch.float32, device=x2.device)
    coef = torch.empty(B, H - p + 1, W - p + 1, 4, dtype=torch.float32, device=x2.device) if want_coef else None
    if B:
        _launch("ddrr_ncc_patch_forward", x2.device, x1.data_ptr(), 0 if shared else H * W, x2.data_ptr(), B,
                H, W, int(p), float(eps), out.data_ptr(), _ptr(coef))
    return out, coef


def ncc_patch_backward(x1, x2, coef, g_out, p):
    """d (sum_b g_out[b] ncc[b]) / d x2 (B, H, W) of :func:`ncc_patch_forward`."""
    B, H, W = x2.shape
    shared = x1.shape[0] == 1 and B != 1
    x1, x2 = x1.contiguous(), x2.contiguous()
    g_stride = 0 if (g_out.dim() == 0 or (g_out.dim() == 1 and B > 1 and g_out.stride(0) == 0)) else 1
    if g_stride:
        g_out = g_out.contiguous()
    g_x2 = torch.empty_like(x2)
    if B:
        _launch("ddrr_ncc_patch_backward", x2.device, x1.data_ptr(), 0 if shared else H * W, x2.data_ptr(),
                coef.data_ptr(), g_out.data_ptr(), g_stride, B, H, W, int(p), g_x2.data_ptr())
    return g_x2


def sobel_forward(img):
    """img (B, H, W) -> (B, 2, H, W): the Sobel x / y responses (reference metrics.py:69-94)."""
    _require_gpu(img)
    B, H, W = img.shape
    img = img.contiguous()
    out = torch.empty(B, 2, H, W, dtype=torch.float32, device=img.device)
    if B:
        _launch("ddrr_sobel_forward", img.device, img.data_ptr(), B, H, W, out.data_ptr())
    return out


def sobel_backward(g_out):
    """Adjoint of :func:`sobel_forward`: g_out (B, 2, H, W) -> (B, H, W)."""
    B, _, H, W = g_out.shape
    g_out = g_out.contiguous()
    g_img = torch.empty(B, H, W, dtype=torch.float32, device=g_out.device)
    if B:
        _launch("ddrr_sobel_backward", g_out.device, g_out.data_ptr(), B, H, W, g_img.data_ptr())
    return g_img


def blur_sobel_forward(img, taps):
    """img (B, H, W) -- or one image expanded over the batch -- -> (B, 2, H, W): the Sobel responses of the
    Gaussian-blurred image (reference metrics.py:66, 88-93) in one launch; taps: (k) on the device."""
    _require_gpu(img)
    B, H, W = img.shape
    shared = B > 1 and img.stride(0) == 0
    if shared:
        img = img[:1]
    img = img.contiguous()
    out = torch.empty(B, 2, H, W, dtype=torch.float32, device=img.device)
    if B:
        _launch("ddrr_blur_sobel_forward", img.device, img.data_ptr(), 0 if shared else H * W, B, H, W,
                taps.data_ptr(), int(taps.numel()), out.data_ptr())
    return out


def blur_sobel_backward(g_out, taps):
    """Adjoint of :func:`blur_sobel_forward`: g_out (B, 2, H, W) -> (B, H, W)."""
    B, _, H, W = g_out.shape
    g_out = g_out.contiguous()
    g_img = torch.empty(B, H, W, dtype=torch.float32, device=g_out.device)
    if B:
        _launch("ddrr_blur_sobel_backward", g_out.device, g_out.data_ptr(), B, H, W, taps.data_ptr(),
                int(taps.numel()), g_img.data_ptr())
    return g_img


def raygen_forward(Mw, Ainv, P):
    """Fused ray generation (detector.py:151-153 + drr.py:201-205).  Mw (B,3,4) world pose
    per DRR, Ainv (3,4) world -> voxel, P (N,3) calibrated detector points.
    -> (source_v (B,1,3), target_v (B,N,3), img (B,N))"""
    _require_gpu(Mw)
    B, N = Mw.shape[0], P.shape[0]
    if Mw.shape[1:] != (3, 4) or Ainv.shape != (3, 4) or P.shape != (N, 3):
        raise ValueError("raygen_forward: Mw (B,3,4), Ainv (3,4), P (N,3) expected")
    for t in (Mw, Ainv, P):
        if t.dtype != torch.float32 or t.device != Mw.device:
            raise NotImplementedError("raygen_forward needs float32 tensors on one device")
    Mw, Ainv, P = Mw.contiguous(), Ainv.contiguous(), P.contiguous()
    dev = Mw.device
    source = torch.empty(B, 1, 3, dtype=torch.float32, device=dev)
    target = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
    img = torch.empty(B, N, dtype=torch.float32, device=dev)
    if not _empty(B, N):
        _launch("ddrr_raygen_forward", dev, Mw.data_ptr(), Ainv.data_ptr(), P.data_ptr(), B, N,
                source.data_ptr(), target.data_ptr(), img.data_ptr())
    elif B > 0:
        source.zero_()
    return source, target, img


def siddon_backward_pose(aux, grad_out, source, target, img, Mw, Ainv, P, *, eps=1e-8,
                         with_img_path=True):
    """dLoss/dMw (B,3,4): the renderer's ray gradients chained through the ray generation
    and reduced per pose in one kernel (reduce sum)."""
    B, N, _ = target.shape
    layout = _aux_layout(aux, B, N)
    gMw = torch.empty(B, 3, 4, dtype=torch.float32, device=target.device)  # (zero-filled by the call)
    if B == 0:
        return gMw
    # (named, so that a contiguous copy outlives the launch)
    grad_out, source, target, img = (t.contiguous() for t in (grad_out, source, target, img))
    Mw, Ainv, P = Mw.contiguous(), Ainv.contiguous(), P.contiguous()
    _launch("ddrr_siddon_backward_pose", target.device, aux.data_ptr(), layout,
            grad_out.data_ptr(), source.data_ptr(), target.data_ptr(), img.data_ptr(),
            Mw.data_ptr(), Ainv.data_ptr(), P.data_ptr(), B, N, float(eps),
            int(bool(with_img_path)), gMw.data_ptr())
    return gMw


# ---------------------------------------------------------------- the fused registration step
_ncc_ws = {}  # (device, stream, B) -> workspace of ddrr_siddon_ncc_* (zero between calls)


def siddon_ncc_workspace(B, device):
    """The caller-owned accumulators / tickets of :func:`siddon_ncc_forward` and
    :func:`siddon_ncc_backward_pose`: zero when first handed over, left zero by every call, so one
    buffer per (device, stream, batch size) serves every call on that stream without a fill.
    (First use inside a stream capture: the zero-fill becomes a node of the graph -- harmless, one
    small launch per replay; ``GraphedIteration`` creates the buffer for its capture stream before
    it captures.)"""
    dev = torch.device(device)
    sid = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    key = (dev, sid, int(B))
    ws = _ncc_ws.get(key)
    if ws is None:
        n = int(_query("ddrr_siddon_ncc_workspace_bytes", int(B)))
        ws = _ncc_ws[key] = torch.zeros((n + 7) // 8, dtype=torch.float64, device=dev)
    return ws


def pose_raygen_forward(rot, xyz, axes, reorient34, Ainv, P, *, clear=None, clear_launch_ws=None):
    """pose_euler_forward + raygen_forward in one launch -> (Mw (B,3,4), source_v (B,1,3),
    target_v (B,N,3), img (B,N)).  ``clear`` (a float32 tensor) and ``clear_launch_ws`` (the launch
    workspace of the render that follows) are zeroed by the same launch: hand them to
    :func:`siddon_forward_bricks` with ``cleared=True``."""
    _require_gpu(rot)
    B, N = rot.shape[0], P.shape[0]
    rot, xyz, reorient34, Ainv, P = (t.contiguous() for t in (rot, xyz, reorient34, Ainv, P))
    dev = rot.device
    Mw = torch.empty(B, 3, 4, dtype=torch.float32, device=dev)
    source = torch.empty(B, 1, 3, dtype=torch.float32, device=dev)
    target = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
    img = torch.empty(B, N, dtype=torch.float32, device=dev)
    if not _empty(B, N):
        _launch("ddrr_pose_raygen_forward", dev, rot.data_ptr(), xyz.data_ptr(), *axes,
                reorient34.data_ptr(), Ainv.data_ptr(), P.data_ptr(), B, N, Mw.data_ptr(),
                source.data_ptr(), target.data_ptr(), img.data_ptr(), _ptr(clear),
                0 if clear is None else clear.numel(), _ptr(clear_launch_ws))
    elif clear is not None or clear_launch_ws is not None:
        raise ValueError("pose_raygen_forward: nothing is launched for an empty batch, nothing is cleared")
    return Mw, source, target, img


def siddon_ncc_forward(aux, img, x1, eps, *, want_image=False, want_sum=False):
    """Per-pose NCC of the fixed image(s) ``x1`` ((B | 1), N) with the DRRs whose blocked record
    is ``aux`` (``img`` (B,N): the rays' lengths): the image is formed from the record on the fly.
    -> (ncc (B), stats (B,5), image (B,N) | None[, sum of the B values (0-dim), with ``want_sum``:
    put together by the same launch])"""
    B, N = img.shape
    shared = x1.shape[0] == 1 and B != 1
    x1, img = x1.contiguous(), img.contiguous()
    dev = img.device
    ncc = torch.empty(B, dtype=torch.float32, device=dev)
    stats = torch.empty(B, 5, dtype=torch.float32, device=dev)
    out = torch.empty(B, N, dtype=torch.float32, device=dev) if want_image else None
    total = (torch.empty((), dtype=torch.float32, device=dev) if B else
             torch.zeros((), dtype=torch.float32, device=dev)) if want_sum else None
    if B:
        _launch("ddrr_siddon_ncc_forward", dev, aux.data_ptr(), img.data_ptr(), x1.data_ptr(),
                0 if shared else N, B, N, float(eps), siddon_ncc_workspace(B, dev).data_ptr(),
                ncc.data_ptr(), stats.data_ptr(), _ptr(out), _ptr(total))
    return (ncc, stats, out, total) if want_sum else (ncc, stats, out)


def pose_adam_step(rot, xyz, g_rot, g_xyz, m_rot, v_rot, m_xyz, v_xyz, step_rot, step_xyz, *, lr_rot, lr_xyz,
                   betas=(0.9, 0.999), eps=1e-8, maximize=False):
    """One Adam step of the two pose parameter groups, in place, in one launch (include/diffdrr_hip.h
    ddrr_pose_adam_step: torch.optim.Adam's update; `step_*` are 1-element float tensors)."""
    _require_gpu(rot)
    B = rot.shape[0]
    for t in (rot, xyz, g_rot, g_xyz, m_rot, v_rot, m_xyz, v_xyz):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != (B, 3):
            raise ValueError("pose_adam_step: contiguous float32 (B, 3) tensors")
    for t in (step_rot, step_xyz):
        if t.dtype != torch.float32 or t.numel() != 1:
            raise ValueError("pose_adam_step: 1-element float32 step counters")
    if B:
        _launch("ddrr_pose_adam_step", rot.device, rot.data_ptr(), xyz.data_ptr(), g_rot.data_ptr(),
                g_xyz.data_ptr(), m_rot.data_ptr(), v_rot.data_ptr(), m_xyz.data_ptr(), v_xyz.data_ptr(),
                step_rot.data_ptr(), step_xyz.data_ptr(), B, float(lr_rot), float(lr_xyz), float(betas[0]),
                float(betas[1]), float(eps), int(bool(maximize)))


def siddon_ncc_backward_pose(aux, img, x1, stats, g_out, source, target, Mw, Ainv, P, rot, xyz, axes,
                             reorient34, *, eps=1e-8, with_img_path=True):
    """(g_rot (B,3), g_xyz (B,3)) of sum_b g_out[b] ncc[b]: ncc_backward, siddon_backward_pose and
    pose_euler_backward in one launch."""
    B, N = img.shape
    shared = x1.shape[0] == 1 and B != 1
    # (one value for the whole batch: the gradient of a summed objective -- a 0-dim tensor, or what
    # autograd expands it to)
    g_stride = 0 if (g_out.dim() == 0 or (g_out.dim() == 1 and B > 1 and g_out.stride(0) == 0)) else 1
    if g_stride:
        g_out = g_out.contiguous()
    x1, img, source, target = (t.contiguous() for t in (x1, img, source, target))
    Mw, Ainv, P, rot, xyz, reorient34 = (t.contiguous() for t in (Mw, Ainv, P, rot, xyz, reorient34))
    dev = img.device
    g_rot = torch.empty(B, 3, dtype=torch.float32, device=dev)
    g_xyz = torch.empty(B, 3, dtype=torch.float32, device=dev)
    if B:
        _launch("ddrr_siddon_ncc_backward_pose", dev, aux.data_ptr(), img.data_ptr(), x1.data_ptr(),
                0 if shared else N, stats.data_ptr(), g_out.data_ptr(), g_stride, source.data_ptr(),
                target.data_ptr(), Mw.data_ptr(), Ainv.data_ptr(), P.data_ptr(), rot.data_ptr(),
                xyz.data_ptr(), *axes, reorient34.data_ptr(), B, N, float(eps),
                int(bool(with_img_path)), siddon_ncc_workspace(B, dev).data_ptr(), g_rot.data_ptr(),
                g_xyz.data_ptr())
    return g_rot, g_xyz


def siddon_backward_pose_euler(aux, grad_out, source, Mw, Ainv, P, rot, xyz, axes, reorient34, *, eps=1e-8,
                               with_img_path=True):
    """(g_rot (B,3), g_xyz (B,3)) of any objective of the image with per-pixel gradient ``grad_out`` (B, N):
    siddon_backward_pose and pose_euler_backward in one launch (the rays as pose_raygen_forward made them)."""
    B, N = grad_out.shape
    grad_out, source = grad_out.contiguous(), source.contiguous()
    Mw, Ainv, P, rot, xyz, reorient34 = (t.contiguous() for t in (Mw, Ainv, P, rot, xyz, reorient34))
    dev = grad_out.device
    g_rot = torch.empty(B, 3, dtype=torch.float32, device=dev)
    g_xyz = torch.empty(B, 3, dtype=torch.float32, device=dev)
    if B:
        _launch("ddrr_siddon_backward_pose_euler", dev, aux.data_ptr(), grad_out.data_ptr(), source.data_ptr(),
                Mw.data_ptr(), Ainv.data_ptr(), P.data_ptr(), rot.data_ptr(), xyz.data_ptr(), *axes,
                reorient34.data_ptr(), B, N, float(eps), int(bool(with_img_path)),
                siddon_ncc_workspace(B, dev).data_ptr(), g_rot.data_ptr(), g_xyz.data_ptr())
    return g_rot, g_xyz


def siddon_backward_volume(volume, source, target, img, grad_out, *, voxel_shift=0.5, eps=1e-8,
                           reducefn="sum", det=None, tile=None):
    B, N = _check_rays(volume, source, target, img)
    g_volume = torch.zeros_like(volume, memory_format=torch.contiguous_format)
    dh, dw, th, tw = _hints(det, tile, N)
    if _empty(B, N):
        return g_volume
    volume, source, target, grad_out = (t.contiguous() for t in (volume, source, target, grad_out))
    img = None if img is None else img.contiguous()
    _launch(
        "ddrr_siddon_backward_volume", volume.device, volume.data_ptr(), *volume.shape, source.data_ptr(),
        source.shape[1], target.data_ptr(), _ptr(img), grad_out.data_ptr(), B, N,
        float(voxel_shift), float(eps), reduce_code(reducefn), dh, dw, th, tw,
        g_volume.data_ptr())
    return g_volume


def siddon_backward_volume_bricks(volume_shape, source, target, img, grad_out, det, *,
                                  voxel_shift=0.5, eps=1e-8):
    """Detector-grid Siddon (sum) volume gradient through the volume-stationary brick
    kernel: every 32^3 brick of the gradient is accumulated in LDS and stored once.
    -> g_volume (Dx,Dy,Dz), fully written"""
    B, N, _ = target.shape
    H, W = int(det[0]), int(det[1])
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    _require_gpu(target)
    source, target, grad_out = source.contiguous(), target.contiguous(), grad_out.contiguous()
    img = None if img is None else img.contiguous()
    Dx, Dy, Dz = (int(v) for v in volume_shape)
    g_volume = torch.empty(Dx, Dy, Dz, dtype=torch.float32, device=target.device)
    _launch("ddrr_siddon_backward_volume_bricks", target.device, Dx, Dy, Dz, source.data_ptr(),
            target.data_ptr(), _ptr(img), grad_out.data_ptr(), B, H, W, float(voxel_shift),
            float(eps), g_volume.data_ptr(),
            launch_workspace((Dx, Dy, Dz), target.device).data_ptr())
    return g_volume


def siddon_forward_channels(volume, labels_u8, n_channels, source, target, img, *,
                            voxel_shift=0.5, eps=1e-8, det=None, tile=None):
    """-> (B, C, N)"""
    B, N = _check_rays(volume, source, target, img)
    if labels_u8.dtype != torch.uint8 or labels_u8.shape != volume.shape:
        raise ValueError("labels must be a uint8 tensor of the volume's shape")
    out = torch.empty(B, n_channels, N, dtype=torch.float32, device=volume.device)
    dh, dw, th, tw = _hints(det, tile, N)
    if _empty(B, N):
        return out
    labels_u8, volume = labels_u8.contiguous(), volume.contiguous()
    _launch(
        "ddrr_siddon_forward_channels", volume.device, volume.data_ptr(), labels_u8.data_ptr(),
        *volume.shape, source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img), B, N,
        int(n_channels), float(voxel_shift), float(eps), dh, dw, th, tw, out.data_ptr())
    return out


_words_cache = {}  # (id(volume), id(labels), C) -> _ChannelWords


class _ChannelWords:
    """Cache entry of :func:`channel_words`."""
    __slots__ = ("vol", "lab", "tracked", "words", "state", "seen", "churn")

    def __init__(self, vol, lab):
        self.vol, self.lab = vol, lab   # weak references
        self.tracked = None             # what PyTorch tracks of the pair when the words were last packed
        self.words = self.state = None
        self.seen = 0                   # renders of this pair in this tracked state
        self.churn = 0                  # repacks in a row because the pair had changed between two renders


def channel_words(volume, labels_u8, n_channels, build=True):
    """The channel render's staged words (value with a 16-bit mantissa | label) for the whole volume,
    one float per voxel, for a (volume, label map) pair that is rendered again and again:
    :func:`siddon_forward_channels_bricks` then stages a brick with straight 16-byte copies -- no label
    loads, no packing (one pose 0.089 -> 0.079 ms, 8 poses 0.280 -> 0.262 on the reference's example
    shape and label map, the comparison launches included).  Cached per (volume tensor, label tensor, channels) while both live;
    +100 % of the volume's bytes.  EVERY call launches ``ddrr_channel_words``: it compares a
    fingerprint of volume and labels on the device (1024 voxels each) and returns after a few
    microseconds if nothing changed; a change PyTorch tracks (version counters, storage addresses)
    forces the repack, one it does not (``volume.data[...] = x``) is found by the fingerprint.
    ``build=False``: None unless the pair has been seen before in its present state (the first render
    of a pair does not pay for a pass it may never use)."""
    key = (id(volume), id(labels_u8), int(n_channels))
    tracked = (volume._version, labels_u8._version, volume.data_ptr(), labels_u8.data_ptr())
    ent = _words_cache.get(key)
    if ent is None or ent.vol() is not volume or ent.lab() is not labels_u8:
        drop = lambda _, k=key: _words_cache.pop(k, None)  # noqa: E731
        ent = _words_cache[key] = _ChannelWords(weakref.ref(volume, drop), weakref.ref(labels_u8, drop))
    if not build:
        if ent.words is None:
            ent.seen = ent.seen + 1 if ent.tracked == tracked else 1
            ent.tracked = tracked
            if ent.seen < 2:
                return None
    force = ent.words is None or ent.tracked != tracked
    if not build:
        # a pair that changes between renders again and again (a volume edited in place every
        # iteration) gains nothing from words that are packed anew for every render: given up for it
        ent.churn = ent.churn + 1 if (force and ent.words is not None) else 0
        if ent.churn >= 3:
            ent.words = ent.state = None
            ent.tracked, ent.seen, ent.churn = tracked, -(1 << 30), 0
            return None
    if ent.words is None:
        ent.words = torch.empty_like(volume)
        n = int(_query("ddrr_channel_words_state_bytes"))
        ent.state = torch.zeros((n + 3) // 4, dtype=torch.int32, device=volume.device)
    ent.tracked = tracked
    if volume.numel():
        _launch("ddrr_channel_words", volume.device, volume.data_ptr(), labels_u8.data_ptr(), volume.numel(),
                int(n_channels), ent.words.data_ptr(), ent.state.data_ptr(), int(force))
    return ent.words


def channel_words_repacks(volume, labels_u8, n_channels):
    """How often the words of this pair were packed (1 after the first build; +1 for every change found
    by the tracked state or by the device-side fingerprint), or None.  Reads a word from the device."""
    ent = _words_cache.get((id(volume), id(labels_u8), int(n_channels)))
    if ent is None or ent.state is None or ent.vol() is not volume:
        return None
    return int(ent.state[1].item())


def siddon_forward_channels_bricks(volume, labels_u8, n_channels, source, target, img, det, *,
                                   voxel_shift=0.5, eps=1e-8, words=None):
    """:func:`siddon_forward_channels` for a detector grid on the volume-stationary brick
    kernel (the label rides in the low byte of the staged voxel word).  ``words``: the volume's
    ready-packed words (:func:`channel_words`) -- staged as they are.  -> (B, C, N)"""
    B, N = _check_rays(volume, source, target, img)
    H, W = int(det[0]), int(det[1])
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    if labels_u8.dtype != torch.uint8 or labels_u8.shape != volume.shape:
        raise ValueError("labels must be a uint8 tensor of the volume's shape")
    out = torch.empty(B, n_channels, N, dtype=torch.float32, device=volume.device)
    if _empty(B, N):
        return out
    labels_u8, volume = labels_u8.contiguous(), volume.contiguous()
    source, target = source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    if words is not None:
        if words.shape != volume.shape or words.dtype != torch.float32 or not words.is_contiguous():
            raise ValueError("words: channel_words(volume, labels, n_channels)")
        _launch("ddrr_siddon_forward_channels_bricks_words", volume.device, words.data_ptr(), *volume.shape,
                source.data_ptr(), target.data_ptr(), _ptr(img), B, H, W, int(n_channels), float(voxel_shift),
                float(eps), out.data_ptr(), launch_workspace(volume.shape, volume.device).data_ptr())
        return out
    _launch("ddrr_siddon_forward_channels_bricks", volume.device, volume.data_ptr(),
            labels_u8.data_ptr(), *volume.shape, source.data_ptr(), target.data_ptr(), _ptr(img),
            B, H, W, int(n_channels), float(voxel_shift), float(eps), out.data_ptr(),
            launch_workspace(volume.shape, volume.device).data_ptr())
    return out


def siddon_backward_channels_bricks(volume, labels_u8, source, target, img, grad_out, det, *,
                                    voxel_shift=0.5, eps=1e-8, want_img=True):
    """Ray / img gradients of :func:`siddon_forward_channels_bricks` for grad_out (B, C, N) on the
    volume-stationary bricks: the brick kernel writes the backward record of the volume weighted by
    every voxel's own incoming gradient, ``ddrr_siddon_backward_rays`` turns it into gradients.
    -> (g_source per ray (B,N,3), g_target (B,N,3), g_img (B,N) | None)"""
    B, N = _check_rays(volume, source, target, img)
    H, W = int(det[0]), int(det[1])
    C = grad_out.shape[1]
    if grad_out.shape != (B, C, N):
        raise ValueError(f"grad_out must be (B, C, N) = ({B}, C, {N}), got {tuple(grad_out.shape)}")
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    dev = volume.device
    aux = torch.empty(record_blocks(B, N), _lib.REC_BLOCK_FLOATS, dtype=torch.float32, device=dev)
    ones = torch.ones(B, N, dtype=torch.float32, device=dev)
    if not _empty(B, N):
        labels_u8, volume, grad_out = labels_u8.contiguous(), volume.contiguous(), grad_out.contiguous()
        source, target = source.contiguous(), target.contiguous()
        _launch("ddrr_siddon_backward_channels_bricks", dev, volume.data_ptr(), labels_u8.data_ptr(),
                *volume.shape, source.data_ptr(), target.data_ptr(), grad_out.data_ptr(), B, H, W,
                int(C), float(voxel_shift), float(eps), aux.data_ptr(),
                launch_workspace(volume.shape, dev).data_ptr())
    return siddon_backward_rays(aux, ones, source, target, img, eps=eps, want_img_grad=want_img)


def siddon_backward_channels_volume_bricks(labels_u8, source, target, img, grad_out, det, *,
                                           voxel_shift=0.5, eps=1e-8):
    """Volume gradient of :func:`siddon_forward_channels_bricks` for grad_out (B, C, N) on the
    volume-stationary bricks (the LDS brick is the accumulator, the voxel's label rides in the
    word's low byte).  -> g_volume, the label map's shape"""
    B, N, _ = target.shape
    H, W = int(det[0]), int(det[1])
    C = grad_out.shape[1]
    if grad_out.shape != (B, C, N):
        raise ValueError(f"grad_out must be (B, C, N) = ({B}, C, {N}), got {tuple(grad_out.shape)}")
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    if labels_u8.dtype != torch.uint8 or labels_u8.dim() != 3:
        raise ValueError("labels must be a 3-D uint8 tensor")
    _require_gpu(target)
    dev = target.device
    labels_u8, grad_out = labels_u8.contiguous(), grad_out.contiguous()
    source, target = source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    g_volume = torch.empty(labels_u8.shape, dtype=torch.float32, device=dev)
    _launch("ddrr_siddon_backward_channels_volume_bricks", dev, labels_u8.data_ptr(),
            *labels_u8.shape, source.data_ptr(), target.data_ptr(), _ptr(img), grad_out.data_ptr(), B,
            H, W, int(C), float(voxel_shift), float(eps), g_volume.data_ptr(),
            launch_workspace(labels_u8.shape, dev).data_ptr())
    return g_volume


def channels_fit_bricks(B, C, N):
    """One brick launch addresses the (B, C, N) result with 32-bit byte offsets."""
    return B * C * N < 2 ** 30 and N < 2 ** 22


def siddon_backward_midpoint(volume, source, target, img, grad_out, *, voxel_shift=0.5, eps=1e-8,
                             lookup="mid_nearest", align_corners=False, want_rays=True,
                             want_img=True, want_volume=False):
    """Backward of :func:`siddon_forward` for the midpoint lookups (reduce sum).
    -> (g_source per ray, g_target, g_img, g_volume), None where not asked."""
    B, N = _check_rays(volume, source, target, img)
    dev = volume.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    g_source = new(B, N, 3) if want_rays else None
    g_target = new(B, N, 3) if want_rays else None
    g_img = new(B, N) if want_img else None
    g_volume = torch.zeros_like(volume, memory_format=torch.contiguous_format) \
        if want_volume else None
    if _empty(B, N):
        return g_source, g_target, g_img, g_volume
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    grad_out = grad_out.contiguous()
    _launch("ddrr_siddon_backward_midpoint", dev, volume.data_ptr(), *volume.shape,
            source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img), grad_out.data_ptr(),
            B, N, float(voxel_shift), float(eps), _LOOKUP[lookup], int(bool(align_corners)),
            _ptr(g_source), _ptr(g_target), _ptr(g_img), _ptr(g_volume))
    return g_source, g_target, g_img, g_volume


def siddon_segments(volume, source, target, img, *, voxel_shift=0.5, eps=1e-8):
    """The per-segment terms a callable ``reducefn`` receives (renderers.py:70-71).
    -> (B, M-1, N) with M = Dx+Dy+Dz+3; transpose(1, 2) is the reference's layout."""
    B, N = _check_rays(volume, source, target, img)
    M1 = sum(int(v) for v in volume.shape) + 2
    terms = torch.empty(B, M1, N, dtype=torch.float32, device=volume.device)
    if _empty(B, N):
        return terms
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    _launch("ddrr_siddon_segments", volume.device, volume.data_ptr(), *volume.shape,
            source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img), B, N,
            float(voxel_shift), float(eps), terms.data_ptr())
    return terms


def siddon_segments_backward(volume, source, target, img, grad_terms, *, voxel_shift=0.5, eps=1e-8,
                             want_rays=True, want_img=True, want_volume=False):
    """Backward of :func:`siddon_segments` for grad_terms (B, M-1, N).
    -> (g_source per ray, g_target, g_img, g_volume), None where not asked."""
    B, N = _check_rays(volume, source, target, img)
    dev = volume.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    g_source = new(B, N, 3) if want_rays else None
    g_target = new(B, N, 3) if want_rays else None
    g_img = new(B, N) if want_img else None
    g_volume = torch.zeros_like(volume, memory_format=torch.contiguous_format) \
        if want_volume else None
    if _empty(B, N):
        return g_source, g_target, g_img, g_volume
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    grad_terms = grad_terms.contiguous()
    _launch("ddrr_siddon_segments_backward", dev, volume.data_ptr(), *volume.shape,
            source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img),
            grad_terms.data_ptr(), B, N, float(voxel_shift), float(eps), _ptr(g_source),
            _ptr(g_target), _ptr(g_img), _ptr(g_volume))
    return g_source, g_target, g_img, g_volume


def siddon_backward_channels(volume, labels_u8, source, target, img, grad_out, *, voxel_shift=0.5,
                             eps=1e-8, want_rays=True, want_img=True, want_volume=False,
                             det=None, tile=None):
    """Backward of :func:`siddon_forward_channels` for grad_out (B, C, N).
    -> (g_source per ray (B,N,3), g_target (B,N,3), g_img (B,N), g_volume), None where not asked."""
    B, N = _check_rays(volume, source, target, img)
    dev = volume.device
    C = grad_out.shape[1]
    if grad_out.shape != (B, C, N):
        raise ValueError(f"grad_out must be (B, C, N) = ({B}, C, {N}), got {tuple(grad_out.shape)}")
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    g_source = new(B, N, 3) if want_rays else None
    g_target = new(B, N, 3) if want_rays else None
    g_img = new(B, N) if want_img else None
    g_volume = torch.zeros_like(volume, memory_format=torch.contiguous_format) \
        if want_volume else None
    dh, dw, th, tw = _hints(det, tile, N)
    if _empty(B, N):
        return g_source, g_target, g_img, g_volume
    labels_u8, volume, grad_out = labels_u8.contiguous(), volume.contiguous(), grad_out.contiguous()
    source, target = source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    _launch(
        "ddrr_siddon_backward_channels", dev, volume.data_ptr(), labels_u8.data_ptr(),
        *volume.shape, source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img),
        grad_out.data_ptr(), B, N, int(C), float(voxel_shift), float(eps), dh, dw, th, tw,
        _ptr(g_source), _ptr(g_target), _ptr(g_img), _ptr(g_volume))
    return g_source, g_target, g_img, g_volume


def trilinear_alpha_range(source, target, volume_shape, *, voxel_shift=0.5, eps=1e-8):
    """The batch-global marching range of reference renderers.py:220-223 in one pass over the
    rays (no gradient flows through it).  -> (alphamin, alphamax), 0-dim device tensors."""
    _require_gpu(target)
    B, N, _ = target.shape
    if source.dtype != torch.float32 or target.dtype != torch.float32:
        raise NotImplementedError("trilinear_alpha_range needs float32 rays")
    source, target = source.contiguous(), target.contiguous()
    rng = torch.empty(2, dtype=torch.float32, device=target.device)
    _launch("ddrr_trilinear_alpha_range", target.device, source.data_ptr(), source.shape[1],
            target.data_ptr(), B, N, *(int(v) for v in volume_shape), float(voxel_shift),
            float(eps), rng.data_ptr())
    return rng[0], rng[1]


def trilinear_forward(volume, source, target, img, alphamin, alphamax, *, n_points=500,
                      voxel_shift=0.5, eps=1e-8, reducefn="sum", mode="bilinear",
                      align_corners=False, det=None, tile=None):
    """alphamin / alphamax: 0-dim (or 1-element) device tensors.  -> out (B,N)"""
    B, N = _check_rays(volume, source, target, img)
    out = torch.empty(B, N, dtype=torch.float32, device=volume.device)
    dh, dw, th, tw = _hints(det, tile, N)
    if _empty(B, N):
        return out
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    _launch(
        "ddrr_trilinear_forward", volume.device, volume.data_ptr(), *volume.shape, source.data_ptr(),
        source.shape[1], target.data_ptr(), _ptr(img), B, N, float(voxel_shift), float(eps),
        int(n_points), alphamin.data_ptr(), alphamax.data_ptr(), int(mode == "nearest"),
        reduce_code(reducefn), int(bool(align_corners)), dh, dw, th, tw, out.data_ptr())
    return out


def trilinear_forward_channels(volume, labels_u8, n_channels, source, target, img, alphamin,
                               alphamax, *, n_points=500, voxel_shift=0.5, eps=1e-8,
                               align_corners=False, det=None, tile=None):
    """Trilinear.forward with a mask (renderers.py:242-252).  -> (B, C, N)"""
    B, N = _check_rays(volume, source, target, img)
    if labels_u8.dtype != torch.uint8 or labels_u8.shape != volume.shape:
        raise ValueError("labels must be a uint8 tensor of the volume's shape")
    out = torch.empty(B, n_channels, N, dtype=torch.float32, device=volume.device)
    dh, dw, th, tw = _hints(det, tile, N)
    if _empty(B, N):
        return out
    labels_u8, volume = labels_u8.contiguous(), volume.contiguous()
    source, target = source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    _launch(
        "ddrr_trilinear_forward_channels", volume.device, volume.data_ptr(), labels_u8.data_ptr(),
        *volume.shape, source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img), B, N,
        int(n_channels), float(voxel_shift), float(eps), int(n_points), alphamin.data_ptr(),
        alphamax.data_ptr(), int(bool(align_corners)), dh, dw, th, tw, out.data_ptr())
    return out


def trilinear_forward_channels_bricks(volume, labels_u8, n_channels, source, target, img,
                                      alphamin, alphamax, det, *, n_points=500, voxel_shift=0.5,
                                      eps=1e-8):
    """The marcher's mask_to_channels for a detector grid on the volume-stationary bricks
    (ddrr_trilinear_forward_channels_bricks; mode "bilinear", align_corners=False).  -> (B, C, N)"""
    B, N = _check_rays(volume, source, target, img)
    H, W = int(det[0]), int(det[1])
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    if labels_u8.dtype != torch.uint8 or labels_u8.shape != volume.shape:
        raise ValueError("labels must be a uint8 tensor of the volume's shape")
    out = torch.empty(B, n_channels, N, dtype=torch.float32, device=volume.device)
    if _empty(B, N):
        return out
    labels_u8, volume = labels_u8.contiguous(), volume.contiguous()
    source, target = source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    _launch(
        "ddrr_trilinear_forward_channels_bricks", volume.device, volume.data_ptr(),
        labels_u8.data_ptr(), *volume.shape, source.data_ptr(), target.data_ptr(), _ptr(img), B,
        H, W, int(n_channels), float(voxel_shift), float(eps), int(n_points), alphamin.data_ptr(),
        alphamax.data_ptr(), out.data_ptr(),
        launch_workspace(volume.shape, volume.device).data_ptr())
    return out


TRI_AUX_PLANES = 7  # sum T, sum dT_xyz, sum alpha dT_xyz (include/diffdrr_hip.h)


def trilinear_backward_channels_volume_bricks(labels_u8, source, target, img, grad_out, alphamin,
                                              alphamax, det, *, n_points=500, voxel_shift=0.5,
                                              eps=1e-8):
    """Volume gradient of :func:`trilinear_forward_channels_bricks` for grad_out (B, C, N) on the
    owner bricks (LDS accumulator, labels in the words' low byte).  -> g_volume, the label map's
    shape"""
    B, N, _ = target.shape
    H, W = int(det[0]), int(det[1])
    C = grad_out.shape[1]
    if grad_out.shape != (B, C, N):
        raise ValueError(f"grad_out must be (B, C, N) = ({B}, C, {N}), got {tuple(grad_out.shape)}")
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    if labels_u8.dtype != torch.uint8 or labels_u8.dim() != 3:
        raise ValueError("labels must be a 3-D uint8 tensor")
    _require_gpu(target)
    dev = target.device
    labels_u8, grad_out = labels_u8.contiguous(), grad_out.contiguous()
    source, target = source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    g_volume = torch.empty(labels_u8.shape, dtype=torch.float32, device=dev)
    _launch("ddrr_trilinear_backward_channels_volume_bricks", dev, labels_u8.data_ptr(),
            *labels_u8.shape, source.data_ptr(), target.data_ptr(), _ptr(img), grad_out.data_ptr(), B,
            H, W, int(C), float(voxel_shift), float(eps), int(n_points), alphamin.data_ptr(),
            alphamax.data_ptr(), g_volume.data_ptr(), launch_workspace(labels_u8.shape, dev).data_ptr())
    return g_volume


def trilinear_backward_channels_bricks(volume, labels_u8, source, target, img, grad_out, alphamin,
                                       alphamax, det, *, n_points=500, voxel_shift=0.5, eps=1e-8,
                                       want_rays=True, want_img=True, want_alpha=True):
    """Ray / img / range gradients of :func:`trilinear_forward_channels_bricks` for grad_out
    (B, C, N) on the volume-stationary bricks: the brick kernel writes the marcher's record with
    every sample weighted by the incoming gradient of its channel,
    ``ddrr_trilinear_backward_rays`` turns it into gradients.  Results as
    :func:`trilinear_backward` (without g_volume)."""
    B, N = _check_rays(volume, source, target, img)
    H, W = int(det[0]), int(det[1])
    C = grad_out.shape[1]
    if grad_out.shape != (B, C, N):
        raise ValueError(f"grad_out must be (B, C, N) = ({B}, C, {N}), got {tuple(grad_out.shape)}")
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    dev = volume.device
    aux = torch.empty(TRI_AUX_PLANES, B, N, dtype=torch.float32, device=dev)
    ones = torch.ones(B, N, dtype=torch.float32, device=dev)
    if not _empty(B, N):
        labels_u8, volume, grad_out = labels_u8.contiguous(), volume.contiguous(), grad_out.contiguous()
        source, target = source.contiguous(), target.contiguous()
        _launch("ddrr_trilinear_backward_channels_bricks", dev, volume.data_ptr(),
                labels_u8.data_ptr(), *volume.shape, source.data_ptr(), target.data_ptr(),
                grad_out.data_ptr(), B, H, W, int(C), float(voxel_shift), float(eps), int(n_points),
                alphamin.data_ptr(), alphamax.data_ptr(), aux.data_ptr(),
                launch_workspace(volume.shape, dev).data_ptr())
    return trilinear_backward_rays(aux, ones, source, target, img, alphamin, alphamax,
                                   n_points=n_points, eps=eps, want_rays=want_rays,
                                   want_img=want_img, want_alpha=want_alpha)


def trilinear_forward_bricks(volume, source, target, img, alphamin, alphamax, det, *,
                             n_points=500, voxel_shift=0.5, eps=1e-8, want_aux=False):
    """Detector-grid trilinear march (bilinear, sum, align_corners=False) through the
    volume-stationary brick kernel.  -> out (B,N), or (out, aux (7,B,N)) with ``want_aux``:
    the planar backward record for :func:`trilinear_backward_rays`."""
    B, N = _check_rays(volume, source, target, img)
    H, W = int(det[0]), int(det[1])
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    out = torch.empty(B, N, dtype=torch.float32, device=volume.device)
    aux = torch.empty(TRI_AUX_PLANES, B, N, dtype=torch.float32, device=volume.device) \
        if want_aux else None
    if not _empty(B, N):
        _launch("ddrr_trilinear_forward_bricks", volume.device, volume.data_ptr(), *volume.shape,
                source.data_ptr(), target.data_ptr(), _ptr(img), B, H, W, float(voxel_shift),
                float(eps), int(n_points), alphamin.data_ptr(), alphamax.data_ptr(),
                out.data_ptr(), _ptr(aux),
                launch_workspace(volume.shape, volume.device).data_ptr())
    return (out, aux) if want_aux else out


def trilinear_backward_rays(aux, grad_out, source, target, img, alphamin, alphamax, *,
                            n_points=500, eps=1e-8, want_rays=True, want_img=True,
                            want_alpha=True):
    """Ray / range gradients of the march from the record of :func:`trilinear_forward_bricks`;
    results as :func:`trilinear_backward` (without g_volume)."""
    B, N, _ = target.shape
    dev = target.device
    _require_gpu(target)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    g_source = new(B, N, 3) if want_rays else None
    g_target = new(B, N, 3) if want_rays else None
    g_img = new(B, N) if want_img else None
    g_alpha = new(B, N, 2) if want_alpha else None
    res = {"g_source": g_source, "g_target": g_target, "g_img": g_img, "g_alpha": g_alpha,
           "g_volume": None}
    if _empty(B, N):
        return res
    source, target, grad_out = source.contiguous(), target.contiguous(), grad_out.contiguous()
    img = None if img is None else img.contiguous()
    _launch("ddrr_trilinear_backward_rays", dev, aux.data_ptr(), grad_out.data_ptr(),
            source.data_ptr(), target.data_ptr(), _ptr(img), B, N, float(eps), int(n_points),
            alphamin.data_ptr(), alphamax.data_ptr(), _ptr(g_source), _ptr(g_target), _ptr(g_img),
            _ptr(g_alpha))
    return res


def trilinear_backward_volume_bricks(volume_shape, source, target, img, grad_out, alphamin,
                                     alphamax, det, *, n_points=500, voxel_shift=0.5, eps=1e-8):
    """Volume gradient of the detector-grid trilinear march through the brick kernel (LDS
    accumulation).  -> g_volume (Dx,Dy,Dz)"""
    B, N, _ = target.shape
    H, W = int(det[0]), int(det[1])
    if H * W != N or source.shape[1] != 1 or min(H, W) < 2:
        raise ValueError("the brick path needs one source per pose and an H*W >= 2x2 ray grid")
    _require_gpu(target)
    source, target, grad_out = source.contiguous(), target.contiguous(), grad_out.contiguous()
    img = None if img is None else img.contiguous()
    Dx, Dy, Dz = (int(v) for v in volume_shape)
    g_volume = torch.empty(Dx, Dy, Dz, dtype=torch.float32, device=target.device)
    _launch("ddrr_trilinear_backward_volume_bricks", target.device, Dx, Dy, Dz,
            source.data_ptr(), target.data_ptr(), _ptr(img), grad_out.data_ptr(), B, H, W,
            float(voxel_shift), float(eps), int(n_points), alphamin.data_ptr(),
            alphamax.data_ptr(), g_volume.data_ptr(),
            launch_workspace((Dx, Dy, Dz), target.device).data_ptr())
    return g_volume


def trilinear_samples(volume, source, target, img, alphamin, alphamax, *, n_points=500,
                      voxel_shift=0.5, eps=1e-8, mode="bilinear", align_corners=False):
    """The per-sample terms a callable ``reducefn`` of the marcher receives
    (renderers.py:226-238).  -> (B, P, N); transpose(1, 2) is the reference's layout."""
    B, N = _check_rays(volume, source, target, img)
    out = torch.empty(B, int(n_points), N, dtype=torch.float32, device=volume.device)
    if _empty(B, N):
        return out
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    _launch("ddrr_trilinear_samples", volume.device, volume.data_ptr(), *volume.shape,
            source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img), B, N,
            float(voxel_shift), float(eps), int(n_points), alphamin.data_ptr(),
            alphamax.data_ptr(), int(mode == "nearest"), int(bool(align_corners)), out.data_ptr())
    return out


def trilinear_samples_backward(volume, source, target, img, grad_samples, alphamin, alphamax, *,
                               n_points=500, voxel_shift=0.5, eps=1e-8, mode="bilinear",
                               align_corners=False, want_rays=True, want_img=True,
                               want_alpha=True, want_volume=False):
    """Backward of :func:`trilinear_samples` for grad_samples (B, P, N); results as
    :func:`trilinear_backward`."""
    B, N = _check_rays(volume, source, target, img)
    dev = volume.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    g_source = new(B, N, 3) if want_rays else None
    g_target = new(B, N, 3) if want_rays else None
    g_img = new(B, N) if want_img else None
    g_alpha = new(B, N, 2) if want_alpha else None
    g_volume = torch.zeros_like(volume, memory_format=torch.contiguous_format) \
        if want_volume else None
    res = {"g_source": g_source, "g_target": g_target, "g_img": g_img, "g_alpha": g_alpha,
           "g_volume": g_volume}
    if _empty(B, N):
        return res
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    grad_samples = grad_samples.contiguous()
    _launch("ddrr_trilinear_samples_backward", dev, volume.data_ptr(), *volume.shape,
            source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img),
            grad_samples.data_ptr(), B, N, float(voxel_shift), float(eps), int(n_points),
            alphamin.data_ptr(), alphamax.data_ptr(), int(mode == "nearest"),
            int(bool(align_corners)), _ptr(g_source), _ptr(g_target), _ptr(g_img), _ptr(g_alpha),
            _ptr(g_volume))
    return res


def trilinear_backward_channels(volume, labels_u8, source, target, img, grad_out, alphamin,
                                alphamax, *, n_points=500, voxel_shift=0.5, eps=1e-8,
                                align_corners=False, want_rays=True, want_img=True,
                                want_alpha=True, want_volume=False, det=None, tile=None):
    """Backward of :func:`trilinear_forward_channels` for grad_out (B, C, N); results as
    :func:`trilinear_backward`."""
    B, N = _check_rays(volume, source, target, img)
    dev = volume.device
    C = grad_out.shape[1]
    if grad_out.shape != (B, C, N):
        raise ValueError(f"grad_out must be (B, C, N) = ({B}, C, {N}), got {tuple(grad_out.shape)}")
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    g_source = new(B, N, 3) if want_rays else None
    g_target = new(B, N, 3) if want_rays else None
    g_img = new(B, N) if want_img else None
    g_alpha = new(B, N, 2) if want_alpha else None
    g_volume = torch.zeros_like(volume, memory_format=torch.contiguous_format) \
        if want_volume else None
    dh, dw, th, tw = _hints(det, tile, N)
    res = {"g_source": g_source, "g_target": g_target, "g_img": g_img, "g_alpha": g_alpha,
           "g_volume": g_volume}
    if _empty(B, N):
        return res
    labels_u8, volume, grad_out = labels_u8.contiguous(), volume.contiguous(), grad_out.contiguous()
    source, target = source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    _launch(
        "ddrr_trilinear_backward_channels", dev, volume.data_ptr(), labels_u8.data_ptr(),
        *volume.shape, source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img),
        grad_out.data_ptr(), B, N, int(C), float(voxel_shift), float(eps), int(n_points),
        alphamin.data_ptr(), alphamax.data_ptr(), int(bool(align_corners)), dh, dw, th, tw,
        _ptr(g_source), _ptr(g_target), _ptr(g_img), _ptr(g_alpha), _ptr(g_volume))
    return res


def trilinear_backward(volume, source, target, img, grad_out, alphamin, alphamax, *, n_points=500,
                       voxel_shift=0.5, eps=1e-8, mode="bilinear", align_corners=False,
                       want_rays=True, want_img=True, want_alpha=True, want_volume=False,
                       det=None, tile=None, reducefn="sum"):
    """-> dict(g_source per ray, g_target, g_img, g_alpha (B,N,2), g_volume)"""
    B, N = _check_rays(volume, source, target, img)
    dev = volume.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    g_source = new(B, N, 3) if want_rays else None
    g_target = new(B, N, 3) if want_rays else None
    g_img = new(B, N) if want_img else None
    g_alpha = new(B, N, 2) if want_alpha else None
    g_volume = torch.zeros_like(volume, memory_format=torch.contiguous_format) \
        if want_volume else None
    dh, dw, th, tw = _hints(det, tile, N)
    res = {"g_source": g_source, "g_target": g_target, "g_img": g_img, "g_alpha": g_alpha,
           "g_volume": g_volume}
    if _empty(B, N):
        return res
    grad_out = grad_out.contiguous()
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    if reduce_code(reducefn) == REDUCE_MAX:
        _launch(
            "ddrr_trilinear_backward_max", dev, volume.data_ptr(), *volume.shape,
            source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img), grad_out.data_ptr(),
            B, N, float(voxel_shift), float(eps), int(n_points), alphamin.data_ptr(),
            alphamax.data_ptr(), int(mode == "nearest"), int(bool(align_corners)),
            _ptr(g_source), _ptr(g_target), _ptr(g_img), _ptr(g_alpha), _ptr(g_volume))
        return res
    _launch(
        "ddrr_trilinear_backward", dev, volume.data_ptr(), *volume.shape, source.data_ptr(),
        source.shape[1], target.data_ptr(), _ptr(img), grad_out.data_ptr(), B, N,
        float(voxel_shift), float(eps), int(n_points), alphamin.data_ptr(),
        alphamax.data_ptr(), int(mode == "nearest"), int(bool(align_corners)), dh, dw, th, tw,
        _ptr(g_source), _ptr(g_target), _ptr(g_img), _ptr(g_alpha), _ptr(g_volume))
    return res


# ---------------------------------------------------------------- double precision
# (`DRR(...).to(torch.float64)`, reference drr.py:71-75: per-ray kernels of csrc/f64_rays.hip)

def siddon_forward_f64(volume, source, target, img, *, voxel_shift=0.5, eps=1e-8, reducefn="sum",
                       want_aux=False):
    """-> (out (B,N) float64, aux (B,N,8) float64 | None)"""
    B, N = _check_rays(volume, source, target, img, torch.float64)
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    out = torch.empty(B, N, dtype=torch.float64, device=volume.device)
    aux = torch.empty(B, N, SIDDON_AUX, dtype=torch.float64, device=volume.device) \
        if want_aux else None
    if not _empty(B, N):
        _launch("ddrr_siddon_forward_f64", volume.device, volume.data_ptr(), *volume.shape,
                source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img), B, N,
                float(voxel_shift), float(eps), reduce_code(reducefn), out.data_ptr(), _ptr(aux))
    return out, aux


def siddon_backward_f64(volume_shape, source, target, img, grad_out, aux, *, voxel_shift=0.5,
                        eps=1e-8, want_rays=True, want_img=True, want_volume=False):
    """-> (g_source per ray (B,N,3), g_target (B,N,3), g_img (B,N), g_volume), None where not asked"""
    B, N, _ = target.shape
    dev = target.device
    new = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)  # noqa: E731
    g_source = new(B, N, 3) if want_rays else None
    g_target = new(B, N, 3) if want_rays else None
    g_img = new(B, N) if want_img else None
    Dx, Dy, Dz = (int(v) for v in volume_shape)
    g_volume = torch.zeros(Dx, Dy, Dz, dtype=torch.float64, device=dev) if want_volume else None
    if not _empty(B, N):
        source, target, grad_out = source.contiguous(), target.contiguous(), grad_out.contiguous()
        img = None if img is None else img.contiguous()
        _launch("ddrr_siddon_backward_f64", dev, Dx, Dy, Dz, source.data_ptr(), source.shape[1],
                target.data_ptr(), _ptr(img), grad_out.data_ptr(), _ptr(aux), B, N,
                float(voxel_shift), float(eps), _ptr(g_source), _ptr(g_target), _ptr(g_img),
                _ptr(g_volume))
    return g_source, g_target, g_img, g_volume


def trilinear_forward_f64(volume, source, target, img, alphamin, alphamax, *, n_points=500,
                          voxel_shift=0.5, eps=1e-8):
    B, N = _check_rays(volume, source, target, img, torch.float64)
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    out = torch.empty(B, N, dtype=torch.float64, device=volume.device)
    if not _empty(B, N):
        _launch("ddrr_trilinear_forward_f64", volume.device, volume.data_ptr(), *volume.shape,
                source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img), B, N,
                float(voxel_shift), float(eps), int(n_points), alphamin.data_ptr(),
                alphamax.data_ptr(), out.data_ptr())
    return out


def trilinear_backward_f64(volume, source, target, img, grad_out, alphamin, alphamax, *,
                           n_points=500, voxel_shift=0.5, eps=1e-8, want_rays=True, want_img=True,
                           want_alpha=True, want_volume=False):
    """-> dict(g_source per ray, g_target, g_img, g_alpha (B,N,2), g_volume)"""
    B, N = _check_rays(volume, source, target, img, torch.float64)
    dev = volume.device
    new = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)  # noqa: E731
    res = {"g_source": new(B, N, 3) if want_rays else None,
           "g_target": new(B, N, 3) if want_rays else None,
           "g_img": new(B, N) if want_img else None,
           "g_alpha": new(B, N, 2) if want_alpha else None,
           "g_volume": torch.zeros_like(volume, memory_format=torch.contiguous_format)
           if want_volume else None}
    if not _empty(B, N):
        volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
        img = None if img is None else img.contiguous()
        grad_out = grad_out.contiguous()
        _launch("ddrr_trilinear_backward_f64", dev, volume.data_ptr(), *volume.shape,
                source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img),
                grad_out.data_ptr(), B, N, float(voxel_shift), float(eps), int(n_points),
                alphamin.data_ptr(), alphamax.data_ptr(), _ptr(res["g_source"]),
                _ptr(res["g_target"]), _ptr(res["g_img"]), _ptr(res["g_alpha"]),
                _ptr(res["g_volume"]))
    return res


# ---------------------------------------------------------------- the materialising general path
# (csrc/general_rays.hip: the per-segment / per-sample tensors of the reference and their
# autograd, float32 or float64, for the keyword combinations the fused kernels do not take)

def _general_inputs(volume, source, target, img):
    if volume.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"the general path renders float32 or float64, not {volume.dtype}")
    B, N = _check_rays(volume, source, target, img, volume.dtype)
    volume, source, target = volume.contiguous(), source.contiguous(), target.contiguous()
    img = None if img is None else img.contiguous()
    return B, N, volume, source, target, img, int(volume.dtype == torch.float64)


def siddon_segments_general(volume, source, target, img, *, voxel_shift=0.5, eps=1e-8,
                            lookup="step", align_corners=False, raw=False):
    """-> terms (B, M-1, N), M = Dx+Dy+Dz+3: ``img * value * interval`` per segment
    (renderers.py:66-71), or with ``raw`` the looked-up values alone (the label lookup)."""
    B, N, volume, source, target, img, f64 = _general_inputs(volume, source, target, img)
    M1 = sum(volume.shape) + 2
    terms = torch.empty(B, M1, N, dtype=volume.dtype, device=volume.device)
    if not _empty(B, N):
        _launch("ddrr_siddon_segments_general", volume.device, volume.data_ptr(), f64,
                *volume.shape, source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img),
                B, N, float(voxel_shift), float(eps), _LOOKUP[lookup], int(bool(align_corners)),
                int(bool(raw)), terms.data_ptr())
    return terms


def siddon_segments_general_backward(volume, source, target, img, grad_terms, *, voxel_shift=0.5,
                                     eps=1e-8, lookup="step", align_corners=False,
                                     through_lookup=True, want_rays=True, want_img=True,
                                     want_volume=False):
    """-> (g_source per ray (B,N,3), g_target (B,N,3), g_img (B,N), g_volume), None where not asked"""
    B, N, volume, source, target, img, f64 = _general_inputs(volume, source, target, img)
    dev = volume.device
    new = lambda *s: torch.empty(*s, dtype=volume.dtype, device=dev)  # noqa: E731
    g_source = new(B, N, 3) if want_rays else None
    g_target = new(B, N, 3) if want_rays else None
    g_img = new(B, N) if want_img and through_lookup else None
    g_volume = torch.zeros_like(volume, memory_format=torch.contiguous_format) \
        if want_volume and through_lookup else None
    if not _empty(B, N):
        grad_terms = grad_terms.to(volume.dtype).contiguous()
        _launch("ddrr_siddon_segments_general_backward", dev, volume.data_ptr(), f64,
                *volume.shape, source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img),
                grad_terms.data_ptr(), B, N, float(voxel_shift), float(eps), _LOOKUP[lookup],
                int(bool(align_corners)), int(bool(through_lookup)), _ptr(g_source),
                _ptr(g_target), _ptr(g_img), _ptr(g_volume))
    elif g_img is not None:
        g_img.zero_()
    return g_source, g_target, g_img, g_volume


def trilinear_samples_general(volume, source, target, img, alphamin, alphamax, *, n_points=500,
                              voxel_shift=0.5, eps=1e-8, mode="bilinear", align_corners=False,
                              raw=False):
    """-> samples (B, P, N): ``img * step * value`` per sample (renderers.py:224-236), or with
    ``raw`` the looked-up values alone (the label lookup)."""
    B, N, volume, source, target, img, f64 = _general_inputs(volume, source, target, img)
    samples = torch.empty(B, int(n_points), N, dtype=volume.dtype, device=volume.device)
    if not _empty(B, N):
        alphamin = alphamin.to(volume.dtype).reshape(1).contiguous()
        alphamax = alphamax.to(volume.dtype).reshape(1).contiguous()
        _launch("ddrr_trilinear_samples_general", volume.device, volume.data_ptr(), f64,
                *volume.shape, source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img),
                B, N, float(voxel_shift), float(eps), int(n_points), alphamin.data_ptr(),
                alphamax.data_ptr(), int(mode == "nearest"), int(bool(align_corners)),
                int(bool(raw)), samples.data_ptr())
    return samples


def trilinear_samples_general_backward(volume, source, target, img, grad_samples, alphamin,
                                       alphamax, *, n_points=500, voxel_shift=0.5, eps=1e-8,
                                       mode="bilinear", align_corners=False, want_rays=True,
                                       want_img=True, want_alpha=True, want_volume=False):
    """-> dict(g_source per ray, g_target, g_img, g_alpha (B,N,2) per ray, g_volume)"""
    B, N, volume, source, target, img, f64 = _general_inputs(volume, source, target, img)
    dev = volume.device
    new = lambda *s: torch.zeros(*s, dtype=volume.dtype, device=dev)  # noqa: E731
    res = {"g_source": new(B, N, 3) if want_rays else None,
           "g_target": new(B, N, 3) if want_rays else None,
           "g_img": new(B, N) if want_img else None,
           "g_alpha": new(B, N, 2) if want_alpha else None,
           "g_volume": torch.zeros_like(volume, memory_format=torch.contiguous_format)
           if want_volume else None}
    if not _empty(B, N):
        alphamin = alphamin.to(volume.dtype).reshape(1).contiguous()
        alphamax = alphamax.to(volume.dtype).reshape(1).contiguous()
        grad_samples = grad_samples.to(volume.dtype).contiguous()
        _launch("ddrr_trilinear_samples_general_backward", dev, volume.data_ptr(), f64,
                *volume.shape, source.data_ptr(), source.shape[1], target.data_ptr(), _ptr(img),
                grad_samples.data_ptr(), B, N, float(voxel_shift), float(eps), int(n_points),
                alphamin.data_ptr(), alphamax.data_ptr(), int(mode == "nearest"),
                int(bool(align_corners)), _ptr(res["g_source"]), _ptr(res["g_target"]),
                _ptr(res["g_img"]), _ptr(res["g_alpha"]), _ptr(res["g_volume"]))
    return res
