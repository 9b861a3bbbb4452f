"""CPU oracle for the DRR hot path -- TEST INFRASTRUCTURE ONLY.

Thin numpy/ctypes front-end over ``oracle/drr_oracle.c`` (a per-ray C
restatement of the reference's ``diffdrr/renderers.py``; see that file's
header for the file:line map and for how it is pinned).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  ``diffdrr_amd`` never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdrr_oracle.so")
_lib = None

REDUCE = {"sum": 0, "max": 1}


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("drr_oracle.c", "drr_oracle_impl.h")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        assert _lib.oracle_abi_version() == 1
    return _lib


def _sfx(dtype):
    return {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[np.dtype(dtype)]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _real(dtype, v):
    return ctypes.c_float(v) if np.dtype(dtype) == np.float32 else ctypes.c_double(v)


def _prep(volume, source, target, img):
    dtype = volume.dtype
    volume = np.ascontiguousarray(volume)
    source = np.ascontiguousarray(source, dtype=dtype)
    target = np.ascontiguousarray(target, dtype=dtype)
    B, N, _ = target.shape
    assert source.shape[0] == B and source.shape[1] in (1, N) and source.shape[2] == 3
    if img is not None:
        img = np.ascontiguousarray(img, dtype=dtype).reshape(B, N)
    return dtype, volume, source, target, img, B, N


def siddon(volume, source, target, img=None, *, voxel_shift=0.5, eps=1e-8, reducefn="sum",
           mode="nearest", align_corners=False, grad_out=None, want_volume_grad=False,
           count_voxels=False):
    """Siddon.forward (+ analytic autograd when ``grad_out`` is given).

    Returns a dict with ``out`` (B,1,N) and, if requested, ``g_source``
    (B,src_n,3), ``g_target`` (B,N,3), ``g_img`` (B,1,N), ``g_volume``
    (Dx,Dy,Dz), ``n_inside`` (B,N)."""
    dtype, volume, source, target, img, B, N = _prep(volume, source, target, img)
    Dx, Dy, Dz = volume.shape
    out = np.empty((B, N), dtype)
    go = gs = gt = gi = gv = ni = None
    if grad_out is not None:
        go = np.ascontiguousarray(grad_out, dtype=dtype).reshape(B, N)
        gs = np.zeros((B, N, 3), np.float64)
        gt = np.zeros((B, N, 3), np.float64)
        gi = np.zeros((B, N), np.float64)
        if want_volume_grad:
            gv = np.zeros((Dx, Dy, Dz), np.float64)
    if count_voxels:
        ni = np.zeros((B, N), np.int64)
    fn = getattr(lib(), "oracle_siddon_" + _sfx(dtype))
    fn.restype = None
    fn(_p(volume), Dx, Dy, Dz, _p(source), source.shape[1], _p(target), _p(img), B, N,
       _real(dtype, voxel_shift), _real(dtype, eps), REDUCE[reducefn],
       int(mode == "bilinear"), int(bool(align_corners)), _p(out), _p(go), _p(gs), _p(gt), _p(gi),
       _p(gv), _p(ni))
    res = {"out": out.reshape(B, 1, N)}
    if go is not None:
        res["g_source"] = gs.sum(1, keepdims=True) if source.shape[1] == 1 else gs
        res["g_target"] = gt
        res["g_img"] = gi.reshape(B, 1, N)
        if gv is not None:
            res["g_volume"] = gv
    if ni is not None:
        res["n_inside"] = ni
    return res


def siddon_segments(volume, source, target, img=None, mask=None, *, voxel_shift=0.5, eps=1e-8):
    """The (B,N,M-1) per-segment terms (renderers.py:71) and per-segment labels."""
    dtype, volume, source, target, img, B, N = _prep(volume, source, target, img)
    Dx, Dy, Dz = volume.shape
    M = Dx + Dy + Dz + 3
    terms = np.zeros((B, N, M - 1), dtype)
    labels = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=dtype)
        labels = np.zeros((B, N, M - 1), dtype)
    fn = getattr(lib(), "oracle_siddon_segments_" + _sfx(dtype))
    fn.restype = None
    fn(_p(volume), _p(mask), Dx, Dy, Dz, _p(source), source.shape[1], _p(target), _p(img), B, N,
       _real(dtype, voxel_shift), _real(dtype, eps), _p(terms), _p(labels))
    return terms, labels


def siddon_channels(volume, mask, source, target, img=None, *, n_channels=None, **kw):
    """mask_to_channels branch of Siddon.forward (renderers.py:77-89)."""
    terms, labels = siddon_segments(volume, source, target, img, mask, **kw)
    C = int(mask.max() + 1) if n_channels is None else n_channels
    B, N, _ = terms.shape
    out = np.zeros((B, C, N), np.float64)
    lab = labels.astype(np.int64)
    for c in range(C):
        out[:, c, :] = np.where(lab == c, terms.astype(np.float64), 0.0).sum(-1)
    return out.astype(terms.dtype)


def siddon_channels_grad(volume, mask, source, target, img, grad_out, **kw):
    """Autograd of the mask_to_channels branch (renderers.py:77-89) for grad_out (B, C, N).
    With a nearest lookup, channel c is exactly the Siddon render of the volume with every
    voxel not labelled c set to zero, so the gradients are the sum over channels of the
    single-channel analytic gradients (g_volume of channel c masked to its voxels)."""
    mask = np.asarray(mask)
    C = grad_out.shape[1]
    tot = None
    for c in range(C):
        sel = mask == c
        r = siddon(np.where(sel, volume, 0), source, target, img, grad_out=grad_out[:, c, :],
                   want_volume_grad=True, **kw)
        r["g_volume"] = np.where(sel, r["g_volume"], 0)
        if tot is None:
            tot = {k: np.array(r[k], dtype=np.float64) for k in ("g_source", "g_target", "g_img", "g_volume")}
        else:
            for k in tot:
                tot[k] += r[k]
    return tot


def alpha_minmax(source, target, dims, *, voxel_shift=0.5, eps=1e-8):
    """_get_alpha_minmax (renderers.py:124-140), per ray -> (B,N,1) each."""
    dtype = target.dtype
    source = np.ascontiguousarray(source, dtype=dtype)
    target = np.ascontiguousarray(target)
    B, N, _ = target.shape
    amin = np.empty((B, N), dtype)
    amax = np.empty((B, N), dtype)
    fn = getattr(lib(), "oracle_alpha_minmax_" + _sfx(dtype))
    fn.restype = None
    fn(_p(source), source.shape[1], _p(target), B, N, int(dims[0]), int(dims[1]), int(dims[2]),
       _real(dtype, voxel_shift), _real(dtype, eps), _p(amin), _p(amax))
    return amin[..., None], amax[..., None]


def linspace01(n_points):
    """The fp32 table torch.linspace(0, 1, P) (renderers.py:224).  aten's CPU
    kernel is used when torch is importable (it always is where tests run); the
    numpy fallback is the scalar formula aten documents."""
    try:
        import torch

        return torch.linspace(0, 1, int(n_points)).numpy().astype(np.float32, copy=True)
    except ImportError:  # pragma: no cover
        P = int(n_points)
        step = np.float32(1) / np.float32(P - 1)
        i = np.arange(P)
        lo = (i.astype(np.float32) * step).astype(np.float32)
        hi = (np.float32(1) - (P - 1 - i).astype(np.float32) * step).astype(np.float32)
        return np.where(i < P // 2, lo, hi).astype(np.float32)


def trilinear(volume, source, target, img=None, *, n_points=500, alphamin=None, alphamax=None,
              voxel_shift=0.5, eps=1e-8, reducefn="sum", mode="bilinear", align_corners=False,
              grad_out=None, want_volume_grad=False):
    """Trilinear.forward (+ analytic autograd when ``grad_out`` is given)."""
    dtype, volume, source, target, img, B, N = _prep(volume, source, target, img)
    Dx, Dy, Dz = volume.shape
    if alphamin is None or alphamax is None:
        lo, hi = alpha_minmax(source, target, volume.shape, voxel_shift=voxel_shift, eps=eps)
        alphamin, alphamax = lo.min(), hi.max()  # renderers.py:222-223
    out = np.empty((B, N), dtype)
    go = gs = gt = gi = gv = None
    ga = np.zeros(2, np.float64)
    if grad_out is not None:
        go = np.ascontiguousarray(grad_out, dtype=dtype).reshape(B, N)
        gs = np.zeros((B, N, 3), np.float64)
        gt = np.zeros((B, N, 3), np.float64)
        gi = np.zeros((B, N), np.float64)
        if want_volume_grad:
            gv = np.zeros((Dx, Dy, Dz), np.float64)
    lin01 = linspace01(n_points)
    fn = getattr(lib(), "oracle_trilinear_" + _sfx(dtype))
    fn.restype = None
    fn(_p(volume), Dx, Dy, Dz, _p(source), source.shape[1], _p(target), _p(img), B, N,
       _real(dtype, voxel_shift), _real(dtype, eps), int(n_points), _p(lin01),
       _real(dtype, float(alphamin)), _real(dtype, float(alphamax)), int(mode == "bilinear"),
       REDUCE[reducefn], int(bool(align_corners)), _p(out), _p(go), _p(gs), _p(gt), _p(gi), _p(gv),
       ga[0:1].ctypes.data_as(ctypes.c_void_p), ga[1:2].ctypes.data_as(ctypes.c_void_p))
    res = {"out": out.reshape(B, 1, N), "alphamin": alphamin, "alphamax": alphamax}
    if go is not None:
        res["g_source"] = gs.sum(1, keepdims=True) if source.shape[1] == 1 else gs
        res["g_target"] = gt
        res["g_img"] = gi.reshape(B, 1, N)
        res["g_alphamin"] = ga[0]
        res["g_alphamax"] = ga[1]
        if gv is not None:
            res["g_volume"] = gv
    return res
