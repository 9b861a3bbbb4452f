"""ctypes binding of ``libdiffdrr_hip.so`` (C ABI: ``include/diffdrr_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``hipcc
--offload-arch=gfx950``) and loaded *after* torch so that it binds to the HIP
runtime torch already has in the process (same ``libamdhip64.so.7`` soname):
kernels can then be launched on torch's current stream with torch's device
pointers.  There is deliberately no CPU or pure-PyTorch fallback: if the
library is missing, rendering raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdiffdrr_hip.so")
ABI_VERSION = 33
BRICKS_CLEARED = 2  # include/diffdrr_hip.h DDRR_BRICKS_CLEARED (a bit of ranges_valid)

REDUCE_SUM, REDUCE_MAX = 0, 1
LOOKUP_STEP, LOOKUP_MID_NEAREST, LOOKUP_MID_TRILINEAR = 0, 1, 2
SIDDON_AUX = 8
AUX_INTERLEAVED, AUX_BLOCKED, AUX_PACKED = 0, 1, 2
REC_BLOCK_RAYS, REC_BLOCK_FLOATS = 16, 80  # blocked float record (csrc/record_layout.h)
BRICKS_F32, BRICKS_Q16, BRICKS_Q16_PACKED = 0, 1, 2  # how a brick is held in LDS (ddrr_siddon_forward_bricks)
PACKED_AUX_PLANES = 7  # fixed-point record (csrc/record_pack.h)

_P, _I, _F, _L, _D = c_void_p, c_int, c_float, c_long, c_double

# name -> argtypes, in the order of include/diffdrr_hip.h
_SIGNATURES = {
    "ddrr_siddon_forward": [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _F, _F, _I, _I, _I, _I, _I,
                            _I, _I, _P, _P, _P, _P],
    "ddrr_siddon_forward_bricks": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _F, _I,
                                   _P, _I, _P, _P],
    "ddrr_siddon_forward_bricks_masked": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _F, _I,
                                          _P, _I, _P, _P, _P],
    "ddrr_siddon_backward_rays": [_P, _I, _P, _P, _I, _P, _P, _I, _I, _F, _I, _P, _P, _P, _P],
    "ddrr_siddon_backward_volume": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _I, _I,
                                    _I, _I, _I, _P, _P],
    "ddrr_siddon_forward_channels": [_P, _P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _I, _F, _F, _I,
                                     _I, _I, _I, _P, _P],
    "ddrr_siddon_forward_channels_bricks": [_P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _F,
                                            _P, _P, _P],
    "ddrr_channel_words": [_P, _P, _L, _I, _P, _P, _I, _P],
    "ddrr_siddon_forward_channels_bricks_words": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _F,
                                                  _P, _P, _P],
    "ddrr_siddon_backward_channels_bricks": [_P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _F,
                                             _P, _P, _P],
    "ddrr_siddon_backward_channels_volume_bricks": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _F,
                                                    _F, _P, _P, _P],
    "ddrr_trilinear_backward_channels_volume_bricks": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I,
                                                       _F, _F, _I, _P, _P, _P, _P, _P],
    "ddrr_trilinear_backward_channels_bricks": [_P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F,
                                                _F, _I, _P, _P, _P, _P, _P],
    "ddrr_trilinear_alpha_range": [_P, _I, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P],
    "ddrr_trilinear_backward_max": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _I, _P, _P,
                                    _I, _I, _P, _P, _P, _P, _P, _P],
    "ddrr_trilinear_samples": [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _F, _F, _I, _P, _P, _I, _I,
                               _P, _P],
    "ddrr_trilinear_samples_backward": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _I, _P,
                                        _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "ddrr_siddon_backward_midpoint": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _I, _I,
                                      _P, _P, _P, _P, _P],
    "ddrr_siddon_segments": [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _F, _F, _P, _P],
    "ddrr_siddon_segments_backward": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _P, _P,
                                      _P, _P, _P],
    "ddrr_siddon_backward_channels": [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _F, _F,
                                      _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "ddrr_trilinear_backward_channels": [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _F,
                                         _F, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P,
                                         _P],
    "ddrr_trilinear_forward": [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _F, _F, _I, _P, _P, _I,
                               _I, _I, _I, _I, _I, _I, _P, _P],
    "ddrr_trilinear_backward": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _I, _P, _P,
                                _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "ddrr_siddon_backward_volume_bricks": [_I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P,
                                           _P],
    "ddrr_trilinear_forward_channels": [_P, _P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _I, _F, _F, _I,
                                        _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "ddrr_trilinear_forward_bricks": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P, _P,
                                      _P, _P, _P, _P],
    "ddrr_trilinear_backward_rays": [_P, _P, _P, _P, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P,
                                     _P],
    "ddrr_trilinear_backward_volume_bricks": [_I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I,
                                              _P, _P, _P, _P, _P],
    "ddrr_pose_euler_forward": [_P, _P, _I, _I, _I, _P, _I, _P, _P],
    "ddrr_pose_euler_backward": [_P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P],
    "ddrr_siddon_ncc_workspace_bytes": [_I],
    "ddrr_pose_raygen_forward": [_P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _L, _P, _P],
    "ddrr_siddon_ncc_forward": [_P, _P, _P, _L, _I, _I, _F, _P, _P, _P, _P, _P, _P],
    "ddrr_siddon_ncc_backward_pose": [_P, _P, _P, _L, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I,
                                      _P, _I, _I, _F, _I, _P, _P, _P, _P],
    "ddrr_siddon_backward_pose_euler": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _F, _I, _P,
                                        _P, _P, _P],
    "ddrr_pose_adam_step": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _I, _P],
    "ddrr_ncc_forward": [_P, _L, _P, _I, _I, _F, _P, _P, _P],
    "ddrr_ncc_backward": [_P, _L, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "ddrr_ncc_patch_forward": [_P, _L, _P, _I, _I, _I, _I, _F, _P, _P, _P],
    "ddrr_ncc_patch_backward": [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "ddrr_sobel_forward": [_P, _I, _I, _I, _P, _P],
    "ddrr_sobel_backward": [_P, _I, _I, _I, _P, _P],
    "ddrr_blur_sobel_forward": [_P, _L, _I, _I, _I, _P, _I, _P, _P],
    "ddrr_blur_sobel_backward": [_P, _I, _I, _I, _P, _I, _P, _P],
    "ddrr_raygen_forward": [_P, _P, _P, _I, _I, _P, _P, _P, _P],
    "ddrr_siddon_backward_pose": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P, _P],
    "ddrr_siddon_forward_f64": [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _D, _D, _I, _P, _P, _P],
    "ddrr_siddon_backward_f64": [_I, _I, _I, _P, _I, _P, _P, _P, _P, _I, _I, _D, _D, _P, _P, _P,
                                 _P, _P],
    "ddrr_trilinear_forward_f64": [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _D, _D, _I, _P, _P, _P,
                                   _P],
    "ddrr_trilinear_backward_f64": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _D, _D, _I, _P, _P,
                                    _P, _P, _P, _P, _P, _P],
    "ddrr_siddon_segments_general": [_P, _I, _I, _I, _I, _P, _I, _P, _P, _I, _I, _D, _D, _I, _I, _I,
                                     _P, _P],
    "ddrr_siddon_segments_general_backward": [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _D,
                                              _D, _I, _I, _I, _P, _P, _P, _P, _P],
    "ddrr_trilinear_samples_general": [_P, _I, _I, _I, _I, _P, _I, _P, _P, _I, _I, _D, _D, _I, _P,
                                       _P, _I, _I, _I, _P, _P],
    "ddrr_trilinear_samples_general_backward": [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _D,
                                                _D, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "ddrr_channel_words_state_bytes": [],
    "ddrr_brick_workspace_bytes": [_I, _I, _I, _I],
    "ddrr_brick_launch_workspace_bytes": [_I, _I, _I],
    "ddrr_trilinear_forward_channels_bricks": [_P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _F,
                                               _I, _P, _P, _P, _P, _P],
}
# (entries that return a size, not a status)
_RESTYPES = {"ddrr_brick_workspace_bytes": c_long, "ddrr_brick_launch_workspace_bytes": c_long,
             "ddrr_siddon_ncc_workspace_bytes": c_long, "ddrr_channel_words_state_bytes": c_long}
EXPORTS = ["ddrr_abi_version", "ddrr_last_error", *_SIGNATURES]


class DdrrLibrary:
    """A loaded implementation of the C ABI (the HIP product library; the
    tests also bind their host emulation build through this class)."""

    def __init__(self, path: str):
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name in EXPORTS:
            if not hasattr(self.cdll, name):
                raise RuntimeError(f"{path} does not export {name}")
        self.cdll.ddrr_abi_version.restype = c_int
        self.cdll.ddrr_last_error.restype = ctypes.c_char_p
        got = self.cdll.ddrr_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version {got}, expected {ABI_VERSION}")
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(self.cdll, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)

    def query(self, name: str, *args):
        """An entry that returns a value (``_RESTYPES``), not a status."""
        return getattr(self.cdll, name)(*args)

    def call(self, name: str, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            msg = self.cdll.ddrr_last_error().decode(errors="replace")
            raise RuntimeError(f"{name} failed (code {rc}): {msg}")


_lib: DdrrLibrary | None = None


def get_lib() -> DdrrLibrary:
    """The HIP library, loaded on first use.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (must own the HIP runtime before we bind to it)

        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X renderers have not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); "
                "diffdrr_amd has no CPU fallback."
            )
        _lib = DdrrLibrary(LIB_PATH)
    return _lib
