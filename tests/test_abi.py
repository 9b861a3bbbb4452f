"""The C ABI: include/diffdrr_hip.h <-> ctypes signatures <-> built library.
No compute is issued here (no GPU needed)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT
from diffdrr_amd import _lib

HEADER = os.path.join(ROOT, "include", "diffdrr_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|long|const char \*)\s*(ddrr_\w+)\s*\(([^;]*?)\)\s*;", text, re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args == "void" else len(args.split(","))
    return decls


def test_header_matches_ctypes_signatures():
    decls = _declared()
    assert set(decls) == set(_lib.EXPORTS)
    for name, argtypes in _lib._SIGNATURES.items():
        assert decls[name] == len(argtypes), name


def test_header_constants_match():
    text = open(HEADER).read()
    const = dict(re.findall(r"#define (DDRR_\w+) (\d+)", text))
    assert int(const["DDRR_ABI_VERSION"]) == _lib.ABI_VERSION
    assert int(const["DDRR_REDUCE_SUM"]) == _lib.REDUCE_SUM
    assert int(const["DDRR_REDUCE_MAX"]) == _lib.REDUCE_MAX
    assert int(const["DDRR_LOOKUP_STEP"]) == _lib.LOOKUP_STEP
    assert int(const["DDRR_LOOKUP_MID_NEAREST"]) == _lib.LOOKUP_MID_NEAREST
    assert int(const["DDRR_LOOKUP_MID_TRILINEAR"]) == _lib.LOOKUP_MID_TRILINEAR
    assert int(const["DDRR_SIDDON_AUX"]) == _lib.SIDDON_AUX
    assert int(const["DDRR_BRICKS_CLEARED"]) == _lib.BRICKS_CLEARED == 2  # (a bit of ranges_valid, next to bit 0)


def test_library_builds_loads_and_exports_every_symbol():
    """hipcc cross-compiles gfx950 without a GPU; loading the .so and reading its
    ABI version needs no device."""
    import __graft_entry__ as entry

    entry.build_hip()
    assert os.path.exists(_lib.LIB_PATH)
    lib = _lib.DdrrLibrary(_lib.LIB_PATH)  # checks every export + the ABI version
    assert lib.cdll.ddrr_abi_version() == _lib.ABI_VERSION
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                          text=True, check=True).stdout
    for name in _declared():
        assert re.search(rf"\bT {name}\b", syms), name
    # ... and nothing else: experiment switches (ddrr_set_*, the phase profile) exist only in
    # the development builds of tools/explib.py, never in the product library
    exported = set(re.findall(r"\bT (ddrr_\w+)", syms))
    assert exported == set(_declared()), exported ^ set(_declared())
    # ... of ANY kind: no kernel stub, no C++ helper shared by the translation units, no data
    # (csrc/exports.map; round 4 leaked `__device_stub__alpha_range_kernel` and the ddrr_rt:: /
    # ddrr_brick:: functions)
    every = {line.split()[-1] for line in syms.splitlines() if line.strip()}
    assert every == set(_declared()), every ^ set(_declared())


def test_every_entry_rejects_null_pointers_and_negative_sizes_before_any_launch():
    """The boundary's error behaviour (include/diffdrr_hip.h: "0 on success, a negative code and
    ddrr_last_error() otherwise"): every status-returning entry point checks its arguments BEFORE it
    touches a device -- so it can be shown without one.  All pointers NULL: -1 and a message that names
    the pointers; pointers to host memory with every size -1: -1 and a message that names the size (or
    the enum) that is wrong.  No entry gets as far as a launch (which, here, would report the missing
    device instead).  The reference raises Python exceptions for bad arguments (drr.py:99-101,
    renderers.py:183); through the binding these codes surface as RuntimeError (_lib.DdrrLibrary.call)."""
    import ctypes

    import __graft_entry__ as entry

    entry.build_hip()
    lib = _lib.DdrrLibrary(_lib.LIB_PATH)
    buf = (ctypes.c_char * 4096)()
    addr = ctypes.addressof(buf)
    status_entries = [n for n in _lib._SIGNATURES if n not in _lib._RESTYPES]
    assert len(status_entries) >= 54
    for name in status_entries:
        argtypes = _lib._SIGNATURES[name]
        for pointers, ints, expect in ((None, 0, "null"), (addr, -1, None)):
            args = [pointers if t is _lib._P else (ints if t in (_lib._I, _lib._L) else 0.0) for t in argtypes]
            args[-1] = None  # the stream
            rc = getattr(lib.cdll, name)(*args)
            msg = lib.cdll.ddrr_last_error().decode(errors="replace")
            assert rc < 0 and msg, (name, rc, msg)
            assert "hip" not in msg.lower() and "device" not in msg.lower(), (name, msg)  # (never launched)
            if expect:
                assert expect in msg, (name, msg)
        # ... and through the binding: an exception that carries the entry's name and its message
        with pytest.raises(RuntimeError, match=name):
            lib.call(name, *[None if t is _lib._P else (0 if t in (_lib._I, _lib._L) else 0.0) for t in argtypes])
    # argument rules of the image-space similarity entries, one by one
    for name, args, what in (
            ("ddrr_blur_sobel_forward", (addr, 64, 1, 8, 8, addr, 4, addr, None), "odd number of taps"),
            ("ddrr_blur_sobel_forward", (addr, 64, 1, 8, 8, addr, 33, addr, None), "odd number of taps"),
            ("ddrr_blur_sobel_forward", (addr, 16, 1, 4, 4, addr, 9, addr, None), "reflect padding"),
            ("ddrr_blur_sobel_forward", (addr, 3, 1, 8, 8, addr, 7, addr, None), "img_stride"),
            ("ddrr_blur_sobel_backward", (addr, 1, 8, 3, addr, 7, addr, None), "reflect padding"),
            ("ddrr_ncc_patch_forward", (addr, 64, addr, 1, 8, 8, 9, 1e-5, addr, None, None), "patch_size"),
            ("ddrr_ncc_patch_forward", (addr, 5, addr, 1, 8, 8, 3, 1e-5, addr, None, None), "x1_stride"),
            ("ddrr_ncc_patch_backward", (addr, 64, addr, addr, addr, 2, 1, 8, 8, 3, addr, None), "g_stride"),
            ("ddrr_sobel_forward", (addr, 70000, 8, 8, addr, None), "65535")):
        with pytest.raises(RuntimeError, match=what):
            lib.call(name, *args)


def test_library_contains_gfx950_code_object():
    import __graft_entry__ as entry

    entry.build_hip()
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"siddon_fwd_kernel" in blob


def test_product_refuses_cpu_tensors():
    import torch

    from diffdrr_amd import Siddon

    vol = torch.rand(4, 4, 4)
    src = torch.zeros(1, 1, 3)
    tgt = torch.ones(1, 5, 3)
    with pytest.raises(RuntimeError, match="MI355X only"):
        Siddon()(vol, src, tgt, torch.ones(1, 1, 5))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.get_lib()


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "diffdrr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "ref_loader" not in text and "/root/reference" not in text, f


def test_argument_checks_return_errors_without_touching_a_device():
    """Every entry validates its arguments before the first HIP call: bad calls return a
    non-zero code and leave a message in ddrr_last_error() (no GPU needed to see that)."""
    import ctypes

    import __graft_entry__ as entry

    entry.build_hip()
    lib = _lib.DdrrLibrary(_lib.LIB_PATH)
    c = lib.cdll
    c.ddrr_last_error.restype = ctypes.c_char_p
    buf = (ctypes.c_float * 64)()
    P = ctypes.cast(buf, ctypes.c_void_p)
    f = ctypes.c_float
    # null volume
    rc = c.ddrr_siddon_forward(None, 4, 4, 4, P, 1, P, P, 1, 5, f(0.5), f(1e-8), 0, 0, 0, 0, 0, 1, 64,
                               P, None, None, None)
    assert rc != 0 and b"null" in c.ddrr_last_error()
    # bad dims / src_n / negative batch
    assert c.ddrr_siddon_forward(P, 0, 4, 4, P, 1, P, P, 1, 5, f(0.5), f(1e-8), 0, 0, 0, 0, 0, 1, 64,
                                 P, None, None, None) != 0
    assert b"dims" in c.ddrr_last_error()
    assert c.ddrr_siddon_forward(P, 4, 4, 4, P, 3, P, P, 1, 5, f(0.5), f(1e-8), 0, 0, 0, 0, 0, 1, 64,
                                 P, None, None, None) != 0
    assert b"src_n" in c.ddrr_last_error()
    assert c.ddrr_trilinear_forward(P, 4, 4, 4, P, 1, P, P, -1, 5, f(0.5), f(1e-8), 8, P, P, 0, 0,
                                    0, 0, 0, 1, 64, P, None) != 0
    # the marcher needs at least two samples; the brick paths a 2x2 detector
    assert c.ddrr_trilinear_forward(P, 4, 4, 4, P, 1, P, P, 1, 5, f(0.5), f(1e-8), 1, P, P, 0, 0, 0,
                                    0, 0, 1, 64, P, None) != 0
    assert b"n_points" in c.ddrr_last_error()
    assert c.ddrr_siddon_forward_bricks(P, 4, 4, 4, P, P, P, 1, 1, 5, f(0.5), f(1e-8), P, None,
                                        f(0.0), 0, None, 0, P, None) != 0
    assert b"2x2" in c.ddrr_last_error()
    # 16-bit bricks need their range workspace
    assert c.ddrr_siddon_forward_bricks(P, 4, 4, 4, P, P, P, 1, 4, 5, f(0.5), f(1e-8), P, None,
                                        f(0.0), 1, None, 0, P, None) != 0
    assert b"brick_ranges" in c.ddrr_last_error()
    # every *_bricks entry takes its per-launch state from the caller (no device-side globals)
    assert c.ddrr_brick_launch_workspace_bytes(512, 512, 512) == 256 + 4096 * 8
    assert c.ddrr_brick_launch_workspace_bytes(0, 4, 4) == 0
    # an empty batch is a valid no-op
    assert c.ddrr_siddon_forward(P, 4, 4, 4, P, 1, P, P, 0, 5, f(0.5), f(1e-8), 0, 0, 0, 0, 0, 1, 64,
                                 P, None, None, None) == 0
    assert c.ddrr_pose_euler_forward(P, P, 0, 0, 1, P, 1, P, None) != 0  # repeated axis
    # the fused registration step (ABI 29): the same checks as the entries it fuses
    L = ctypes.c_long
    assert c.ddrr_siddon_ncc_workspace_bytes(3) == 3 * 96 + 16 and c.ddrr_siddon_ncc_workspace_bytes(0) == 0
    assert c.ddrr_pose_raygen_forward(P, P, 2, 2, 1, P, P, P, 1, 4, P, P, P, P, None, L(0), None, None) != 0
    assert b"Euler" in c.ddrr_last_error()
    assert c.ddrr_pose_raygen_forward(P, P, 2, 0, 1, P, P, P, 1, 4, P, P, P, P, None, L(8), None, None) != 0
    assert b"clear_floats" in c.ddrr_last_error()
    assert c.ddrr_pose_raygen_forward(P, P, 2, 0, 1, P, P, P, 0, 4, P, P, P, P, P, L(8), None, None) != 0
    assert b"empty batch" in c.ddrr_last_error()
    assert c.ddrr_siddon_ncc_forward(P, P, P, L(3), 1, 4, f(1e-5), P, P, P, None, None, None) != 0
    assert b"x1_stride" in c.ddrr_last_error()
    assert c.ddrr_siddon_ncc_forward(P, P, P, L(0), 1, 4, f(1e-5), None, P, P, None, None, None) != 0
    assert b"null" in c.ddrr_last_error()
    assert c.ddrr_siddon_ncc_backward_pose(P, P, P, L(0), P, P, 2, P, P, P, P, P, P, P, 2, 0, 1, P, 1, 4,
                                           f(1e-8), 1, P, P, P, None) != 0
    assert b"g_stride" in c.ddrr_last_error()
    assert c.ddrr_siddon_ncc_forward(P, P, P, L(0), 0, 4, f(1e-5), P, P, P, None, None, None) == 0  # empty batch
    # the fused pose Adam step (ABI 30)
    assert c.ddrr_pose_adam_step(P, P, P, P, P, P, P, P, P, None, 1, f(0.1), f(5.0), f(0.9), f(0.999), f(1e-8), 1,
                                 None) != 0
    assert b"null" in c.ddrr_last_error()
    assert c.ddrr_pose_adam_step(P, P, P, P, P, P, P, P, P, P, 1, f(0.1), f(5.0), f(1.0), f(0.999), f(1e-8), 1,
                                 None) != 0
    assert b"Adam" in c.ddrr_last_error()
    assert c.ddrr_pose_adam_step(P, P, P, P, P, P, P, P, P, P, 0, f(0.1), f(5.0), f(0.9), f(0.999), f(1e-8), 1,
                                 None) == 0  # no poses
    # the record alone: out may be NULL only together with aux
    assert c.ddrr_siddon_forward_bricks(P, 4, 4, 4, P, P, P, 1, 4, 5, f(0.5), f(1e-8), None, None,
                                        f(0.0), 0, None, 0, P, None) != 0
    assert b"out" in c.ddrr_last_error()


def test_brick_kernels_fit_their_register_budget():
    """The brick kernels run 1024-thread workgroups = 4 waves per SIMD, i.e. 128 vector
    registers per lane; a change that makes the compiler spill (scratch memory in the walk)
    costs 10-50 % and is silent.  Read the kernel descriptors of the built code objects: no
    private segment, at most 128 VGPRs, for every instance of siddon_brick_kernel."""
    import struct

    import __graft_entry__ as entry

    entry.build_hip()
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not available")
    data = open(_lib.LIB_PATH, "rb").read()
    kernels = {}
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        off = m.start()
        n = struct.unpack_from("<Q", data, off + 24)[0]
        p = off + 32
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" not in triple:
                continue
            path = os.path.join(ROOT, "tests", "emu", "_co.elf")
            with open(path, "wb") as f:
                f.write(data[off + o:off + o + size])
            notes = subprocess.run([readelf, "--notes", path], capture_output=True, text=True).stdout
            os.remove(path)
            name, scratch = None, None
            for line in notes.splitlines():  # kernel-level keys come in alphabetical order
                m2 = re.match(r"\s+\.(name|private_segment_fixed_size|vgpr_count):\s+(\S+)", line)
                if not m2:
                    continue
                key, val = m2.groups()
                if key == "name" and val.startswith("_Z"):
                    name, scratch = val, None
                elif key == "private_segment_fixed_size":
                    scratch = int(val)
                elif key == "vgpr_count" and name is not None and scratch is not None:
                    kernels[name] = (scratch, int(val))
                    name = None
    bricks = {k: v for k, v in kernels.items()
              if "siddon_brick_kernel" in k or "siddon_fwd_brick_kernel" in k}
    assert len(bricks) >= 6, sorted(kernels)
    for name, (scratch, vgpr) in bricks.items():
        assert scratch == 0, (name, scratch)
        assert vgpr <= 128, (name, vgpr)
