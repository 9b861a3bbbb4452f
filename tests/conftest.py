"""Shared test plumbing.

Markers: ``gpu`` = needs a real MI355X (run by the driver with ``-m gpu``);
everything else runs in the GPU-less build container.

Helpers here are TEST infrastructure: the golden-fixture loader, the host
emulation build of the kernel cores (tests/emu) and a monkeypatch that routes
``diffdrr_amd.ops`` launches to it so that the autograd wiring can be checked
without a GPU.  None of this is reachable from the product package.
"""
from __future__ import annotations

import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_SRC = os.path.join(ROOT, "tests", "emu", "ddrr_emu.cpp")
EMU_SO = os.path.join(ROOT, "tests", "emu", "_build", "libddrr_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    """max |a - b| / max |b|: the image-normalised error of SURVEY.md section 8(d)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def build_emu():
    csrc = os.path.join(ROOT, "diffdrr_amd", "csrc")
    deps = [EMU_SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] \
        + [os.path.join(ROOT, "include", "diffdrr_hip.h")]
    if os.path.exists(EMU_SO) and all(os.path.getmtime(d) <= os.path.getmtime(EMU_SO)
                                      for d in deps):
        return EMU_SO
    os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
    subprocess.run(
        ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-mfma", "-ffp-contract=off",
         "-Wno-unknown-pragmas", EMU_SRC, "-o", EMU_SO], check=True)
    return EMU_SO


@pytest.fixture(scope="session")
def emu_lib():
    from diffdrr_amd._lib import DdrrLibrary

    return DdrrLibrary(build_emu())


@pytest.fixture()
def emulated_ops(emu_lib, monkeypatch):
    """Route diffdrr_amd.ops launches to the host emulation of the kernel cores
    (same source as the HIP kernels, compiled for the CPU) so that the Python /
    autograd layer can be exercised on CPU tensors."""
    from diffdrr_amd import ops

    monkeypatch.setattr(ops, "_require_gpu", lambda volume: None)
    monkeypatch.setattr(ops, "on_device", lambda t: True)
    monkeypatch.setattr(ops, "_launch", lambda name, device, *a: emu_lib.call(name, *a, None))
    return ops


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from diffdrr_amd import _lib

    _lib.get_lib()  # fail loudly if the HIP library was not built
    return torch.device("cuda:0")
