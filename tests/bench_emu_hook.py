"""Test-only hook for `bench.py --device cpu` (DDRR_BENCH_HOOK=bench_emu_hook): routes
diffdrr_amd.ops launches to the host emulation of the kernel cores, so that the bench harness's
multi-rank logic (self-spawn, sharding, all_gather, max-over-ranks timing, the JSON contract)
can run on gloo ranks in the GPU-less container.  Never used by a measurement."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from conftest import build_emu  # noqa: E402
from diffdrr_amd import ops  # noqa: E402
from diffdrr_amd._lib import DdrrLibrary  # noqa: E402

_emu = DdrrLibrary(build_emu())
ops._require_gpu = lambda volume: None
ops.on_device = lambda t: True
ops._launch = lambda name, device, *a: _emu.call(name, *a, None)
