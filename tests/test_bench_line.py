"""The line `bench.py` prints for the driver stays small enough to be parsed (round 5's grew to
26.6 KB and came back `parsed: null`): `bench.compact_line` is held to its byte limit on the real
full records committed under profiles/ and on a record with every section at once, and no key
called `frac` exceeds 1 (a quotient that can is called `work_rate`)."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline")
FULL_RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]", "bench_config_headline.json")) +
                      glob.glob(os.path.join(ROOT, "profiles", "r0[6-9]", "bench_full.json")))


def _fracs(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            if k in ("frac", "forward_frac") and isinstance(v, (int, float)):
                yield path + "." + k, v
            else:
                yield from _fracs(v, path + "." + k)
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _fracs(v, f"{path}[{i}]")


def _check(line):
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT < 8192
    back = json.loads(text)
    for k in CONTRACT:
        assert k in back, k
    rf = back["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms",
              "algorithmic_bytes_per_launch", "units_per_launch"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "workload" in back["config"] and all(len(str(v)) <= 200 for v in back["config"].values())
    for where, v in _fracs(back):
        assert 0.0 <= v <= 1.0, (where, v)
    return back


@pytest.mark.parametrize("path", FULL_RECORDS, ids=[os.path.relpath(p, ROOT) for p in FULL_RECORDS])
def test_committed_full_records_compact_to_a_parseable_line(path):
    with open(path) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 8192  # (the record that did not parse, or as large)
    back = _check(bench.compact_line(full))
    assert "cpu_baseline" in back and back["cpu_baseline"]["cores"] >= 1
    assert set(back["parity"]) >= {"fwd_rel_err", "fwd_rel_err_vs_fp64", "pose_grad_rel_err_vs_fp64"}
    assert "dropped_for_size" not in back  # (today's sections fit without dropping any)
    for cfg in ("2", "3", "4", "5"):
        c = back["configs"][cfg]
        assert c["value"] > 0 and c["ms_per_step"] > 0 and c["kernel_ms"] > 0


def test_line_is_held_to_its_limit_whatever_the_record_holds():
    """Sections are dropped, least important first, before the limit is crossed: a record with
    absurdly long strings and forty configs still yields a parseable line with the contract keys."""
    with open(FULL_RECORDS[0]) as f:
        full = json.load(f)
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["roofline"]["algorithmic_bytes_per_unit"] = "u" * 5000
    for i in range(40):
        full["configs"][f"x{i}"] = dict(full["configs"]["2"])
    back = _check(bench.compact_line(full))
    assert "configs" in back["dropped_for_size"] and "cpu_baseline" in back and "parity" in back


def test_floats_of_the_line_keep_five_digits():
    assert bench._sig(21818.639216118456) == 21819.0 and bench._sig(8.792982407612726e-05) == 8.793e-05
    assert bench._sig(True) is True and bench._sig(7) == 7 and bench._sig(float("nan")) is None


def test_more_ranks_than_devices_is_one_json_error_line_and_rc_2():
    """`bench.py --gpus N` with fewer than N devices visible: one JSON line and rc 2 before anything
    is spawned (this container has no GPU at all), not N torchrun tracebacks."""
    import subprocess

    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("needs fewer than 64 devices")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 2, res.stderr[-2000:]
    lines = res.stdout.strip().splitlines()
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert "error" in out and out["n_gpus_requested"] == 64 and out["n_gpus_visible"] == torch.cuda.device_count()


def test_a_failing_side_run_is_recorded_not_fatal():
    """`configs.*` of the default line are side runs: on one GPU a failure is recorded in its place (and
    the line keeps its size and its contract keys); with several ranks the exception propagates."""
    def boom():
        raise RuntimeError("no such kernel " + "x" * 1000)

    out = bench.guarded_run(1, "ct", boom)
    assert out["error"].startswith("RuntimeError: no such kernel") and len(out["error"]) <= 300 and out["wall_s"] >= 0
    assert bench.guarded_run(1, "ok", lambda: {"value": 1.0})["value"] == 1.0
    with pytest.raises(RuntimeError):
        bench.guarded_run(2, "ct", boom)
    with open(FULL_RECORDS[-1]) as f:
        full = json.load(f)
    full["configs"]["ct"] = out
    full["configs"]["3"] = bench.guarded_run(1, "3", boom)
    back = _check(bench.compact_line(full))
    assert back["configs"]["ct"]["error"].startswith("RuntimeError") and "error" in back["configs"]["3"]
