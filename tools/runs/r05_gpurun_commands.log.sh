#!/bin/bash
# DISPOSABLE LOG, not source: the command lists of every gpurun call of round 5, in order (tools/runs/README.md)

# ---------------------------------------------------------------- 2026-09-27T00:12:38Z  round 5 first pass: GPU tests, default bench line (forward_f32, configs 4 / ct / 3.b4), trilinear bench for the alpha-range kernel
mkdir -p gpurun_out/r05a; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05a/gpu_tests.txt; tail -4 gpurun_out/r05a/gpu_tests.txt; timeout 600 python bench.py > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err; tail -30 gpurun_out/r05a/bench.err; python -c "
import json; d=json.load(open(\"gpurun_out/r05a/bench.json\")); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"roofline\"][\"forward\"][\"frac\"], d[\"roofline\"][\"forward_f32\"][\"frac\"]); print(json.dumps(d[\"configs\"][\"ct\"], indent=1)[:3000]); print(d[\"configs\"][\"4\"][\"value\"], d[\"configs\"][\"3\"][\"value\"], d[\"configs\"][\"3\"][\"b4\"][\"value\"]); print([(k[\"kernel\"], round(k[\"kernel_ms\"],4)) for k in d[\"configs\"][\"3\"][\"kernels\"]])"

# ---------------------------------------------------------------- 2026-09-27T00:26:14Z  f32p storage: affected GPU tests + storage_bench on 4 scenes
mkdir -p gpurun_out/r05b; timeout 900 python -m pytest tests/test_gpu_brick_storage.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -8 > gpurun_out/r05b/gpu_tests_subset.txt; tail -4 gpurun_out/r05b/gpu_tests_subset.txt; timeout 600 python tools/storage_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05b/storage_bench.txt; cat gpurun_out/r05b/storage_bench.txt
