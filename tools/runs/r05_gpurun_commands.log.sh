#!/bin/bash
# DISPOSABLE LOG, not source: the command lists of every gpurun call of round 5, in order (tools/runs/README.md)

# ---------------------------------------------------------------- 2026-09-27T00:12:38Z  round 5 first pass: GPU tests, default bench line (forward_f32, configs 4 / ct / 3.b4), trilinear bench for the alpha-range kernel
mkdir -p gpurun_out/r05a; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05a/gpu_tests.txt; tail -4 gpurun_out/r05a/gpu_tests.txt; timeout 600 python bench.py > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err; tail -30 gpurun_out/r05a/bench.err; python -c "
import json; d=json.load(open(\"gpurun_out/r05a/bench.json\")); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"roofline\"][\"forward\"][\"frac\"], d[\"roofline\"][\"forward_f32\"][\"frac\"]); print(json.dumps(d[\"configs\"][\"ct\"], indent=1)[:3000]); print(d[\"configs\"][\"4\"][\"value\"], d[\"configs\"][\"3\"][\"value\"], d[\"configs\"][\"3\"][\"b4\"][\"value\"]); print([(k[\"kernel\"], round(k[\"kernel_ms\"],4)) for k in d[\"configs\"][\"3\"][\"kernels\"]])"

# ---------------------------------------------------------------- 2026-09-27T00:26:14Z  f32p storage: affected GPU tests + storage_bench on 4 scenes
mkdir -p gpurun_out/r05b; timeout 900 python -m pytest tests/test_gpu_brick_storage.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -8 > gpurun_out/r05b/gpu_tests_subset.txt; tail -4 gpurun_out/r05b/gpu_tests_subset.txt; timeout 600 python tools/storage_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05b/storage_bench.txt; cat gpurun_out/r05b/storage_bench.txt

# ---------------------------------------------------------------- 2026-09-27T00:31:04Z  static few-pose schedule: brick tests, storage_bench q16p 1/2/4/8 poses, config 4
mkdir -p gpurun_out/r05c; timeout 900 python -m pytest tests/test_gpu_brick_storage.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -5 > gpurun_out/r05c/gpu_tests_subset.txt; tail -3 gpurun_out/r05c/gpu_tests_subset.txt; timeout 600 python tools/storage_bench.py --storages q16p --poses 1,2,4,8 --scenes noise512,phantom512,ct 2>&1 | grep -v amdgpu.ids > gpurun_out/r05c/storage_bench.txt; cat gpurun_out/r05c/storage_bench.txt; timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r05c/bench_config_4.json 2> gpurun_out/r05c/c4.err; grep "config 4:" gpurun_out/r05c/c4.err | cut -c1-220; python -c "
import json; d=json.load(open(\"gpurun_out/r05c/bench_config_4.json\")); print(d[\"value\"], d[\"ms_per_step\"], d[\"registration\"])"

# ---------------------------------------------------------------- 2026-09-27T00:33:07Z  one-pose stage stamps, trace-only build
mkdir -p gpurun_out/r05d; DDRR_EXP_FLAGS=-DDDRR_TRACE_ONLY timeout 600 python tools/brick_times.py --cases pert1,pert1aux,pert8aux --variant 5 --storage q16p 2>&1 | grep -v amdgpu.ids > gpurun_out/r05d/brick_times_trace_only.txt; cat gpurun_out/r05d/brick_times_trace_only.txt

# ---------------------------------------------------------------- 2026-09-27T00:35:35Z  look-ahead launches: detector models once per launch in LDS, look-ahead published behind the staging barrier
mkdir -p gpurun_out/r05e; timeout 900 python -m pytest tests/test_gpu_brick_storage.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -5 > gpurun_out/r05e/gpu_tests_subset.txt; tail -3 gpurun_out/r05e/gpu_tests_subset.txt; timeout 600 python tools/storage_bench.py --storages q16p --poses 1,2,8,32 --scenes noise512,phantom512,ct 2>&1 | grep -v amdgpu.ids > gpurun_out/r05e/storage_bench.txt; cat gpurun_out/r05e/storage_bench.txt; DDRR_EXP_FLAGS=-DDDRR_TRACE_ONLY timeout 600 python tools/brick_times.py --cases pert1,pert1aux --variant 5 --storage q16p 2>&1 | grep -v amdgpu.ids | grep -A2 "product" > gpurun_out/r05e/brick_times_trace_only.txt; cat gpurun_out/r05e/brick_times_trace_only.txt

# ---------------------------------------------------------------- 2026-09-27T00:39:04Z  one-pose stage stamps, finer (image stored / rows written)
mkdir -p gpurun_out/r05f; DDRR_EXP_FLAGS=-DDDRR_TRACE_ONLY timeout 600 python tools/brick_times.py --cases pert1,pert1aux --variant 5 --storage q16p 2>&1 | grep -v amdgpu.ids | grep -A2 "product" > gpurun_out/r05f/brick_times_trace_only.txt; cat gpurun_out/r05f/brick_times_trace_only.txt

# ---------------------------------------------------------------- 2026-09-27T00:44:06Z  lane-parallel rows + lazy image store (one-pose path)
mkdir -p gpurun_out/r05g; timeout 900 python -m pytest tests/test_gpu_brick_storage.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -5 > gpurun_out/r05g/gpu_tests_subset.txt; tail -3 gpurun_out/r05g/gpu_tests_subset.txt; timeout 600 python tools/storage_bench.py --storages q16p --poses 1,2,4,8,32 --scenes noise512,phantom512,ct 2>&1 | grep -v amdgpu.ids > gpurun_out/r05g/storage_bench.txt; cat gpurun_out/r05g/storage_bench.txt; DDRR_EXP_FLAGS=-DDDRR_TRACE_ONLY timeout 600 python tools/brick_times.py --cases pert1,pert1aux --variant 5 --storage q16p 2>&1 | grep -v amdgpu.ids | grep -A2 "product" > gpurun_out/r05g/brick_times_trace_only.txt; cat gpurun_out/r05g/brick_times_trace_only.txt

# ---------------------------------------------------------------- 2026-09-27T00:45:36Z  config 5 test failure details
mkdir -p gpurun_out/r05h; timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k config5 2>&1 | tail -40 > gpurun_out/r05h/t.txt; cat gpurun_out/r05h/t.txt

# ---------------------------------------------------------------- 2026-09-27T00:49:28Z  few poses: rows-ready flag instead of the staging barrier, lane-parallel rows, wave 0 free
mkdir -p gpurun_out/r05i; timeout 900 python -m pytest tests/test_gpu_brick_storage.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -5 > gpurun_out/r05i/gpu_tests_subset.txt; tail -3 gpurun_out/r05i/gpu_tests_subset.txt; timeout 600 python tools/storage_bench.py --storages q16p --poses 1,2,4,8,32 --scenes noise512,phantom512,ct 2>&1 | grep -v amdgpu.ids > gpurun_out/r05i/storage_bench.txt; cat gpurun_out/r05i/storage_bench.txt; DDRR_EXP_FLAGS=-DDDRR_TRACE_ONLY timeout 600 python tools/brick_times.py --cases pert1,pert1aux --variant 5 --storage q16p 2>&1 | grep -v amdgpu.ids | grep -A2 "product" > gpurun_out/r05i/brick_times_trace_only.txt; cat gpurun_out/r05i/brick_times_trace_only.txt

# ---------------------------------------------------------------- 2026-09-27T00:54:36Z  256^3: bricks handed out in pose parts (split) on fp32 bricks, order weight
mkdir -p gpurun_out/r05j; timeout 600 python tools/brick_bench.py --size 256 --variants 0 --storage f32 --order weight --cases pert32,pert32aux --split 0:1,512:2,512:4,256:2,128:2 2>&1 | grep -v amdgpu.ids | cut -c1-230 > gpurun_out/r05j/split_256.txt; cat gpurun_out/r05j/split_256.txt

# ---------------------------------------------------------------- 2026-09-27T00:55:57Z  forward + record with the record computed but not delivered (bound for delivery through LDS)
mkdir -p gpurun_out/r05k; timeout 600 python tools/brick_bench.py --variants=-2 --storage q16p --order weight --cases pert32aux,pert8aux,pert1aux --dbg 0,131072 2>&1 | grep -v amdgpu.ids | cut -c1-230 > gpurun_out/r05k/no_delivery.txt; cat gpurun_out/r05k/no_delivery.txt

# ---------------------------------------------------------------- 2026-09-27T01:04:57Z  fused registration step: GPU tests (all), bench headline fused vs unfused, config 4, config 2
mkdir -p gpurun_out/r05l; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05l/gpu_tests.txt; tail -4 gpurun_out/r05l/gpu_tests.txt; for f in "" "--unfused"; do timeout 300 python bench.py --no-configs --no-cpu-baseline $f > gpurun_out/r05l/bench_headline$f.json 2> gpurun_out/r05l/h$f.err; grep "config headline:" gpurun_out/r05l/h$f.err | cut -c1-150; done; timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r05l/bench_config_4.json 2> gpurun_out/r05l/c4.err; grep "config 4:" gpurun_out/r05l/c4.err | cut -c1-150; for f in "" "--unfused"; do timeout 300 python bench.py --config 2 --no-cpu-baseline $f > gpurun_out/r05l/bench_config_2$f.json 2> gpurun_out/r05l/c2$f.err; grep "config 2:" gpurun_out/r05l/c2$f.err | cut -c1-150; done; python -c "
import json
for n in (\"bench_headline\",\"bench_headline--unfused\",\"bench_config_4\",\"bench_config_2\",\"bench_config_2--unfused\"):
    d=json.load(open(\"gpurun_out/r05l/\"+n+\".json\")); print(n, round(d[\"value\"],1), round(d[\"ms_per_step\"],4), round(d[\"roofline\"][\"kernel_ms\"],4), [(k[\"kernel\"][5:], round(k[\"kernel_ms\"]*1e3,1)) for k in d[\"roofline\"][\"kernels\"][1:]])"

# ---------------------------------------------------------------- 2026-09-27T01:07:04Z  a5 GPU test failure details
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k filter_intersections 2>&1 | grep -E "assert|Error|rel_err" | head -20

# ---------------------------------------------------------------- 2026-09-27T01:09:34Z  fused step v2 (no agent fences, 1024 rays per workgroup, matrix via LDS): tests + bench fused vs unfused
mkdir -p gpurun_out/r05m; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05m/gpu_tests.txt; tail -3 gpurun_out/r05m/gpu_tests.txt; for f in "" "--unfused"; do timeout 300 python bench.py --no-configs --no-cpu-baseline $f > gpurun_out/r05m/bench_headline$f.json 2> gpurun_out/r05m/h$f.err; done; timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r05m/bench_config_4.json 2> gpurun_out/r05m/c4.err; for f in "" "--unfused"; do timeout 300 python bench.py --config 2 --no-cpu-baseline $f > gpurun_out/r05m/bench_config_2$f.json 2> gpurun_out/r05m/c2$f.err; done; python -c "
import json
for n in (\"bench_headline\",\"bench_headline--unfused\",\"bench_config_4\",\"bench_config_2\",\"bench_config_2--unfused\"):
    d=json.load(open(\"gpurun_out/r05m/\"+n+\".json\")); print(n, round(d[\"value\"],1), round(d[\"ms_per_step\"],4), round(d[\"roofline\"][\"kernel_ms\"],4), [(k[\"kernel\"][5:], round(k[\"kernel_ms\"]*1e3,1)) for k in d[\"roofline\"][\"kernels\"][1:]])"

# ---------------------------------------------------------------- 2026-09-27T01:12:59Z  fused vs unfused step by poses per call (bench --batch B)
mkdir -p gpurun_out/r05n; for B in 1 2 4 8 16; do for f in "--fused-max-poses 64" "--unfused"; do timeout 200 python bench.py --no-configs --no-cpu-baseline --batch $B --steps 300 $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(\"B\", d[\"config\"][\"batch_per_gpu\"], \"\", \"ms_per_step %.4f\" % d[\"ms_per_step\"], \"kernel %.4f\" % d[\"roofline\"][\"kernel_ms\"], [(k[\"kernel\"][5:], round(k[\"kernel_ms\"]*1e3,1)) for k in d[\"roofline\"][\"kernels\"][1:]])"; done; done 2>&1 | tee gpurun_out/r05n/fused_by_batch.txt
