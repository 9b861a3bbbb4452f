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

# ---------------------------------------------------------------- 2026-09-27T01:16:35Z  round 5 evidence pass: GPU tests, smoke, default bench line, configs 2-5 full, rocprofv3 stats + PMC traffic, issue-bound counters
OUT=gpurun_out/r05z; mkdir -p $OUT; (timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/gpu_tests.txt; tail -2 $OUT/gpu_tests.txt; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $OUT/smoke.txt; tail -3 $OUT/smoke.txt; timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err; grep "\[bench\]" $OUT/bench_headline.err | cut -c1-170; for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/c$c.err; grep "config $c:" $OUT/c$c.err | cut -c1-140; done; timeout 900 bash tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -16 $OUT/rocprof_bench.txt | cut -c1-150; cp $OUT/prof/traffic.json $OUT/traffic.json 2>/dev/null; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprof_bench_kernel_stats.csv; cp $OUT/prof/bench_line_under_trace.json $OUT/ 2>/dev/null; rm -rf $OUT/prof; timeout 900 python tools/issue_bound.py $OUT > $OUT/issue_bound.log 2>&1; tail -3 $OUT/issue_bound.log | cut -c1-200; ls $OUT

# ---------------------------------------------------------------- 2026-09-27T01:27:58Z  default bench line with configs.few_poses and 4-pose parity
mkdir -p gpurun_out/r05y; (time timeout 900 python bench.py > gpurun_out/r05y/bench_config_headline.json 2> gpurun_out/r05y/bench_headline.err) 2>&1 | tail -3; grep "few poses\|config 4:\|config headline:" gpurun_out/r05y/bench_headline.err | cut -c1-200; python -c "
import json; d=json.load(open(\"gpurun_out/r05y/bench_config_headline.json\")); print(d[\"value\"], d[\"ms_per_step\"], d[\"parity\"][\"poses\"], d[\"parity\"][\"fwd_rel_err_vs_fp64\"], d[\"parity\"][\"pose_grad_rel_err_vs_fp64\"], d[\"configs\"][\"4\"][\"value\"], d[\"configs\"][\"3\"][\"value\"], d[\"roofline\"][\"forward\"].get(\"traffic\"), d[\"roofline\"][\"forward_f32\"].get(\"traffic\"))"

# ---------------------------------------------------------------- 2026-09-27T01:30:19Z  driver-like bench invocations (N = 1: plain and under torch.distributed.run)
mkdir -p gpurun_out/r05x; timeout 500 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r05x/driver_like.json 2> gpurun_out/r05x/driver_like.err; python -c "
import json; d=json.load(open(\"gpurun_out/r05x/driver_like.json\")); print(d[\"value\"], d[\"ms_per_step\"], d[\"steps\"], d[\"warmup\"], sorted(d.keys())); print(sorted(d[\"configs\"].keys()), d[\"roofline\"][\"frac\"], d[\"cpu_baseline\"][\"value\"])"; timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-configs > gpurun_out/r05x/driver_like_tdr.json 2> gpurun_out/r05x/driver_like_tdr.err; python -c "
import json; d=json.load(open(\"gpurun_out/r05x/driver_like_tdr.json\")); print(d[\"value\"], d[\"ms_per_step\"], d[\"n_gpus\"])"

# ---------------------------------------------------------------- 2026-09-27T01:32:52Z  new test: the 32-pose headline launch vs the oracle for every pose
mkdir -p gpurun_out/r05w; (time timeout 1200 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -s -k "32_poses_vs_oracle" 2>&1 | grep -v amdgpu.ids | tail -8) 2>&1 | tee gpurun_out/r05w/headline_32_poses_vs_oracle.txt

# ---------------------------------------------------------------- 2026-09-27T01:35:08Z  32-pose oracle test: per-pose error listing
timeout 1200 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -s -k "32_poses_vs_oracle" 2>&1 | grep "^pose\|passed\|failed\|headline launch" | cut -c1-330 | tee gpurun_out/r05w/per_pose.txt

# ---------------------------------------------------------------- 2026-09-27T01:36:40Z  32-pose oracle test, restated gates
(timeout 1200 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -s -k "32_poses_vs_oracle" 2>&1 | grep "passed\|failed\|headline launch\|Error" | cut -c1-400) | tee gpurun_out/r05w/headline_32_poses_vs_oracle.txt

# ---------------------------------------------------------------- 2026-09-27T01:39:15Z  new test: CT-like volume vs oracle
timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "ct_like" 2>&1 | tail -12

# ---------------------------------------------------------------- 2026-09-27T01:41:14Z  randomised sweep of the brick kernels on the final tree
mkdir -p gpurun_out/r05v; (timeout 600 python tools/fuzz_bricks.py --cases 96 --seed 5; timeout 400 python tools/fuzz_bricks.py --cases 32 --seed 6 --smooth) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05v/fuzz_bricks.txt; tail -12 gpurun_out/r05v/fuzz_bricks.txt | cut -c1-220

# ---------------------------------------------------------------- 2026-09-27T01:44:24Z  final tree: full GPU suite + smoke
mkdir -p gpurun_out/r05final; (timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > gpurun_out/r05final/gpu_tests.txt; cat gpurun_out/r05final/gpu_tests.txt; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3

# ---------------------------------------------------------------- 2026-09-27T01:51:44Z  empty launches: the staging loop alone
mkdir -p gpurun_out/r05u; timeout 600 python tools/empty_launch_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05u/empty_launch_probe.txt

# ---------------------------------------------------------------- 2026-09-27T02:05:51Z  few-pose kernel: first run (storage tests + timing)
timeout 500 python -m pytest tests/test_gpu_brick_storage.py -x -q -k "look_ahead or any_depth or q16_storage or non_finite" 2>&1 | tail -25 > gpurun_out/few_tests.txt; timeout 300 python tools/storage_bench.py --scenes noise512,phantom512 --poses 1,2,4 --storages q16p > gpurun_out/few_bench.txt 2>&1; tail -5 gpurun_out/few_tests.txt; cat gpurun_out/few_bench.txt

# ---------------------------------------------------------------- 2026-09-27T02:07:16Z  few-pose kernel: static parts
timeout 300 python tools/storage_bench.py --scenes noise512,phantom512 --poses 1,2,4 --storages q16p > gpurun_out/few_bench.txt 2>&1; cat gpurun_out/few_bench.txt; timeout 200 python tools/empty_launch_probe.py 2>&1 | grep -v "^B   8\|^B  32" | tee gpurun_out/few_probe.txt

# ---------------------------------------------------------------- 2026-09-27T02:08:29Z  product kernel at 1/2/4 poses for the parts experiment's table
timeout 300 python tools/storage_bench.py --scenes noise512,phantom512 --poses 1,2,4 --storages q16p 2>&1 | tee gpurun_out/few_bench_product.txt

# ---------------------------------------------------------------- 2026-09-27T02:11:05Z  claim-ahead in the general brick kernel: config 3 + full gpu suite
python bench.py --config 3 --no-cpu-baseline > gpurun_out/c3_ahead.json 2> gpurun_out/c3_ahead.err; python - <<EOF
import json
d=json.load(open("gpurun_out/c3_ahead.json"))
print(d["value"], d["ms_per_step"], [(k["kernel"], round(k["kernel_ms"],4)) for k in d.get("kernels",[])], d.get("b4",{}).get("value"))
EOF
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/gpu_tests_ahead.txt

# ---------------------------------------------------------------- 2026-09-27T02:15:13Z  phase profile of the marcher's brick kernels at config 3
timeout 600 python tools/tri_profile.py 2>&1 | tee gpurun_out/tri_profile.txt

# ---------------------------------------------------------------- 2026-09-27T02:18:56Z  phase profile of the marcher's brick kernels at config 3
timeout 600 python tools/tri_profile.py 2>&1 | tee gpurun_out/tri_profile.txt

# ---------------------------------------------------------------- 2026-09-27T02:22:09Z  marcher: ray parts at <= 2 poses: trilinear tests + config 3
timeout 600 python -m pytest tests -m gpu -x -q -k "tri or march or Tri" 2>&1 | tail -4; python bench.py --config 3 --no-cpu-baseline > gpurun_out/c3_parts.json 2> gpurun_out/c3_parts.err; python - <<EOF
import json
d=json.load(open("gpurun_out/c3_parts.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], json.dumps(d.get("parity")))
EOF
python tools/trilinear_bench.py 2>&1 | tail -6

# ---------------------------------------------------------------- 2026-09-27T02:24:21Z  stage stamps of the marcher's brick kernels
timeout 600 python tools/tri_stamps.py 2>&1 | tee gpurun_out/tri_stamps.txt; timeout 300 python tools/tri_stamps.py 4 2>&1 | tee -a gpurun_out/tri_stamps.txt

# ---------------------------------------------------------------- 2026-09-27T02:25:35Z  volume gradient: what the LDS atomics cost
timeout 600 python tools/scratch/volgrad_noatomic.py 2>&1 | tee gpurun_out/volgrad_noatomic.txt

# ---------------------------------------------------------------- 2026-09-27T02:29:52Z  marcher volume gradient: branch-free owner scatter
timeout 600 python -m pytest tests -m gpu -x -q -k "tri or march or Tri or volume" 2>&1 | tail -3; python tools/trilinear_bench.py 2>&1 | grep "volume-grad"

# ---------------------------------------------------------------- 2026-09-27T02:34:25Z  marcher volume gradient: hoisted accumulator choice, prescale, branch-free lin01
timeout 700 python -m pytest tests -m gpu -x -q -k "tri or march or Tri or volume or channel" 2>&1 | tail -3; python tools/trilinear_bench.py 2>&1 | grep "volume-grad\|forward+record"

# ---------------------------------------------------------------- 2026-09-27T02:40:41Z  volume-gradient walks with the hoisted accumulators: full gpu suite + Siddon volgrad + config 3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/gpu_tests_now.txt; python tools/volgrad_bench.py 2>&1 | tail -4; python bench.py --config 3 --no-cpu-baseline > gpurun_out/c3_new.json 2>/dev/null; python - <<EOF
import json
d=json.load(open("gpurun_out/c3_new.json"))
print("config 3:", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])
EOF

# ---------------------------------------------------------------- 2026-09-27T02:43:38Z  marcher forward: 8 quads in flight while staging + one membership region
timeout 500 python -m pytest tests -m gpu -x -q -k "tri or march or Tri" 2>&1 | tail -2; python tools/trilinear_bench.py 2>&1 | grep "volume-grad"

# ---------------------------------------------------------------- 2026-09-27T02:45:35Z  marcher: halo bricks staged with 16-byte loads
timeout 500 python -m pytest tests -m gpu -x -q -k "tri or march or Tri or depth" 2>&1 | tail -2; python tools/trilinear_bench.py 2>&1 | grep "volume-grad\|forward+record"

# ---------------------------------------------------------------- 2026-09-27T02:48:19Z  pooled end in the general brick kernel: A/B
timeout 700 python tools/pool_ab.py 2>&1 | tee gpurun_out/pool_ab.txt

# ---------------------------------------------------------------- 2026-09-27T02:49:11Z  stage stamps of the marcher's kernels after the round's changes
timeout 600 python tools/tri_stamps.py 2>&1 | tee gpurun_out/tri_stamps2.txt

# ---------------------------------------------------------------- 2026-09-27T02:50:32Z  marcher forward: eight 16-byte quads in flight
python tools/trilinear_bench.py 2>&1 | grep "volume-grad"

# ---------------------------------------------------------------- 2026-09-27T02:51:25Z  stage stamps of the channel render
timeout 500 python tools/channel_stamps.py 8 2>&1 | tee gpurun_out/channel_stamps.txt; timeout 300 python tools/channel_stamps.py 1 2>&1 | tee -a gpurun_out/channel_stamps.txt

# ---------------------------------------------------------------- 2026-09-27T02:53:18Z  general brick kernel: detector models cached in LDS
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; python tools/trilinear_bench.py 2>&1 | grep "volume-grad"; python tools/channels_bench.py --real-mask 2>&1 | tail -5

# ---------------------------------------------------------------- 2026-09-27T02:55:05Z  channels bench, real mask (detector models cached)
python tools/channels_bench.py --real-mask 2>&1 | grep -v "^trilinear" | tee gpurun_out/channels_now.txt

# ---------------------------------------------------------------- 2026-09-27T03:00:28Z  channels bench, real mask (detector models cached)
python tools/channels_bench.py --real-mask 2>&1 | grep -v "^trilinear" | tee gpurun_out/channels_now.txt

# ---------------------------------------------------------------- 2026-09-27T03:06:14Z  marcher: approx reciprocal for the sample-run bounds, flush by multiplication
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; python tools/trilinear_bench.py 2>&1 | grep "volume-grad\|forward+record"; python tools/volgrad_bench.py | tail -4

# ---------------------------------------------------------------- 2026-09-27T03:08:49Z  round 5 evidence pass 2 (after the marcher / volume-gradient work): GPU tests, smoke, default bench line, configs 2-5, rocprofv3 stats + PMC traffic, issue-bound counters
OUT=gpurun_out/r05z2; mkdir -p $OUT; (timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/gpu_tests.txt; tail -2 $OUT/gpu_tests.txt; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $OUT/smoke.txt; tail -3 $OUT/smoke.txt; timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err; grep "\[bench\]" $OUT/bench_headline.err | cut -c1-170; for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/c$c.err; grep "config $c:" $OUT/c$c.err | cut -c1-140; done; timeout 900 bash tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -16 $OUT/rocprof_bench.txt | cut -c1-150; cp $OUT/prof/traffic.json $OUT/traffic.json 2>/dev/null; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprof_bench_kernel_stats.csv; cp $OUT/prof/bench_line_under_trace.json $OUT/ 2>/dev/null; rm -rf $OUT/prof; timeout 900 python tools/issue_bound.py $OUT > $OUT/issue_bound.log 2>&1; tail -3 $OUT/issue_bound.log | cut -c1-200; ls $OUT

# ---------------------------------------------------------------- 2026-09-27T03:14:55Z  randomised sweep of every brick kernel on the tree with the new volume-gradient loops
mkdir -p gpurun_out/r05w; (timeout 600 python tools/fuzz_bricks.py --cases 96 --seed 7; timeout 400 python tools/fuzz_bricks.py --cases 32 --seed 8 --smooth) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05w/fuzz_bricks.txt; tail -14 gpurun_out/r05w/fuzz_bricks.txt | cut -c1-260

# ---------------------------------------------------------------- 2026-09-27T03:18:39Z  guard constant 12: full gpu suite, few-pose timing on the phantom, config 4
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6; timeout 300 python tools/storage_bench.py --scenes phantom512,ct --poses 1,8,32 --storages q16p 2>&1 | grep -v amdgpu; timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/c4_r12.json 2> gpurun_out/c4_r12.err; python - <<EOF
import json
d=json.load(open("gpurun_out/c4_r12.json")); print("config 4:", d["value"], d["ms_per_step"], d["config"].get("brick_storage_fallbacks"), d.get("registration"))
EOF

# ---------------------------------------------------------------- 2026-09-27T03:21:08Z  round 5 evidence pass 3 (guard at 12x, marcher work): GPU tests, smoke, default bench line, configs 2-5, rocprofv3 stats + PMC traffic, issue-bound counters
OUT=gpurun_out/r05z3; mkdir -p $OUT; (timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/gpu_tests.txt; tail -2 $OUT/gpu_tests.txt; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $OUT/smoke.txt; tail -3 $OUT/smoke.txt; timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err; grep "\[bench\]" $OUT/bench_headline.err | cut -c1-170; for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/c$c.err; grep "config $c:" $OUT/c$c.err | cut -c1-140; done; timeout 900 bash tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -6 $OUT/rocprof_bench.txt | cut -c1-150; cp $OUT/prof/traffic.json $OUT/traffic.json 2>/dev/null; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprof_bench_kernel_stats.csv; cp $OUT/prof/bench_line_under_trace.json $OUT/ 2>/dev/null; rm -rf $OUT/prof; ls $OUT | wc -l

# ---------------------------------------------------------------- 2026-09-27T03:27:24Z  timeline of a registration iteration inside its graph
cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace -d /tmp/g4 -o x --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/graph_timeline.py /tmp/g4 | tee gpurun_out/graph_timeline.txt

# ---------------------------------------------------------------- 2026-09-27T03:34:00Z  one clear launch + PoseAdam: tests, config 4, timeline
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3; for f in "" "--torch-adam"; do timeout 300 python bench.py --config 4 --no-cpu-baseline $f > gpurun_out/c4$f.json 2> gpurun_out/c4$f.err; python - <<EOF
import json
d=json.load(open("gpurun_out/c4$f.json")); print("config 4 $f:", d["value"], d["ms_per_step"], d.get("registration"))
EOF
done; cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace -d /tmp/g4 -o x --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/graph_timeline.py /tmp/g4 | tee gpurun_out/graph_timeline2.txt; timeout 200 python tools/storage_bench.py --scenes noise512 --poses 1,2,8,32 --storages q16p 2>&1 | grep -v amdgpu

# ---------------------------------------------------------------- 2026-09-27T03:39:20Z  registration step without its clear launch: tests, config 4, timeline
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3; timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/c4.json 2> gpurun_out/c4.err; python - <<EOF
import json
d=json.load(open("gpurun_out/c4.json")); print("config 4:", d["value"], d["ms_per_step"], d.get("registration"))
EOF
cd /tmp && rm -rf /tmp/g4 && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace -d /tmp/g4 -o x --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/graph_timeline.py /tmp/g4 | tee gpurun_out/graph_timeline3.txt

# ---------------------------------------------------------------- 2026-09-27T03:42:04Z  headline step: fused step at 32 poses again, after the epilogue work
for f in "" "--fused-max-poses 64"; do timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 300 $f > gpurun_out/h.json 2> gpurun_out/h.err; python - <<EOF
import json
d=json.load(open("gpurun_out/h.json")); print("headline [$f]:", round(d["value"]), d["ms_per_step"], d["roofline"]["kernel_ms"], [(k["kernel"].replace("ddrr_",""), round(k["kernel_ms"]*1e3,1)) for k in d.get("kernels",[])[1:8]])
EOF
done; for f in "" "--fused-max-poses 64"; do timeout 300 python bench.py --config 2 --no-cpu-baseline $f > gpurun_out/h.json 2> gpurun_out/h.err; python - <<EOF
import json
d=json.load(open("gpurun_out/h.json")); print("config 2 [$f]:", round(d["value"]), d["ms_per_step"], d["roofline"]["kernel_ms"])
EOF
done

# ---------------------------------------------------------------- 2026-09-27T03:43:32Z  round 5 evidence pass 4 (fused step to 32 poses, PoseAdam, clears merged): GPU tests, smoke, default bench line, configs 2-5, rocprofv3 stats + PMC traffic, issue-bound counters
OUT=gpurun_out/r05z4; mkdir -p $OUT; (timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/gpu_tests.txt; tail -2 $OUT/gpu_tests.txt; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $OUT/smoke.txt; tail -3 $OUT/smoke.txt; timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err; grep "\[bench\]" $OUT/bench_headline.err | cut -c1-170 | head -12; for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/c$c.err; grep "config $c:" $OUT/c$c.err | cut -c1-140; done; timeout 900 bash tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -14 $OUT/rocprof_bench.txt | cut -c1-150; cp $OUT/prof/traffic.json $OUT/traffic.json 2>/dev/null; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rocprof_bench_kernel_stats.csv; cp $OUT/prof/bench_line_under_trace.json $OUT/ 2>/dev/null; rm -rf $OUT/prof; ls $OUT | wc -l

# ---------------------------------------------------------------- 2026-09-27T03:48:50Z  PoseAdam in the graphed loop vs torch Adam eager
timeout 400 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "graphed or adam" 2>&1 | tail -6

# ---------------------------------------------------------------- 2026-09-27T03:49:11Z  PoseAdam graphed test: see the failure
timeout 400 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "pose_adam" 2>&1 | grep -E "^E|assert" | head -12

# ---------------------------------------------------------------- 2026-09-27T03:49:40Z  PoseAdam graphed test again
timeout 400 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "pose_adam or graphed" 2>&1 | tail -4

# ---------------------------------------------------------------- 2026-09-27T03:50:27Z  PoseAdam graphed test: failure details
timeout 400 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "pose_adam" 2>&1 | grep -E "^E  " | head -8 | cut -c1-400

# ---------------------------------------------------------------- 2026-09-27T03:50:48Z  PoseAdam graphed test: failure details
timeout 400 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "pose_adam" 2>&1 | tail -30 | cut -c1-300

# ---------------------------------------------------------------- 2026-09-27T03:51:13Z  PoseAdam graphed test with its neighbour
for i in 1 2 3; do timeout 400 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "pose_adam or graphed" 2>&1 | grep -E "^E   |passed|failed" | head -6 | cut -c1-420; done

# ---------------------------------------------------------------- 2026-09-27T03:51:52Z  PoseAdam graphed test, six iterations, five times
for i in 1 2 3 4 5; do timeout 400 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "pose_adam or graphed" 2>&1 | grep -E "^E   |passed|failed" | head -4 | cut -c1-300; done

# ---------------------------------------------------------------- 2026-09-27T03:52:33Z  PoseAdam graphed test, eight times
for i in 1 2 3 4 5 6 7 8; do timeout 400 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -k "pose_adam or graphed" 2>&1 | grep -E "^E   |passed|failed" | head -3 | cut -c1-300; done

# ---------------------------------------------------------------- 2026-09-27T03:55:29Z  final tree: full GPU suite + smoke
mkdir -p gpurun_out/r05final; (timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > gpurun_out/r05final/gpu_tests.txt; cat gpurun_out/r05final/gpu_tests.txt; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05final/smoke.txt | tail -3

# ---------------------------------------------------------------- 2026-09-27T03:57:19Z  module call cost at one pose
timeout 400 python tools/module_call_bench.py 2>&1 | grep -v amdgpu

# ---------------------------------------------------------------- 2026-09-27T03:57:48Z  module call cost at one pose (order check)
timeout 400 python tools/module_call_bench.py 2>&1 | grep -v amdgpu

# ---------------------------------------------------------------- 2026-09-27T03:58:17Z  module call cost (for the record)
timeout 400 python tools/module_call_bench.py 2>&1 | grep -v amdgpu > gpurun_out/module_call.txt; cat gpurun_out/module_call.txt

# ---------------------------------------------------------------- 2026-09-27T04:00:21Z  marcher volume gradient: bricks handed out heaviest first
python tools/trilinear_bench.py 2>&1 | grep "volume-grad"

# ---------------------------------------------------------------- 2026-09-27T04:00:46Z  marcher volume gradient ordered: tests
timeout 700 python -m pytest tests -m gpu -x -q -k "tri or march or Tri or volume" 2>&1 | tail -2; python bench.py --config 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d[\"value\"], d[\"ms_per_step\"])"

# ---------------------------------------------------------------- 2026-09-27T04:02:12Z  final tree: default bench line + config 3 + GPU suite + smoke
OUT=gpurun_out/r05z5; mkdir -p $OUT; (timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt; tail -1 $OUT/gpu_tests.txt; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $OUT/smoke.txt; tail -1 $OUT/smoke.txt; timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err; grep "\[bench\] config headline:\|config 4:\|config 3:" $OUT/bench_headline.err | cut -c1-150; timeout 600 python bench.py --config 3 > $OUT/bench_config_3.json 2> $OUT/c3.err; grep "config 3:" $OUT/c3.err | cut -c1-140

# ---------------------------------------------------------------- 2026-09-27T04:06:53Z  final tree: extended randomised sweep (256 + 64 smooth cases, new seeds)
mkdir -p gpurun_out/r05x; (timeout 900 python tools/fuzz_bricks.py --cases 256 --seed 9; timeout 600 python tools/fuzz_bricks.py --cases 64 --seed 10 --smooth) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05x/fuzz_bricks.txt; grep -n "worst" gpurun_out/r05x/fuzz_bricks.txt | cut -c1-400; grep -c "<<<" gpurun_out/r05x/fuzz_bricks.txt

# ---------------------------------------------------------------- 2026-09-27T04:10:41Z  inference path: GPU test + module call cost
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; timeout 300 python tools/module_call_bench.py 2>&1 | grep -v amdgpu

# ---------------------------------------------------------------- 2026-09-27T04:12:32Z  inference path without the cache miss: module call cost
timeout 300 python tools/module_call_bench.py 2>&1 | grep -v amdgpu; timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "inference" 2>&1 | tail -1

# ---------------------------------------------------------------- 2026-09-27T04:13:03Z  module call cost, longer warm-up
timeout 300 python tools/module_call_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/module_call2.txt

# ---------------------------------------------------------------- 2026-09-27T04:13:54Z  final tree: GPU suite + smoke
mkdir -p gpurun_out/r05final2; (timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > gpurun_out/r05final2/gpu_tests.txt; cat gpurun_out/r05final2/gpu_tests.txt | tail -2; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05final2/smoke.txt | tail -1

# ---------------------------------------------------------------- 2026-09-27T04:15:39Z  final tree: default bench + config 5 (inference path in the sweep)
timeout 900 python bench.py > gpurun_out/bh.json 2> gpurun_out/bh.err; echo rc=$?; grep "\[bench\] config headline:\|config 5:\|config 4:" gpurun_out/bh.err | cut -c1-160; timeout 600 python bench.py --config 5 > gpurun_out/b5.json 2> gpurun_out/b5.err; echo rc=$?; python - <<EOF
import json
d=json.load(open("gpurun_out/bh.json")); print(d["value"], d["ms_per_step"], d["sweep"]["value"], d["configs"]["5"]["value"], d["configs"]["4"]["value"], d["parity"]["fwd_rel_err_vs_fp64"])
x=json.load(open("gpurun_out/b5.json")); print(x["value"], x["ms_per_step"], [(k["kernel"], round(k["kernel_ms"],3), k["launches_per_step"]) for k in x.get("kernels",[])], x.get("parity"))
EOF
