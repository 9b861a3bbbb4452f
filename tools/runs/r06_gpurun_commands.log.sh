#!/bin/bash
# DISPOSABLE LOG, not source: the command lists of every gpurun call of round 6, in order (tools/runs/README.md)

# ---------------------------------------------------------------- 2026-10-01T03:33:56Z  r06 first: new gpu tests + bench line parses
mkdir -p gpurun_out/r06a; python -m pytest tests -m gpu -x -q -k "guarded_16bit or fused_ncc or pose_adam or PoseAdam or graph" 2>&1 | tail -15 > gpurun_out/r06a/tests_new.txt; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a/bench_stdout.txt 2> gpurun_out/r06a/bench_stderr.txt; echo rc=$? >> gpurun_out/r06a/tests_new.txt; cp bench_full.json gpurun_out/r06a/; wc -c gpurun_out/r06a/bench_stdout.txt; cat gpurun_out/r06a/tests_new.txt; python bench.py --gpus 4 --steps 2 > gpurun_out/r06a/preflight.txt 2>&1; echo rc=$?; cat gpurun_out/r06a/preflight.txt | tail -3

# ---------------------------------------------------------------- 2026-10-01T03:35:27Z  r06: is the 3.29 ms headline kernel reproducible?
mkdir -p gpurun_out/r06b; for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>&1 >/dev/null | grep "config headline:" ; done; python bench.py --steps 400 --warmup 10 --no-configs --no-cpu-baseline 2>&1 >/dev/null | grep "config headline:"; git -C . log -1 --format=%h 2>/dev/null

# ---------------------------------------------------------------- 2026-10-01T03:36:08Z  r06 diag: event creation cost; bench per-step
python tools/_diag_events.py; python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>&1 >/dev/null | grep "config headline:"; python bench.py --steps 20 --warmup 30 --no-configs --no-cpu-baseline 2>&1 >/dev/null | grep "config headline:"; python bench.py --steps 100 --warmup 5 --no-configs --no-cpu-baseline 2>&1 >/dev/null | grep "config headline:"; python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --unfused 2>&1 >/dev/null | grep "config headline:"

# ---------------------------------------------------------------- 2026-10-01T03:43:58Z  r06: sparse levers parity on GPU + bench
mkdir -p gpurun_out/r06c; python -m pytest tests -m gpu -x -q -k "subsample or guarded_16bit or patches" 2>&1 | tail -15 > gpurun_out/r06c/tests.txt; cat gpurun_out/r06c/tests.txt; python tools/sparse_levers_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c/sparse.txt

# ---------------------------------------------------------------- 2026-10-01T03:51:26Z  r06: full gpu suite + bench driver command
mkdir -p gpurun_out/r06d; python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r06d/gpu_suite.txt; tail -5 gpurun_out/r06d/gpu_suite.txt; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06d/bench_stdout.txt 2> gpurun_out/r06d/bench_stderr.txt; echo bench rc=$?; cp bench_full.json gpurun_out/r06d/; grep -v "full record" gpurun_out/r06d/bench_stderr.txt | grep "config headline\|sparse\|few poses"

# ---------------------------------------------------------------- 2026-10-01T03:54:10Z  r06: ncc grad probe 128 phantom
python tools/ncc_grad_probe.py 128 128 2.4 phantom 31 0 8 16 2>&1 | grep -v amdgpu.ids

# ---------------------------------------------------------------- 2026-10-01T03:54:48Z  r06: ncc oracle chain + untracked edits + sparse tests
python -m pytest tests -m gpu -x -q -s -k "ncc_vs_fp64 or untracked or subsample or PoseAdam or pose_adam" 2>&1 | grep -v "^$" | tail -30

# ---------------------------------------------------------------- 2026-10-01T03:57:18Z  r06: bench driver command after gc/clock fix + fingerprint cost
mkdir -p gpurun_out/r06e; for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>&1 >/dev/null | grep "config headline:"; done; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06e/bench_stdout.txt 2> gpurun_out/r06e/bench_stderr.txt; echo bench rc=$?; cp bench_full.json gpurun_out/r06e/; grep -v "full record" gpurun_out/r06e/bench_stderr.txt | grep "config headline\|sparse\|few poses\|config ct\|config 4"; python -m pytest tests -m gpu -x -q -k "ncc_vs_fp64 or untracked or subsample" 2>&1 | tail -3

# ---------------------------------------------------------------- 2026-10-01T04:01:55Z  r06: fingerprint hidden behind the first claim: brick tests + bench
python -m pytest tests -m gpu -x -q -k "brick or look_ahead or untracked or headline or storage" 2>&1 | tail -3; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -v "full record" | grep "config headline\|few poses\|config ct\|config 4:"

# ---------------------------------------------------------------- 2026-10-01T04:04:19Z  r06: A/B fingerprint placement
for r in 1 2; do for v in A_preclaim B_before_loop; do cp tools/_build/ab/lib$v.so diffdrr_amd/csrc/libdiffdrr_hip.so; echo == $v; python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --no-configs 2>&1 >/dev/null | grep -v "full record" | grep "config headline" | sed "s/| backward.*//"; done; done

# ---------------------------------------------------------------- 2026-10-01T04:05:36Z  r06: PRE instantiation: brick tests + bench legs
python -m pytest tests -m gpu -x -q -k "brick or look_ahead or untracked or headline or storage or registration or graph" 2>&1 | tail -3; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -v "full record" | grep "config headline\|few poses\|config ct\|config 4:"

# ---------------------------------------------------------------- 2026-10-01T04:12:58Z  r06: epilogues v2: tests + bench + kernel trace
mkdir -p gpurun_out/r06f; python -m pytest tests -m gpu -x -q -k "ncc or registration or graph or trajectory or adam" 2>&1 | tail -3; python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-configs 2>&1 >/dev/null | grep -v "full record" | grep "config headline:"; python - <<PY
import json
d=json.load(open("bench_full.json"))
for k in d["roofline"]["kernels"]: print(k["kernel"], round(k["kernel_ms"],4), k["launches_per_step"])
print("step-kernel", d["ms_per_step"]-d["roofline"]["kernel_ms"])
PY
python bench.py --gpus 1 --config 2 --steps 200 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -v "full record" | grep "config 2:"; python - <<PY
import json
d=json.load(open("bench_full.json"))
for k in d["roofline"]["kernels"]: print(k["kernel"], round(k["kernel_ms"],4), k["launches_per_step"])
print("step-kernel", d["ms_per_step"]-d["roofline"]["kernel_ms"])
PY
python bench.py --gpus 1 --config 4 --no-cpu-baseline 2>&1 >/dev/null | grep -v "full record" | grep "config 4:"

# ---------------------------------------------------------------- 2026-10-01T04:14:54Z  r06: bwd epilogue 2048 rays per workgroup
python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-configs 2>&1 >/dev/null | grep -v "full record" | grep "config headline:" | cut -c1-120; python - <<PY
import json
d=json.load(open("bench_full.json"))
for k in d["roofline"]["kernels"]: print(k["kernel"], round(k["kernel_ms"],4), k["launches_per_step"])
print("step-kernel", d["ms_per_step"]-d["roofline"]["kernel_ms"])
PY
python -m pytest tests -m gpu -x -q -k "fused_ncc" 2>&1 | tail -2

# ---------------------------------------------------------------- 2026-10-01T04:15:18Z  r06: config 4 / 2 with the 2048-ray backward epilogue
for c in 4 4 2; do python bench.py --gpus 1 --config $c --no-cpu-baseline 2>&1 >/dev/null | grep -v "full record" | grep "config $c:" | cut -c1-150; done

# ---------------------------------------------------------------- 2026-10-01T04:16:38Z  r06: epilogues by launch size: tests + configs
python -m pytest tests -m gpu -x -q -k "ncc or registration or graph or trajectory or adam" 2>&1 | tail -2; for c in 4 4 2; do python bench.py --gpus 1 --config $c --no-cpu-baseline 2>&1 >/dev/null | grep -v "full record" | grep "config $c:" | cut -c1-150; done; python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-configs 2>&1 >/dev/null | grep -v "full record" | grep "config headline:" | cut -c1-120; python - <<PY
import json
d=json.load(open("bench_full.json"))
for k in d["roofline"]["kernels"]: print(k["kernel"], round(k["kernel_ms"],4), k["launches_per_step"])
print("step-kernel", d["ms_per_step"]-d["roofline"]["kernel_ms"])
PY

# ---------------------------------------------------------------- 2026-10-01T04:24:36Z  r06: storage table for thin volumes
python tools/storage_table.py 2>&1 | grep -v amdgpu.ids

# ---------------------------------------------------------------- 2026-10-01T04:25:23Z  r06: storage table, the band around 8 poses
python tools/storage_table.py 5 6 7 8 9 10 12 14 2>&1 | grep -v amdgpu.ids

# ---------------------------------------------------------------- 2026-10-01T04:26:43Z  r06: fp32 bricks general vs configurable kernel by pose count
python tools/f32_kernel_table.py 2>&1 | grep -v amdgpu.ids

# ---------------------------------------------------------------- 2026-10-01T04:32:54Z  r06: patch NCC kernels: tests + bench vs composition
python -m pytest tests -m gpu -x -q -k "patch_ncc or metrics" 2>&1 | tail -3; python tools/patch_ncc_bench.py 2>&1 | grep -v amdgpu.ids

# ---------------------------------------------------------------- 2026-10-01T04:38:07Z  r06: patch NCC kernels: tests + bench vs composition (retry)
python -m pytest tests -m gpu -x -q -k "patch_ncc or metrics" 2>&1 | tail -3; python tools/patch_ncc_bench.py 2>&1 | grep -v amdgpu.ids

# ---------------------------------------------------------------- 2026-10-01T04:45:39Z  r06: A/B mask branch at few poses
bash tools/_build/ab/ab.sh 2>&1 | grep "==\|few poses\|config ct" | cut -c1-200

# ---------------------------------------------------------------- 2026-10-01T04:47:50Z  r06: SUB instantiation: tests + bench legs
python -m pytest tests -m gpu -x -q -k "subsample or patches or brick or headline" 2>&1 | tail -2; python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -v "full record" | grep "config headline\|sparse\|few poses\|config ct\|config 4:" | cut -c1-200

# ---------------------------------------------------------------- 2026-10-01T04:51:46Z  r06: full GPU suite with printouts
mkdir -p gpurun_out/r06g; python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > gpurun_out/r06g/gpu_suite_full.txt; tail -3 gpurun_out/r06g/gpu_suite_full.txt; grep -c "^\[" gpurun_out/r06g/gpu_suite_full.txt; python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06g/smoke.txt 2>&1; tail -3 gpurun_out/r06g/smoke.txt

# ---------------------------------------------------------------- 2026-10-01T04:55:02Z  r06: evidence: rocprof + PMC of the bench, driver command, configs, tools
mkdir -p gpurun_out/r06h; tools/prof_bench.sh gpurun_out/r06prof > gpurun_out/r06h/rocprof_bench.txt 2>&1; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06h/bench_line_driver_command.json 2> gpurun_out/r06h/bench_stderr_driver_command.txt; cp bench_full.json gpurun_out/r06h/bench_full_driver_command.json; python bench.py > gpurun_out/r06h/bench_line_default.json 2> gpurun_out/r06h/bench_stderr_default.txt; cp bench_full.json gpurun_out/r06h/bench_full.json; for c in 2 3 4 5; do python bench.py --config $c > gpurun_out/r06h/bench_line_config_$c.json 2>/dev/null; cp bench_full.json gpurun_out/r06h/bench_config_$c.json; done; python tools/sparse_levers_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06h/sparse.txt; python tools/patch_ncc_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06h/patch_ncc.txt; python tools/fuzz_bricks.py --cases 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r06h/fuzz_bricks.txt; tail -2 gpurun_out/r06h/fuzz_bricks.txt | cut -c1-300; head -20 gpurun_out/r06h/rocprof_bench.txt | cut -c1-200; wc -c gpurun_out/r06h/bench_line_*.json

# ---------------------------------------------------------------- 2026-10-01T04:59:13Z  r06: multiscale 32-pose slowness probe
python tools/_ms_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200

# ---------------------------------------------------------------- 2026-10-01T05:05:29Z  r06: registration loop with multiscale / gradient NCC in the graph
for c in ncc multiscale gradient; do echo == $c; python bench.py --config 4 --criterion $c --no-cpu-baseline 2>&1 >/dev/null | grep -v "full record" | grep "config 4" | cut -c1-120; python -c "
import json; d=json.load(open(\"bench_full.json\")); print(d[\"value\"], d[\"registration\"])"; done

# ---------------------------------------------------------------- 2026-10-01T05:09:21Z  r06: channel render vs plain on the final tree
python tools/channels_fwd_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-220

# ---------------------------------------------------------------- 2026-10-01T05:13:11Z  r06: masked channel flush: tests + channels bench
python -m pytest tests -m gpu -x -q -k "channel or mask" 2>&1 | tail -2; python tools/channels_fwd_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-90; python tools/fuzz_bricks.py --cases 24 2>&1 | tail -1 | cut -c1-300

# ---------------------------------------------------------------- 2026-10-01T05:17:57Z  r06: channel words experiment
python tools/channels_fwd_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-110

# ---------------------------------------------------------------- 2026-10-01T05:22:19Z  r06: channel words with the check: tests + bench
python -m pytest tests -m gpu -x -q -k "channel or ready_packed or mask" 2>&1 | tail -2; python tools/channels_fwd_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-140

# ---------------------------------------------------------------- 2026-10-01T05:27:20Z  r06: channel words with the check: tests + bench (retry)
python -m pytest tests -m gpu -x -q -k "channel or ready_packed or mask" 2>&1 | tail -2; python tools/channels_fwd_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-170

# ---------------------------------------------------------------- 2026-10-01T05:30:12Z  r06: channel words, two-launch check
python -m pytest tests -m gpu -x -q -k "ready_packed" 2>&1 | tail -2; python tools/channels_fwd_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-130

# ---------------------------------------------------------------- 2026-10-01T05:33:41Z  r06: FINAL evidence: suite, smoke, rocprof + PMC, bench lines, configs, tools
mkdir -p gpurun_out/r06z; python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > gpurun_out/r06z/gpu_suite_full.txt; tail -1 gpurun_out/r06z/gpu_suite_full.txt; python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06z/smoke.txt 2>&1; tail -1 gpurun_out/r06z/smoke.txt; tools/prof_bench.sh gpurun_out/r06zprof > gpurun_out/r06z/rocprof_bench.txt 2>&1; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06z/bench_line_driver_command.json 2> gpurun_out/r06z/bench_stderr_driver_command.txt; echo rc=$?; python bench.py > gpurun_out/r06z/bench_line_default.json 2> gpurun_out/r06z/bench_stderr_default.txt; cp bench_full.json gpurun_out/r06z/bench_full.json; for c in 2 3 4 5; do python bench.py --config $c > gpurun_out/r06z/bench_line_config_$c.json 2>/dev/null; cp bench_full.json gpurun_out/r06z/bench_config_$c.json; done; python tools/sparse_levers_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06z/sparse.txt; python tools/patch_ncc_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06z/patch_ncc.txt; python tools/fuzz_bricks.py --cases 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r06z/fuzz_bricks.txt; tail -1 gpurun_out/r06z/fuzz_bricks.txt | cut -c1-300; wc -c gpurun_out/r06z/bench_line_*.json; head -9 gpurun_out/r06z/rocprof_bench.txt | cut -c1-150

# ---------------------------------------------------------------- 2026-10-01T05:43:14Z  r06: last check of the final commit: suite + smoke + driver bench
python -m pytest tests -m gpu -x -q 2>&1 | tail -2; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
l = sys.stdin.read().strip().splitlines()
print(len(l), len(l[-1])); d = json.loads(l[-1]); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"cpu_baseline\"][\"value\"], list(d[\"configs\"]))"

# ---------------------------------------------------------------- 2026-10-01T05:47:45Z  r06: robustness: fuzz 160 cases seed 7, GPU suite twice more
mkdir -p gpurun_out/r06y; python tools/fuzz_bricks.py --cases 160 --seed 7 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y/fuzz_bricks_seed7.txt; tail -1 gpurun_out/r06y/fuzz_bricks_seed7.txt | cut -c1-320; python tools/fuzz_bricks.py --cases 60 --seed 3 --smooth 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y/fuzz_bricks_smooth.txt; tail -1 gpurun_out/r06y/fuzz_bricks_smooth.txt | cut -c1-320; for i in 1 2; do python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done

# ---------------------------------------------------------------- 2026-10-01T05:54:59Z  r06: robustness: fuzz 160 cases seed 7, smooth 60, GPU suite twice more (retry)
mkdir -p gpurun_out/r06y; python tools/fuzz_bricks.py --cases 160 --seed 7 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y/fuzz_bricks_seed7.txt; tail -1 gpurun_out/r06y/fuzz_bricks_seed7.txt | cut -c1-320; python tools/fuzz_bricks.py --cases 60 --seed 3 --smooth 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y/fuzz_bricks_smooth.txt; tail -1 gpurun_out/r06y/fuzz_bricks_smooth.txt | cut -c1-320; for i in 1 2; do python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done

# ---------------------------------------------------------------- 2026-10-01T06:02:41Z  r06: r05's fuzz script (same random stream) on the current library
python tools/_fuzz_r05.py --cases 64 --seed 0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y/fuzz_r05_script.txt; tail -1 gpurun_out/r06y/fuzz_r05_script.txt | cut -c1-200

# ---------------------------------------------------------------- 2026-10-01T06:04:13Z  r06: r05's fuzz script on the current library (retry with dir)
mkdir -p gpurun_out/r06y; python tools/_fuzz_r05.py --cases 64 --seed 0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y/fuzz_r05_script.txt; tail -1 gpurun_out/r06y/fuzz_r05_script.txt | cut -c1-200

# ---------------------------------------------------------------- 2026-10-01T06:05:02Z  r06: A/B fuzz: round 5's tree against the current one, same script, same seed
mkdir -p gpurun_out/r06y; (cd r05tree && python tools/fuzz_bricks.py --cases 64 --seed 0 2>&1 | grep -v amdgpu.ids > ../gpurun_out/r06y/fuzz_r05_tree.txt); python tools/_fuzz_r05.py --cases 64 --seed 0 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y/fuzz_r05_script.txt; tail -1 gpurun_out/r06y/fuzz_r05_tree.txt | cut -c1-160; tail -1 gpurun_out/r06y/fuzz_r05_script.txt | cut -c1-160

# ---------------------------------------------------------------- 2026-10-01T06:07:25Z  r06: bench of the final bench.py
mkdir -p gpurun_out/r06x; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06x/bench_line_driver_command.json 2> gpurun_out/r06x/stderr.txt; echo rc=$?; cp bench_full.json gpurun_out/r06x/; python -c "
import json; l=open(\"gpurun_out/r06x/bench_line_driver_command.json\").read().strip().splitlines(); print(len(l), len(l[-1])); d=json.loads(l[-1]); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"roofline\"][\"step_minus_kernel_ms\"], [k for k,v in d[\"configs\"].items() if \"error\" in v])"

# ---------------------------------------------------------------- 2026-10-01T07:01:49Z  r06: evidence refresh on the final tree (ABI 32)
mkdir -p gpurun_out/r06y; python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > gpurun_out/r06y/gpu_suite_full.txt; tail -1 gpurun_out/r06y/gpu_suite_full.txt; python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06y/smoke.txt 2>&1; tail -1 gpurun_out/r06y/smoke.txt; tools/prof_bench.sh gpurun_out/r06yprof > gpurun_out/r06y/rocprof_bench.txt 2>&1; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06y/bench_line_driver_command.json 2> gpurun_out/r06y/bench_stderr_driver_command.txt; echo rc=$?; cp bench_full.json gpurun_out/r06y/bench_full_driver_command.json; python bench.py > gpurun_out/r06y/bench_line_default.json 2> gpurun_out/r06y/bench_stderr_default.txt; cp bench_full.json gpurun_out/r06y/bench_full.json; for c in 2 3 4 5; do python bench.py --config $c > gpurun_out/r06y/bench_line_config_$c.json 2>/dev/null; cp bench_full.json gpurun_out/r06y/bench_config_$c.json; done; for c in ncc multiscale gradient gradient_patch9; do python bench.py --config 4 --criterion $c 2>/dev/null | tail -1 > gpurun_out/r06y/c4_$c.json; done; python tools/patch_ncc_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06y/patch_ncc.txt; wc -c gpurun_out/r06y/bench_line_*.json; head -9 gpurun_out/r06y/rocprof_bench.txt | cut -c1-150; cp gpurun_out/r06yprof/traffic.json gpurun_out/r06y/ 2>/dev/null; find gpurun_out/r06yprof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06y/rocprof_bench_kernel_stats.csv \;; cp gpurun_out/r06yprof/bench_line_under_trace.json gpurun_out/r06y/

# ---------------------------------------------------------------- 2026-10-01T07:19:16Z  r06: full suite + bench after the one-node render (ABI 33)
mkdir -p gpurun_out/r06w; python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > gpurun_out/r06w/gpu_suite_full.txt; tail -1 gpurun_out/r06w/gpu_suite_full.txt; python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06w/smoke.txt 2>&1; tail -1 gpurun_out/r06w/smoke.txt; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06w/bench_line_driver_command.json 2> gpurun_out/r06w/bench_stderr_driver_command.txt; echo rc=$?; cp bench_full.json gpurun_out/r06w/bench_full_driver_command.json; for c in ncc multiscale gradient gradient_patch9; do python bench.py --config 4 --criterion $c 2>/dev/null | tail -1 > gpurun_out/r06w/c4_$c.json; done; python bench.py --unfused --no-configs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06w/bench_unfused.json
