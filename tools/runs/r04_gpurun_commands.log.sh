#!/bin/bash
# DISPOSABLE LOG, not source: the command lists of every gpurun call of round 4, in order, as they
# were run (hard-coded shapes and output directories; nothing imports or executes this file).
# The summaries they produced are under profiles/r04/.

# ---------------------------------------------------------------- r04_run1.sh
# round 4, first GPU pass: the guarded 16-bit bricks + per-launch workspaces on the device
OUT=gpurun_out/r04a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_brick_storage.py -x -q 2>&1 | tail -15 > $OUT/guard_tests.txt; cat $OUT/guard_tests.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 600 python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -c 3000 $OUT/bench_headline.json

# ---------------------------------------------------------------- r04_run2.sh
# round 4: the default bench line with configs / sweep / forward_sweep; the launch-state stress test
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_brick_storage.py -x -q -k "streams or empty" 2>&1 | tail -5 > $OUT/stress.txt; cat $OUT/stress.txt
( time timeout 900 python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err ) 2>&1 | tail -4
grep "\[bench\]" $OUT/bench_headline.err; tail -5 $OUT/bench_headline.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r04b/bench_headline.json"))
print("value", r["value"], "ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "fwd", r["roofline"]["forward"]["frac"])
print("forward_sweep", {k:v for k,v in r["roofline"].get("forward_sweep",{}).items() if k in("kernel_ms","frac","parity")})
print("sweep", r.get("sweep"))
for k,c in r.get("configs",{}).items():
    print(k, c["value"], c["unit"], c["ms_per_step"], c["dominant_kernel"], c["parity"], c["wall_s"])
print(r["config"])
PY

# ---------------------------------------------------------------- r04_run3.sh
# round 4: why the empty-batch test failed on the device; counters for the issue_bound blocks
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_brick_storage.py -x -q -k "empty" 2>&1 | tail -40 > $OUT/empty.txt; cat $OUT/empty.txt
timeout 1500 python tools/issue_bound.py gpurun_out/r04c 2>&1 | grep -v amdgpu.ids | tail -20

# ---------------------------------------------------------------- r04_run4.sh
# round 4: the forward-only walk with accumulated chord-relative alphas (16 instead of 19 instructions per step)
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.txt; tail -25 $OUT/gpu_tests.txt
timeout 600 python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; grep "\[bench\]" $OUT/bench_headline.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r04d/bench_headline.json"))
print("value", r["value"], "ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "fwd", r["roofline"]["forward"]["frac"], r["roofline"]["forward"]["kernel_ms"])
print("parity", {k:v for k,v in r["parity"].items() if isinstance(v,(int,float))})
print("forward_sweep", {k:v for k,v in r["roofline"].get("forward_sweep",{}).items() if k in("kernel_ms","frac","parity")})
PY
# the record kernel: grouping of 8 / aligned rows on and off (tools build)
timeout 600 python tools/brick_bench.py --cases pert32aux,pert32 --dbg 0,8,1024,1032 2>&1 | grep -v amdgpu.ids > $OUT/record_grouping.txt; cat $OUT/record_grouping.txt

# ---------------------------------------------------------------- r04_run5.sh
# round 4: can better brick weights close the gap between the launch and sum / 256 ?
OUT=gpurun_out/r04e; mkdir -p $OUT
timeout 900 python tools/brick_weights.py 2>&1 | grep -v amdgpu.ids > $OUT/brick_weights.txt; cat $OUT/brick_weights.txt

# ---------------------------------------------------------------- r04_run6.sh
# round 4: channel walk on the accumulating scheme + lean flush; the general kernel's forward on the accumulating walk
OUT=gpurun_out/r04f; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.txt; tail -8 $OUT/gpu_tests.txt
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; cat $OUT/channels.txt

# ---------------------------------------------------------------- r04_run7.sh
# round 4: channel candidates grouped in runs of 8 adjacent pixels (coherent label changes)
OUT=gpurun_out/r04g; mkdir -p $OUT
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; cat $OUT/channels.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.txt; tail -8 $OUT/gpu_tests.txt

# ---------------------------------------------------------------- r04_run8.sh
# round 4: where does the channel render's time go (phase profile, plain vs channels, same kernel)
OUT=gpurun_out/r04h; mkdir -p $OUT
(timeout 600 python tools/channels_profile.py --real-mask --poses 8; timeout 600 python tools/channels_profile.py --real-mask --poses 1) 2>&1 | grep -v amdgpu.ids > $OUT/channels_profile.txt; cat $OUT/channels_profile.txt

# ---------------------------------------------------------------- r04_run9.sh
# round 4: labels staged with aligned dword loads on the unaligned path
OUT=gpurun_out/r04i; mkdir -p $OUT
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; cat $OUT/channels.txt
timeout 600 python tools/channels_profile.py --real-mask --poses 8 2>&1 | grep -v amdgpu.ids > $OUT/channels_profile.txt; cat $OUT/channels_profile.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.txt; tail -5 $OUT/gpu_tests.txt

# ---------------------------------------------------------------- r04_run10.sh
# round 4: what the fp32 fallback of the 16-bit bricks costs; channel backward timings
OUT=gpurun_out/r04j; mkdir -p $OUT
timeout 900 python tools/guard_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/guard_bench.txt; cat $OUT/guard_bench.txt
timeout 600 python tools/channels_bench.py --real-mask 2>&1 | grep -v amdgpu.ids > $OUT/channels_real.txt; cat $OUT/channels_real.txt

# ---------------------------------------------------------------- r04_run11.sh
# round 4: the channel render's ray backward on the bricks
OUT=gpurun_out/r04k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_brick_storage.py -x -q -k "channel" 2>&1 | tail -15 > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "backward\|entry points\|^#" $OUT/channels.txt

# ---------------------------------------------------------------- r04_run12.sh
# round 4: evidence pass -- full GPU tests, smoke, every bench config, rocprofv3 stats + PMC traffic, issue-bound counters
OUT=gpurun_out/r04l; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err; grep "\[bench\]" $OUT/bench_headline.err | head -3
for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_$c.err; grep "\[bench\] config $c:" $OUT/bench_$c.err; done
timeout 900 tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -14 $OUT/rocprof_bench.txt
timeout 1500 python tools/issue_bound.py $OUT 2>&1 | grep -v amdgpu.ids | tail -3
ls $OUT

# ---------------------------------------------------------------- r04_run13.sh
# round 4: the sorted walk (hits of a round counting-sorted by length class through global scratch) against the queues
OUT=gpurun_out/r04m; mkdir -p $OUT
timeout 900 python tools/brick_bench.py --cases pert32,pert32aux,pert8,pert128,pert1 --order weight --sorted 0,8,6,12 2>&1 | grep -v amdgpu.ids > $OUT/sorted_walk.txt; cat $OUT/sorted_walk.txt

# ---------------------------------------------------------------- r04_run14.sh
# round 4: phase profile of the sorted walk
OUT=gpurun_out/r04n; mkdir -p $OUT
(DDRR_SORTED=8 timeout 600 python tools/brick_profile.py --cases pert32,pert32aux --storage q16p; timeout 600 python tools/brick_profile.py --cases pert32 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/sorted_profile.txt; cat $OUT/sorted_profile.txt

# ---------------------------------------------------------------- r04_run15.sh
# round 4: the sorted walk, sort pass and batch entries software-pipelined
OUT=gpurun_out/r04o; mkdir -p $OUT
timeout 900 python tools/brick_bench.py --cases pert32,pert32aux,pert8,pert128,pert1 --order weight --sorted 0,8,5 2>&1 | grep -v amdgpu.ids > $OUT/sorted_walk.txt; cat $OUT/sorted_walk.txt
(DDRR_SORTED=8 timeout 600 python tools/brick_profile.py --cases pert32,pert32aux --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/sorted_profile.txt; cat $OUT/sorted_profile.txt

# ---------------------------------------------------------------- r04_run16.sh
# round 4: the sorted walk without any delivery (timing experiment): are the delivery atomics what its batches wait for?
OUT=gpurun_out/r04p; mkdir -p $OUT
(DDRR_EXP_FLAGS="-DDDRR_NO_DELIVER" timeout 600 python tools/brick_bench.py --cases pert32,pert32aux --order weight --sorted 0,8) 2>&1 | grep -v amdgpu.ids > $OUT/no_deliver.txt; cat $OUT/no_deliver.txt

# ---------------------------------------------------------------- r04_run17.sh
# round 4: the marcher's channel backward on the bricks: parity on the device, timings
OUT=gpurun_out/r04q; mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -x -q -k "trilinear or channel" 2>&1 | tail -8) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; cat $OUT/channels.txt

# ---------------------------------------------------------------- r04_run18.sh
# round 4: randomised sweep of the brick kernels (incl. guarded bricks, channel backward) against the per-ray kernels
OUT=gpurun_out/r04r; mkdir -p $OUT
(timeout 900 python tools/fuzz_bricks.py --cases 64 --seed 4) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz.txt; tail -70 $OUT/fuzz.txt
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 5 --smooth) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_smooth.txt; tail -5 $OUT/fuzz_smooth.txt
(timeout 300 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "^B 8\|^B 1:" $OUT/channels.txt

# ---------------------------------------------------------------- r04_run19.sh
# round 4: few-pose launches (registration: B = 1): brick variants -- does a second workgroup per CU hide the staging latency?
OUT=gpurun_out/r04s; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux --variants=-2,10,2,0,3) 2>&1 | grep -v amdgpu.ids > $OUT/few_poses_variants.txt; cat $OUT/few_poses_variants.txt

# ---------------------------------------------------------------- r04_run20.sh
# round 4: few-pose launches: the product's packed 16-bit bricks against two 32^3 16-bit workgroups per CU
OUT=gpurun_out/r04s; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32aux --variants=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/few_poses_q16p.txt; cat $OUT/few_poses_q16p.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32aux --variants=10,1) 2>&1 | grep -v amdgpu.ids > $OUT/few_poses_q16.txt; cat $OUT/few_poses_q16.txt

# ---------------------------------------------------------------- r04_run21.sh
# round 4: where a single-pose launch spends its wave time (registration, B = 1)
OUT=gpurun_out/r04s; mkdir -p $OUT
(timeout 600 python tools/brick_profile.py --cases pert1aux,pert1,pert4aux --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/phase_profile_few_poses.txt; cat $OUT/phase_profile_few_poses.txt

# ---------------------------------------------------------------- r04_run22.sh
# round 4: the next brick requested ahead (claim, lookups, image into registers): parity, then timings at 1..32 poses
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,8192,4096) 2>&1 | grep -v amdgpu.ids > $OUT/look_ahead.txt; cat $OUT/look_ahead.txt

# ---------------------------------------------------------------- r04_run23.sh
# round 4: lazy staging of few-unit bricks on / off (dbg 8192) / no look-ahead at all (dbg 4096)
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux --variants=-2 --storage q16p --dbg 0,8192,4096) 2>&1 | grep -v amdgpu.ids > $OUT/lazy.txt; cat $OUT/lazy.txt
(timeout 900 python -m pytest tests -m gpu -x -q -k "brick or registration or baseline" 2>&1 | tail -4) > $OUT/tests2.txt; cat $OUT/tests2.txt

# ---------------------------------------------------------------- r04_run24.sh
# round 4: per-brick durations of a single-pose launch (what a brick costs when almost nothing is walked)
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 600 python tools/brick_times.py --cases pert1,pert1aux,pert32aux --variant=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/brick_times_few_poses.txt; cat $OUT/brick_times_few_poses.txt

# ---------------------------------------------------------------- r04_run25.sh
# round 4: look-ahead with rows worked out ahead (few poses) and lazy staging: stage trace, timings, tests
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 600 python tools/brick_times.py --cases pert1,pert1aux --variant=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/brick_times_few_poses.txt; cat $OUT/brick_times_few_poses.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,8192,4096) 2>&1 | grep -v amdgpu.ids > $OUT/lazy.txt; cat $OUT/lazy.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $OUT/tests.txt; cat $OUT/tests.txt

# ---------------------------------------------------------------- r04_run26.sh
# round 4: look-ahead with rows worked out ahead (few poses) and lazy staging: stage trace, timings
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 600 python tools/brick_times.py --cases pert1,pert1aux --variant=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/brick_times_few_poses.txt; cat $OUT/brick_times_few_poses.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,8192,4096) 2>&1 | grep -v amdgpu.ids > $OUT/lazy.txt; cat $OUT/lazy.txt

# ---------------------------------------------------------------- r04_run27.sh
# round 4: stage trace of single-pose launches with the look-ahead (lazy staging on / off)
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 600 python tools/brick_times.py --cases pert1,pert1aux --variant=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/brick_times_few_poses.txt; cat $OUT/brick_times_few_poses.txt

# ---------------------------------------------------------------- r04_run28.sh
# round 4: look-ahead + rows ahead + lazy staging, pool kept: stage trace, timings, tests
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 600 python tools/brick_times.py --cases pert1,pert1aux --variant=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/brick_times_few_poses.txt; cat $OUT/brick_times_few_poses.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,8192,4096) 2>&1 | grep -v amdgpu.ids > $OUT/lazy.txt; cat $OUT/lazy.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $OUT/tests.txt; cat $OUT/tests.txt

# ---------------------------------------------------------------- r04_run29.sh
# round 4: lazy staging debug
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 300 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert32aux --variants=-2 --storage q16p --dbg 8192) 2>&1 | grep -v amdgpu.ids | cut -c1-230 > $OUT/dbg_a.txt; cat $OUT/dbg_a.txt
(timeout 300 python tools/brick_bench.py --cases pert1 --variants=-2 --storage q16p --dbg 0) 2>&1 | grep -v amdgpu.ids | cut -c1-230 | tail -5 > $OUT/dbg_b.txt; cat $OUT/dbg_b.txt

# ---------------------------------------------------------------- r04_run30.sh
# round 4: look-ahead x {pool, second row table}: which combination is fastest at 1 / 8 / 32 poses
OUT=gpurun_out/r04t; mkdir -p $OUT; : > $OUT/matrix.txt
for flags in "" "-DDDRR_NO_POOL" "-DDDRR_NO_ROWS2" "-DDDRR_NO_POOL -DDDRR_NO_ROWS2"; do
  echo "### build flags: [$flags]" >> $OUT/matrix.txt
  (DDRR_EXP_FLAGS="$flags" timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,4096) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 >> $OUT/matrix.txt
done
cat $OUT/matrix.txt

# ---------------------------------------------------------------- r04_run31.sh
# round 4: the pooled end of a brick on / off (dbg 2048) by number of poses, look-ahead on
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2,pert2aux,pert4,pert4aux,pert8,pert8aux,pert16aux --variants=-2 --storage q16p --dbg 0,2048) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/pool_by_poses.txt; cat $OUT/pool_by_poses.txt

# ---------------------------------------------------------------- r04_run32.sh
# round 4: look-ahead in the product build: GPU tests, bench headline + config 4
OUT=gpurun_out/r04u; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 600 python bench.py --no-configs --no-cpu-baseline > $OUT/bench_headline.json) 2> $OUT/bench_headline.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04u/bench_headline.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['forward']['frac'], d['roofline']['forward']['kernel_ms'])
PY
(timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_config_4.json) 2> $OUT/bench_config_4.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04u/bench_config_4.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms'])
PY

# ---------------------------------------------------------------- r04_run33.sh
# round 4: evidence pass 1 on the final kernels -- rocprofv3 stats + PMC traffic of the bench command, issue-bound counters
OUT=gpurun_out/r04v; mkdir -p $OUT
timeout 900 tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -14 $OUT/rocprof_bench.txt
timeout 1500 python tools/issue_bound.py $OUT 2>&1 | grep -v amdgpu.ids | tail -3
ls $OUT $OUT/prof | head -30

# ---------------------------------------------------------------- r04_run34.sh
# round 4: evidence pass 2 on the final kernels -- full GPU tests, smoke, every bench config
OUT=gpurun_out/r04w; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
( time timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err ) 2>&1 | grep real; grep "\[bench\]" $OUT/bench_headline.err | head -12
for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_$c.err; grep "\[bench\] config $c:" $OUT/bench_$c.err; done
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; tail -4 $OUT/channels.txt
ls $OUT

# ---------------------------------------------------------------- r04_run35.sh
# round 4: the look-ahead at large batches (claims held ahead against the balance of the launch's end)
OUT=gpurun_out/r04w; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert64,pert128,pert512,pert64aux,pert128aux --variants=-2 --storage q16p --order weight --dbg 0,4096) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/look_ahead_large.txt; cat $OUT/look_ahead_large.txt

# ---------------------------------------------------------------- r04_run36.sh
OUT=gpurun_out/r04x; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_brick_storage.py -m gpu -x -q -k look_ahead 2>&1 | tail -15) > $OUT/t.txt; cat $OUT/t.txt

# ---------------------------------------------------------------- r04_run37.sh
# round 4: the registration config with every brick storage (the phantom has 644 of 2048 bricks on the fp32 path)
OUT=gpurun_out/r04x; mkdir -p $OUT
for st in q16p q16 f32; do timeout 600 python bench.py --config 4 --no-cpu-baseline --storage $st > $OUT/c4_$st.json 2> $OUT/c4_$st.err; grep "\[bench\] config 4:" $OUT/c4_$st.err | cut -c1-160; python -c "
import json;d=json.load(open('$OUT/c4_$st.json'));print('$st', round(d['value'],1),'it/s', d['config'].get('brick_storage_fallbacks'))"; done

# ---------------------------------------------------------------- r04_run38.sh
# round 4: halves of fp32-path bricks requested ahead too: tests, config 4, fuzz, guard cost
OUT=gpurun_out/r04y; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $OUT/tests.txt; cat $OUT/tests.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err; grep "\[bench\] config 4:" $OUT/c4.err | cut -c1-170; python -c "
import json;d=json.load(open('$OUT/c4.json'));print(round(d['value'],1),'it/s')"
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 7) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz.txt; grep -c "<<<" $OUT/fuzz.txt; tail -1 $OUT/fuzz.txt
(timeout 600 python tools/guard_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/guard.txt; cat $OUT/guard.txt | cut -c1-250

# ---------------------------------------------------------------- r04_run39.sh
# round 4: the next unit's image in shares taken by whichever waves finish first: tests, timings, config 4
OUT=gpurun_out/r04z; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,4096) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/shares.txt; cat $OUT/shares.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err; grep "\[bench\] config 4:" $OUT/c4.err | cut -c1-170; python -c "
import json;d=json.load(open('$OUT/c4.json'));print(round(d['value'],1),'it/s')"

# ---------------------------------------------------------------- r04_run40.sh
# round 4: the record on accumulated alphas with a tie window (24 instructions per step) against plane counters (27)
OUT=gpurun_out/r04aa; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert32aux,pert1aux,pert8aux --variants=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/rec_new.txt; cat $OUT/rec_new.txt
(DDRR_EXP_FLAGS="-DDDRR_RECORD_PLANE_COUNTERS" timeout 900 python tools/brick_bench.py --cases pert32aux,pert1aux,pert8aux --variants=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/rec_old.txt; cat $OUT/rec_old.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $OUT/tests.txt; cat $OUT/tests.txt

# ---------------------------------------------------------------- r04_run41.sh
# round 4: final tree check: GPU tests, smoke, short headline
OUT=gpurun_out/r04ab; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python bench.py --no-configs --no-cpu-baseline > $OUT/h.json 2> $OUT/h.err; grep "\[bench\]" $OUT/h.err | head -3 | cut -c1-200
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err; grep "\[bench\] config 4:" $OUT/c4.err | cut -c1-170

# ---------------------------------------------------------------- r04_run42.sh
# round 4: length-class thresholds of the forward + record kernel (grouped runs of 8 pixels) and the forward kernel
OUT=gpurun_out/r04ac; mkdir -p $OUT
(timeout 1200 python tools/brick_bench.py --cases pert32aux,pert32 --variants=-2 --storage q16p --order weight --classes 14:32,16:36,18:40,20:44,22:48,24:54,18:36,18:44,20:40,16:40) 2>&1 | grep -v amdgpu.ids | cut -c1-40,100-200 > $OUT/classes.txt; cat $OUT/classes.txt

# ---------------------------------------------------------------- r04_run43.sh
# round 4: the channel render's volume gradient on the bricks: tests, timings, fuzz
OUT=gpurun_out/r04ad; mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -x -q -k "channel" 2>&1 | tail -8) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "VOLUME\|channel backward, rays" $OUT/channels.txt | cut -c1-330

# ---------------------------------------------------------------- r04_run45.sh
# round 4: channel volume gradient on the bricks: full GPU tests, channel timings, randomised sweep
OUT=gpurun_out/r04ae; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "VOLUME" $OUT/channels.txt | cut -c1-330
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 9) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz.txt; grep -c "<<<" $OUT/fuzz.txt; tail -1 $OUT/fuzz.txt

# ---------------------------------------------------------------- r04_run46.sh
# round 4: bench with events on the dominant kernel only inside the timed region
OUT=gpurun_out/r04af; mkdir -p $OUT
( time timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err ) 2>&1 | grep real; grep "\[bench\]" $OUT/bench_headline.err | head -12 | cut -c1-230
for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_$c.err; grep "\[bench\] config $c:" $OUT/bench_$c.err | cut -c1-200; done

# ---------------------------------------------------------------- r04_run47.sh
# round 4: kernel trace of the headline step with the lighter instrumentation: what fills the 80 us between two brick launches
OUT=gpurun_out/r04ag; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$ROOT/$OUT/trace" -o bench --output-format csv -- python "$ROOT/bench.py" --no-cpu-baseline --no-configs > "$ROOT/$OUT/line.json") > $OUT/trace.log 2>&1
python - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/r04ag/trace/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'siddon_fwd_brick_kernel<true' in r['Kernel_Name']]
k=len(idx)//2
a,b=idx[k],idx[k+1]
t0=int(rows[a]['Start_Timestamp']); prev=None
for r in rows[a:b+1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"{(s-t0)/1000:9.1f} us  dur {(e-s)/1000:8.1f}  gap {((s-prev)/1000 if prev else 0):6.1f}  {r['Kernel_Name'][:80]}")
    prev=e
print("step period", (int(rows[b]['Start_Timestamp'])-t0)/1000)
PY

# ---------------------------------------------------------------- r04_run48.sh
OUT=gpurun_out/r04ah; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 600 python bench.py --no-configs --no-cpu-baseline > $OUT/h.json 2> $OUT/h.err; grep "\[bench\]" $OUT/h.err | head -3 | cut -c1-200

# ---------------------------------------------------------------- r04_run49.sh
# round 4: final randomised sweeps of every brick kernel against the per-ray kernels
OUT=gpurun_out/r04ai; mkdir -p $OUT
for seed in 22; do (timeout 600 python tools/fuzz_bricks.py --cases 24 --seed $seed) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_$seed.txt; echo "seed $seed flagged: $(grep -c '<<<' $OUT/fuzz_$seed.txt)"; tail -1 $OUT/fuzz_$seed.txt; done
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 23 --smooth) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_smooth.txt; echo "smooth flagged: $(grep -c '<<<' $OUT/fuzz_smooth.txt)"; tail -1 $OUT/fuzz_smooth.txt
grep "<<<" $OUT/fuzz_2*.txt $OUT/fuzz_smooth.txt | cut -c1-330 | head -20

# ---------------------------------------------------------------- r04_run50.sh
# round 4: marcher channel backward on fp32 values + label map: tests, timings, smooth sweep
OUT=gpurun_out/r04aj; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "^trilinear" $OUT/channels.txt | cut -c1-300
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 23 --smooth) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_smooth.txt; echo "smooth flagged: $(grep -c '<<<' $OUT/fuzz_smooth.txt)"; tail -1 $OUT/fuzz_smooth.txt

# ---------------------------------------------------------------- r04_run51.sh
# round 4: merged clear of image / record and counter at few poses: tests, config 4, headline
OUT=gpurun_out/r04ak; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_config_4.json 2> $OUT/c4.err; grep "config 4:" $OUT/c4.err | cut -c1-150
timeout 600 python bench.py --no-configs --no-cpu-baseline > $OUT/h.json 2> $OUT/h.err; grep "config headline:" $OUT/h.err | cut -c1-150

# ---------------------------------------------------------------- r04_run52.sh
# round 4: any D.z on the forward kernel's bricks + its channel mode: new tests, channel tests, A/B timings, headline check
OUT=gpurun_out/r04al; mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -k "any_depth or channel or storage or look_ahead or bricks" 2>&1 | tail -15) > $OUT/gpu_tests_subset.txt; cat $OUT/gpu_tests_subset.txt
(timeout 600 python tools/channels_fwd_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels_fwd.txt; cat $OUT/channels_fwd.txt
timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 20 --warmup 5 > $OUT/h.json 2> $OUT/h.err; grep "config headline:" $OUT/h.err | cut -c1-200

# ---------------------------------------------------------------- r04_run53.sh
# round 4: how many rounds before the launch's end the look-ahead stops claiming (2 / 3 / 4): 512^3 and the 133-slice CT
OUT=gpurun_out/r04am; mkdir -p $OUT
(timeout 600 python tools/brick_bench.py --variants -2 --storage q16p --cases pert1,pert1aux,pert8,pert8aux,pert32,pert32aux --dbg 0,8192,16384,4096) 2>&1 | grep -v amdgpu.ids > $OUT/look_rounds_512.txt; cat $OUT/look_rounds_512.txt | cut -c1-230
(timeout 600 python tools/channels_fwd_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels_fwd.txt; cat $OUT/channels_fwd.txt
(timeout 600 python tools/channels_fwd_bench.py --cube) 2>&1 | grep -v amdgpu.ids > $OUT/channels_fwd_cube.txt; cat $OUT/channels_fwd_cube.txt

# ---------------------------------------------------------------- r04_run54.sh
# round 4: any D.z staged as quads (both brick kernels), look-ahead stops 3 rounds before the end: GPU tests, the 133-slice CT, headline
OUT=gpurun_out/r04an; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
(timeout 600 python tools/channels_fwd_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/example_ct_kernels.txt; cat $OUT/example_ct_kernels.txt
timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 20 --warmup 5 > $OUT/h.json 2> $OUT/h.err; grep "config headline:" $OUT/h.err | cut -c1-200

# ---------------------------------------------------------------- r04_run55.sh
# round 4: fp32 bricks -- the general kernel (quad staging) against the configurable kernel's 32^3 fp32 variant, 256^3 and 512^3
OUT=gpurun_out/r04ao; mkdir -p $OUT
for size in 256 512; do
(timeout 600 python tools/brick_bench.py --size $size --variants=-1,0 --storage f32 --cases pert1,pert1aux,pert8,pert8aux,pert32,pert32aux) 2>&1 | grep -v amdgpu.ids > $OUT/f32_general_vs_cfg_$size.txt; cut -c1-200 $OUT/f32_general_vs_cfg_$size.txt
done

# ---------------------------------------------------------------- r04_run56.sh
# round 4: one pose, fp32 bricks: where the general kernel's 7-12 % over the configurable kernel's 32^3 variant come from (phase profile), and the packed 16-bit bricks beside them
OUT=gpurun_out/r04ap; mkdir -p $OUT
(timeout 600 python tools/brick_profile.py --variants=-1,0 --storage f32 --cases pert1,pert1aux; timeout 600 python tools/brick_profile.py --variants=-2 --storage q16p --cases pert1,pert1aux) 2>&1 | grep -v amdgpu.ids > $OUT/phase_profile_one_pose.txt; cat $OUT/phase_profile_one_pose.txt

# ---------------------------------------------------------------- r04_run57.sh
# round 4: bricks out of every pose's view passed over where they are claimed (dbg 32768: off): tests, 512^3 at 1 / 8 / 32 poses, the example CT, registration, headline
OUT=gpurun_out/r04aq; mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -k "look_ahead or any_depth or channel or storage or bricks or registration" 2>&1 | tail -5) > $OUT/gpu_tests_subset.txt; cat $OUT/gpu_tests_subset.txt
(timeout 600 python tools/brick_bench.py --variants=-2 --storage q16p --cases pert1,pert1aux,pert8,pert8aux,pert32,pert32aux --dbg 0,32768) 2>&1 | grep -v amdgpu.ids > $OUT/in_view_512.txt; cut -c1-200 $OUT/in_view_512.txt
(timeout 600 python tools/brick_bench.py --variants=-1 --storage f32 --cases pert1,pert1aux --dbg 0,32768) 2>&1 | grep -v amdgpu.ids > $OUT/in_view_512_general.txt; cut -c1-200 $OUT/in_view_512_general.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_config_4.json 2> $OUT/c4.err; grep "config 4:" $OUT/c4.err | cut -c1-200
timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 20 --warmup 5 > $OUT/h.json 2> $OUT/h.err; grep "config headline:" $OUT/h.err | cut -c1-200

# ---------------------------------------------------------------- r04_run58.sh
# round 4: evidence on the final tree -- GPU tests, smoke, every bench config, rocprofv3 stats + PMC traffic, channel timings (both label maps)
OUT=gpurun_out/r04ar; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
( time timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err ) 2>&1 | grep real; grep "\[bench\]" $OUT/bench_headline.err | head -12 | cut -c1-230
for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_$c.err; grep "\[bench\] config $c:" $OUT/bench_$c.err | cut -c1-200; done
timeout 900 tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -14 $OUT/rocprof_bench.txt
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "^B \|^# " $OUT/channels.txt | cut -c1-330
ls $OUT $OUT/prof | head -30

# ---------------------------------------------------------------- r04_run59.sh
# round 4: the randomised sweep on the final tree (every D.z now on every brick storage)
OUT=gpurun_out/r04as; mkdir -p $OUT
(timeout 1200 python tools/fuzz_bricks.py --cases 128 --seed 11) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_bricks.txt; echo "flagged: $(grep -c "<<<" $OUT/fuzz_bricks.txt)"; grep "<<<" $OUT/fuzz_bricks.txt | cut -c1-330 | head; tail -2 $OUT/fuzz_bricks.txt | cut -c1-500

# ---------------------------------------------------------------- r04_run60.sh
# round 4: last check of the committed tree -- GPU tests, smoke, the default bench line
OUT=gpurun_out/r04at; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real; grep "\[bench\] config headline" $OUT/bench_default.err | cut -c1-200; python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['metric'][:40], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['forward']['frac'], d['cpu_baseline'])"

# ---------------------------------------------------------------- r04_run61.sh
# round 4: forward-only walks enter a brick with the plain quotients (n * rcp(d), floor cell): GPU tests, forward timings, randomised sweep, headline
OUT=gpurun_out/r04av; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $OUT/gpu_tests.txt; grep "^FAILED\|passed\|failed\|^E   *assert" $OUT/gpu_tests.txt | cut -c1-250
(timeout 600 python tools/brick_bench.py --variants=-2 --storage q16p --cases pert1,pert8,pert32,pert32aux,pert128) 2>&1 | grep -v amdgpu.ids > $OUT/fwd_plain_entry.txt; cut -c1-200 $OUT/fwd_plain_entry.txt
(timeout 1200 python tools/fuzz_bricks.py --cases 128 --seed 11) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_bricks.txt; echo "flagged: $(grep -c "<<<" $OUT/fuzz_bricks.txt)"; tail -1 $OUT/fuzz_bricks.txt | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline > $OUT/h.json 2> $OUT/h.err; grep "config headline\|config 2\|config 5" $OUT/h.err | cut -c1-200

# ---------------------------------------------------------------- r04_run62.sh
# round 4: the driver's multi-GPU launch line at N = 1 (torch.distributed.run, RCCL rendezvous on 127.0.0.1)
OUT=gpurun_out/r04aw; mkdir -p $OUT
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 > $OUT/bench_torchrun_n1.json 2> $OUT/bench_torchrun_n1.err ) 2>&1 | grep real; tail -c 600 $OUT/bench_torchrun_n1.json; grep -c . $OUT/bench_torchrun_n1.json; grep -i "error\|traceback" $OUT/bench_torchrun_n1.err | head -5

# ---------------------------------------------------------------- r04_run63.sh
# round 4: kernel durations under rocprofv3 (no launch gaps) at one pose: general kernel, configurable fp32 32^3, packed 16-bit bricks
OUT=gpurun_out/r04ax; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace" -o t --output-format csv -- python "$ROOT/tools/brick_bench.py" --variants=-1,0 --storage f32 --cases pert1,pert1aux,pert4 ) > $OUT/trace.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace2" -o t --output-format csv -- python "$ROOT/tools/brick_bench.py" --variants=-2 --storage q16p --cases pert1,pert1aux,pert4 ) > $OUT/trace2.log 2>&1
grep "ms (best" $OUT/trace.log $OUT/trace2.log | cut -c1-190
python - <<'PY'
import csv, glob
for d in ("trace", "trace2"):
    for f in glob.glob(f"gpurun_out/r04ax/{d}/**/*kernel_stats.csv", recursive=True):
        for r in list(csv.DictReader(open(f)))[:8]:
            print(d, f"{float(r['AverageNs'])/1e3:9.1f} us avg  min {float(r['MinNs'])/1e3:8.1f}  x{r['Calls']:>5}  {r['Name'][:100]}")
PY

# ---------------------------------------------------------------- r04_run64.sh
# round 4: the any-depth tests with a volume at a dword- but not 16-byte-aligned address
OUT=gpurun_out/r04ay; mkdir -p $OUT
(timeout 600 python -m pytest tests -m gpu -q -k "any_depth or label_alignment" 2>&1 | tail -5) > $OUT/t.txt; cat $OUT/t.txt
