#!/bin/bash
# DISPOSABLE LOG, not source: the command lists of every gpurun call of round 3, in order, as they
# were run (hard-coded shapes and output directories; nothing imports or executes this file).
# The summaries they produced are under profiles/r03/.

# ---------------------------------------------------------------- r03_run1.sh
# round 3, GPU call 1: blocked record (tests, kernel timing with / without the lane swap, PMC, bench)
OUT=gpurun_out/r03a; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/brick_bench.py --cases pert32,pert32aux --dbg 0,128,64 --classes 18:40 > $OUT/brick_bench.txt 2>&1; cat $OUT/brick_bench.txt | grep -v amdgpu.ids
bash tools/pmc_run.sh $OUT/pmc_aux --case pert32 --kernel brick --aux 1 > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pmc_aux > $OUT/pmc_aux_summary.txt 2>&1; head -40 $OUT/pmc_aux_summary.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err

# ---------------------------------------------------------------- r03_run2.sh
# round 3, GPU call 2: brick variants (storage, shape, workgroups per CU)
OUT=gpurun_out/r03b; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/brick_bench.py --cases pert32,pert32aux,base32,pert1 --variants=-1,0,1,2,3,4,5,6,7,8,9 > $OUT/brick_variants.txt 2>&1; grep -v amdgpu.ids $OUT/brick_variants.txt

# ---------------------------------------------------------------- r03_run3.sh
OUT=gpurun_out/r03c; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/brick_bench.py --cases pert32,pert32aux,base32,pert1,pert128 --variants=0,1,2,4,5,6 > $OUT/brick_variants.txt 2>&1; grep -v amdgpu.ids $OUT/brick_variants.txt
python tools/brick_profile.py --cases pert32,pert32aux --variants=0,1,2,4 > $OUT/phase_profile.txt 2>&1; grep -v amdgpu.ids $OUT/phase_profile.txt

# ---------------------------------------------------------------- r03_run4.sh
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 200 python tools/brick_bench.py --cases pert32,pert32aux,base32,pert1 --variants=0,16,32,18,20 --sqw 8 > $OUT/sq_variants.txt 2>&1; grep -v amdgpu.ids $OUT/sq_variants.txt | cut -c1-175
timeout 200 python tools/brick_bench.py --cases pert32,pert32aux --variants=16,20 --sqw 4,6,10,12 > $OUT/sq_widths.txt 2>&1; grep -v amdgpu.ids $OUT/sq_widths.txt | cut -c1-175
timeout 200 python tools/brick_profile.py --cases pert32,pert32aux --variants=16 > $OUT/phase_profile_sq.txt 2>&1; grep -v amdgpu.ids $OUT/phase_profile_sq.txt

# ---------------------------------------------------------------- r03_run5.sh
OUT=gpurun_out/r03e; mkdir -p $OUT
timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=5 --classes 18:40,22:48,26:56,30:64,22:56 > $OUT/classes_z64.txt 2>&1; grep -v amdgpu.ids $OUT/classes_z64.txt | cut -c1-175

# ---------------------------------------------------------------- r03_run6.sh
OUT=gpurun_out/r03f; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -4 $OUT/bench_headline.err; cat $OUT/bench_headline.json
python bench.py --storage f32 --no-cpu-baseline > $OUT/bench_headline_f32.json 2> $OUT/bench_headline_f32.err; tail -3 $OUT/bench_headline_f32.err

# ---------------------------------------------------------------- r03_run7.sh
OUT=gpurun_out/r03g; mkdir -p $OUT
timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=0,5 --dbg 0,256 > $OUT/no_ray_loads.txt 2>&1; grep -v amdgpu.ids $OUT/no_ray_loads.txt | cut -c1-175
DDRR_EXP_FLAGS="-DDDRR_WALK_CHECK4" timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=0,5 > $OUT/check4.txt 2>&1; grep -v amdgpu.ids $OUT/check4.txt | cut -c1-175
bash tools/prof_bench.sh $OUT/prof > $OUT/prof_bench.txt 2>&1; cat $OUT/prof_bench.txt | cut -c1-200

# ---------------------------------------------------------------- r03_run8.sh
OUT=gpurun_out/r03h; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
DDRR_EXP_FLAGS="-DDDRR_WALK_CHECK4" timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=0,5 > $OUT/check4.txt 2>&1; grep -v amdgpu.ids $OUT/check4.txt | cut -c1-175
timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=0,5 > $OUT/check2.txt 2>&1; grep -v amdgpu.ids $OUT/check2.txt | cut -c1-175
timeout 300 python tools/channels_bench.py > $OUT/channels_synthetic.txt 2>&1; grep -v amdgpu.ids $OUT/channels_synthetic.txt
timeout 300 python tools/channels_bench.py --real-mask > $OUT/channels_real_mask.txt 2>&1; grep -v amdgpu.ids $OUT/channels_real_mask.txt

# ---------------------------------------------------------------- r03_run9.sh
OUT=gpurun_out/r03i; mkdir -p $OUT
for c in 2 3 4 5; do
  python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_config_$c.err; tail -3 $OUT/bench_config_$c.err | grep -v amdgpu
  python -c "
import json; d=json.load(open('$OUT/bench_config_$c.json'))
print('config $c', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'fwd', (d['roofline'].get('forward') or {}).get('frac'))
print(' parity', {k:v for k,v in (d.get('parity') or {}).items() if not isinstance(v,(str,dict))})
print(' extra', d.get('registration'))"
done

# ---------------------------------------------------------------- r03_run10.sh
OUT=gpurun_out/r03j; mkdir -p $OUT
bash tools/pmc_quick.sh $OUT/pmc_fwd --case pert32 --kernel brick --aux 0 --reps 5 > /dev/null 2>&1; cat $OUT/pmc_fwd/summary.txt | head -24
bash tools/pmc_quick.sh $OUT/pmc_aux --case pert32 --kernel brick --aux 1 --reps 5 > /dev/null 2>&1; cat $OUT/pmc_aux/summary.txt | head -24
python bench.py --config 2 > $OUT/bench_config_2.json 2> $OUT/bench_config_2.err; tail -2 $OUT/bench_config_2.err

# ---------------------------------------------------------------- r03_run11.sh
OUT=gpurun_out/r03k; mkdir -p $OUT
timeout 300 python tools/brick_profile.py --cases pert32,pert32aux,pert128 --variants=0,5 > $OUT/tail.txt 2>&1; grep -v amdgpu.ids $OUT/tail.txt | grep "variant\|lifetime\|barrier wait"

# ---------------------------------------------------------------- r03_run12.sh
OUT=gpurun_out/r03l; mkdir -p $OUT
timeout 400 python tools/brick_bench.py --cases pert32,pert32aux,pert128 --variants=5 --order id,center,weight --split 0:1,256:2,512:4,1024:4 > $OUT/order_split.txt 2>&1; grep -v amdgpu.ids $OUT/order_split.txt | cut -c1-190

# ---------------------------------------------------------------- r03_run13.sh
OUT=gpurun_out/r03m; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 400 python tools/brick_bench.py --cases pert32,pert32aux,pert128,pert1,base32,pert8 --variants=0,5 --dbg 512,0 > $OUT/order_product.txt 2>&1; grep -v amdgpu.ids $OUT/order_product.txt | cut -c1-190
timeout 300 python tools/brick_profile.py --cases pert32,pert32aux,pert128 --variants=5 > $OUT/tail.txt 2>&1; grep -v amdgpu.ids $OUT/tail.txt | grep "variant\|lifetime\|barrier wait"
python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -3 $OUT/bench_headline.err | grep -v amdgpu

# ---------------------------------------------------------------- r03_run14.sh
OUT=gpurun_out/r03n; mkdir -p $OUT
timeout 300 python tools/brick_profile.py --cases pert32,pert32aux,pert128 --variants=5 > $OUT/span.txt 2>&1; grep -v amdgpu.ids $OUT/span.txt | grep "variant\|launch span"

# ---------------------------------------------------------------- r03_run15.sh
OUT=gpurun_out/r03o; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python tools/brick_times.py --cases pert32,pert32aux,pert128 2>&1 | grep -v amdgpu > $OUT/brick_times.txt; cat $OUT/brick_times.txt
for c in headline 5 2; do
  python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_config_$c.err; tail -3 $OUT/bench_config_$c.err | grep -v amdgpu
done
python bench.py --storage f32 --no-cpu-baseline > $OUT/bench_config_headline_f32.json 2> $OUT/bench_config_headline_f32.err; tail -2 $OUT/bench_config_headline_f32.err | grep -v amdgpu

# ---------------------------------------------------------------- r03_run16.sh
OUT=gpurun_out/r03p; mkdir -p $OUT
timeout 600 python tools/brick_bench.py --variants 0,1 --dbg 0,2048 --cases pert32,pert32aux,pert8,pert128,base32 2>&1 | grep -v amdgpu > $OUT/pool.txt; cat $OUT/pool.txt
timeout 300 python tools/brick_profile.py --variants 1 --cases pert32,pert32aux 2>&1 | grep -v amdgpu > $OUT/prof.txt; cat $OUT/prof.txt
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log

# ---------------------------------------------------------------- r03_run17.sh
OUT=gpurun_out/r03q; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for c in headline 5 2; do
  python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_config_$c.err; tail -3 $OUT/bench_config_$c.err | grep -v amdgpu
  cat $OUT/bench_config_$c.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('forward'))"
done

# ---------------------------------------------------------------- r03_run18.sh
OUT=gpurun_out/r03r; mkdir -p $OUT
timeout 600 python tools/brick_bench.py --variants 0,1 --order weight --dbg 0,4096,8192,12288 --cases pert32,pert32aux,pert8,pert128 2>&1 | grep -v amdgpu > $OUT/pool.txt; cat $OUT/pool.txt
timeout 300 python tools/brick_profile.py --variants 1 --cases pert32,pert32aux 2>&1 | grep -v amdgpu > $OUT/prof.txt; cat $OUT/prof.txt
python -m pytest tests -m gpu -x -q -k "brick or baseline or config or q16" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log

# ---------------------------------------------------------------- r03_run19.sh
OUT=gpurun_out/r03s; mkdir -p $OUT
(echo "## heaviest first"; timeout 300 python tools/volgrad_bench.py 0; echo "## id order"; timeout 300 python tools/volgrad_bench.py 512) 2>&1 | grep -v amdgpu > $OUT/volgrad.txt; cat $OUT/volgrad.txt
(echo "## heaviest first"; timeout 300 python tools/channels_bench.py; echo "## real mask"; timeout 300 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu > $OUT/channels.txt; cat $OUT/channels.txt
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log

# ---------------------------------------------------------------- r03_run20.sh
OUT=gpurun_out/r03aj; mkdir -p $OUT
tools/prof_bench.sh $OUT > $OUT/rocprof_bench.txt 2>&1; head -40 $OUT/rocprof_bench.txt
cat $OUT/bench_line_under_trace.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else d['roofline'])"
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_bench_kernel_stats.csv \;
rm -rf $OUT/bench_trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_HIT_sum

# ---------------------------------------------------------------- r03_run21.sh
OUT=gpurun_out/r03u; mkdir -p $OUT
timeout 600 python tools/packed_bench.py 2>&1 | grep -v amdgpu > $OUT/packed.txt; cat $OUT/packed.txt

# ---------------------------------------------------------------- r03_run22.sh
OUT=gpurun_out/r03v; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for c in headline 4 5 2; do
  python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_config_$c.err; tail -3 $OUT/bench_config_$c.err | grep -v amdgpu
  cat $OUT/bench_config_$c.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('frac'), (d['roofline'].get('forward') or {}).get('frac'), d.get('parity'))"
done

# ---------------------------------------------------------------- r03_run23.sh
OUT=gpurun_out/r03w; mkdir -p $OUT
timeout 300 python tools/brick_profile.py --variants -2 --storage q16p --cases pert32,pert32aux,pert1,pert8 2>&1 | grep -v amdgpu > $OUT/prof.txt; cat $OUT/prof.txt

# ---------------------------------------------------------------- r03_run24.sh
OUT=gpurun_out/r03x; mkdir -p $OUT
(timeout 300 python tools/packed_bench.py 0; timeout 300 python tools/packed_bench.py 4096) 2>&1 | grep -v amdgpu > $OUT/prefetch.txt; cat $OUT/prefetch.txt
python -m pytest tests -m gpu -x -q -k "brick or baseline or config or q16 or sweep" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log

# ---------------------------------------------------------------- r03_run25.sh
OUT=gpurun_out/r03y; mkdir -p $OUT
(for f in 0 16384 32768; do timeout 300 python tools/packed_bench.py $f; done) 2>&1 | grep -v "amdgpu\|f32 bricks   0\.\(2\|4\|5\)" > $OUT/units.txt; cat $OUT/units.txt

# ---------------------------------------------------------------- r03_run26.sh
OUT=gpurun_out/r03z; mkdir -p $OUT
tools/pmc_quick.sh $OUT/fwd --case pert32 --kernel brick --aux 0 --storage q16p > $OUT/pmc_fwd_q16p.txt 2>&1; cat $OUT/pmc_fwd_q16p.txt
tools/pmc_quick.sh $OUT/aux --case pert32 --kernel brick --aux 1 --storage q16p > $OUT/pmc_fwd_record_q16p.txt 2>&1; cat $OUT/pmc_fwd_record_q16p.txt
rm -rf $OUT/fwd/trace $OUT/fwd/pmc_valu $OUT/fwd/pmc_lds $OUT/aux/trace $OUT/aux/pmc_valu $OUT/aux/pmc_lds

# ---------------------------------------------------------------- r03_run27.sh
OUT=gpurun_out/r03aa; mkdir -p $OUT
timeout 300 python tools/channels_bench.py --real-mask 2>&1 | grep -v amdgpu > $OUT/channels_real.txt; cat $OUT/channels_real.txt

# ---------------------------------------------------------------- r03_run28.sh
OUT=gpurun_out/r03ab; mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "channels or mask or general" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 300 python tools/channels_bench.py --real-mask 2>&1 | grep -v amdgpu > $OUT/channels_real.txt; cat $OUT/channels_real.txt

# ---------------------------------------------------------------- r03_run29.sh
OUT=gpurun_out/r03ac; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for c in headline 2 3 4 5; do
  python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_config_$c.err; tail -2 $OUT/bench_config_$c.err | grep -v amdgpu
done
python bench.py --storage f32 --no-cpu-baseline > $OUT/bench_config_headline_f32_bricks.json 2> $OUT/f32.err; tail -1 $OUT/f32.err

# ---------------------------------------------------------------- r03_run30.sh
OUT=gpurun_out/r03ad; mkdir -p $OUT
timeout 900 python tools/fuzz_bricks.py --cases 60 --seed 7 2>&1 | grep -v amdgpu > $OUT/fuzz.txt; grep "<<<\|worst\|Error\|error" $OUT/fuzz.txt | head -20; tail -3 $OUT/fuzz.txt
timeout 600 python tools/fuzz_bricks.py --cases 40 --seed 8 --smooth 2>&1 | grep -v amdgpu > $OUT/fuzz_smooth.txt; grep "<<<\|worst\|Error\|error" $OUT/fuzz_smooth.txt | head -20

# ---------------------------------------------------------------- r03_run31.sh
OUT=gpurun_out/r03ae; mkdir -p $OUT
timeout 600 python tools/brick_bench.py --variants 5,11,0,12 --order weight --cases pert32,pert32aux,pert8,pert128,base32 2>&1 | grep -v amdgpu > $OUT/c6.txt; cat $OUT/c6.txt
timeout 300 python tools/brick_profile.py --variants 5,11 --cases pert32,pert32aux 2>&1 | grep -v amdgpu > $OUT/prof.txt; grep "variant\|wave-steps\|hits per batch\|walk   \|barrier wait\|phase A \|unit pull\|batch pop" $OUT/prof.txt

# ---------------------------------------------------------------- r03_run32.sh
OUT=gpurun_out/r03af; mkdir -p $OUT
timeout 300 python tools/brick_profile.py --variants 11 --cases pert32,pert32aux 2>&1 | grep -v amdgpu > $OUT/prof.txt; cat $OUT/prof.txt

# ---------------------------------------------------------------- r03_run33.sh
OUT=gpurun_out/r03ag; mkdir -p $OUT
(timeout 300 python tools/packed_bench.py 0; timeout 300 python tools/packed_bench.py 4096) 2>&1 | grep -v amdgpu | grep "flags\|B   32\|B  128\|B    8\|B    1 " > $OUT/ticket_ahead.txt; cat $OUT/ticket_ahead.txt

# ---------------------------------------------------------------- r03_run34.sh
OUT=gpurun_out/r03ah; mkdir -p $OUT
python -m pytest tests/test_gpu_baseline_sizes.py -m gpu -x -q -k "properties or published" > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log

# ---------------------------------------------------------------- r03_run35.sh
OUT=gpurun_out/r03ai; mkdir -p $OUT
(for f in 0 8 1024 1032; do timeout 300 python tools/packed_bench.py $f; done) 2>&1 | grep -v amdgpu | grep "flags\|B   32 fwd+record\|B  128 fwd+record\|B    8 fwd+record" > $OUT/aux_grouping.txt; cat $OUT/aux_grouping.txt

# ---------------------------------------------------------------- r03_run36.sh
OUT=gpurun_out/r03aj; mkdir -p $OUT
tools/prof_bench.sh $OUT > $OUT/rocprof_bench.txt 2>&1; head -40 $OUT/rocprof_bench.txt
cat $OUT/bench_line_under_trace.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else d['roofline'])"
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_bench_kernel_stats.csv \;
rm -rf $OUT/bench_trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_HIT_sum

# ---------------------------------------------------------------- r03_run37.sh
OUT=gpurun_out/r03ak; mkdir -p $OUT
python bench.py --packed-record > $OUT/bench_packed_record.json 2> $OUT/err.txt; tail -2 $OUT/err.txt | grep -v amdgpu
python -c "
import json; d=json.load(open('$OUT/bench_packed_record.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']); print(d['parity'])"

# ---------------------------------------------------------------- r03_run38.sh
OUT=gpurun_out/r03al; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $OUT/bench_short.json 2> $OUT/err.txt; python -c "
import json; d=json.load(open('$OUT/bench_short.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"

# ---------------------------------------------------------------- r03_run39.sh
OUT=gpurun_out/r03am; mkdir -p $OUT
for k in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_short$k.json 2> $OUT/err.txt; python -c "
import json; d=json.load(open('$OUT/bench_short$k.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done

# ---------------------------------------------------------------- r03_run40.sh
OUT=gpurun_out/r03an; mkdir -p $OUT
timeout 300 python tools/sparse_bench.py 2>&1 | grep -v amdgpu > $OUT/sparse.txt; cat $OUT/sparse.txt

# ---------------------------------------------------------------- r03_run41.sh
OUT=gpurun_out/r03ao; mkdir -p $OUT
(timeout 300 python tools/packed_bench.py product; timeout 300 python tools/packed_bench.py 0; timeout 300 python tools/sparse_bench.py) 2>&1 | grep -v amdgpu | grep "flags\|B   32\|body" > $OUT/cmp.txt; cat $OUT/cmp.txt

# ---------------------------------------------------------------- r03_run42.sh
OUT=gpurun_out/r03ap; mkdir -p $OUT
(timeout 300 python tools/sparse_bench.py; timeout 300 python tools/packed_bench.py product) 2>&1 | grep -v amdgpu > $OUT/tools.txt; cat $OUT/tools.txt

# ---------------------------------------------------------------- r03_run43.sh
OUT=gpurun_out/r03aq; mkdir -p $OUT
(for f in 0 524288 786432 917504 1048576; do timeout 300 python tools/packed_bench.py $f; done) 2>&1 | grep -v amdgpu | grep "flags\|B   32\|B    8 f\|B  128 forward" > $OUT/early.txt; cat $OUT/early.txt
