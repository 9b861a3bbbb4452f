#!/bin/bash
OUT=gpurun_out/r03o; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python tools/brick_times.py --cases pert32,pert32aux,pert128 2>&1 | grep -v amdgpu > $OUT/brick_times.txt; cat $OUT/brick_times.txt
for c in headline 5 2; do
  python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_config_$c.err; tail -3 $OUT/bench_config_$c.err | grep -v amdgpu
done
python bench.py --storage f32 --no-cpu-baseline > $OUT/bench_config_headline_f32.json 2> $OUT/bench_config_headline_f32.err; tail -2 $OUT/bench_config_headline_f32.err | grep -v amdgpu
