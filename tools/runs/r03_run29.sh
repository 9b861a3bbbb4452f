#!/bin/bash
OUT=gpurun_out/r03ac; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for c in headline 2 3 4 5; do
  python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_config_$c.err; tail -2 $OUT/bench_config_$c.err | grep -v amdgpu
done
python bench.py --storage f32 --no-cpu-baseline > $OUT/bench_config_headline_f32_bricks.json 2> $OUT/f32.err; tail -1 $OUT/f32.err
