#!/bin/bash
OUT=gpurun_out/r03q; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for c in headline 5 2; do
  python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_config_$c.err; tail -3 $OUT/bench_config_$c.err | grep -v amdgpu
  cat $OUT/bench_config_$c.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('forward'))"
done
