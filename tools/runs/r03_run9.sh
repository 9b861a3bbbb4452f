#!/bin/bash
OUT=gpurun_out/r03i; mkdir -p $OUT
for c in 2 3 4 5; do
  python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_config_$c.err; tail -3 $OUT/bench_config_$c.err | grep -v amdgpu
  python -c "
import json; d=json.load(open('$OUT/bench_config_$c.json'))
print('config $c', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'fwd', (d['roofline'].get('forward') or {}).get('frac'))
print(' parity', {k:v for k,v in (d.get('parity') or {}).items() if not isinstance(v,(str,dict))})
print(' extra', d.get('registration'))"
done
