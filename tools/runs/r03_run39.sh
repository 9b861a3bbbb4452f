#!/bin/bash
OUT=gpurun_out/r03am; mkdir -p $OUT
for k in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_short$k.json 2> $OUT/err.txt; python -c "
import json; d=json.load(open('$OUT/bench_short$k.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
