#!/bin/bash
OUT=gpurun_out/r03ak; mkdir -p $OUT
python bench.py --packed-record > $OUT/bench_packed_record.json 2> $OUT/err.txt; tail -2 $OUT/err.txt | grep -v amdgpu
python -c "
import json; d=json.load(open('$OUT/bench_packed_record.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']); print(d['parity'])"
