#!/bin/bash
# round 4: last check of the committed tree -- GPU tests, smoke, the default bench line
OUT=gpurun_out/r04at; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real; grep "\[bench\] config headline" $OUT/bench_default.err | cut -c1-200; python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['metric'][:40], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['forward']['frac'], d['cpu_baseline'])"
