#!/bin/bash
# round 4: halves of fp32-path bricks requested ahead too: tests, config 4, fuzz, guard cost
OUT=gpurun_out/r04y; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $OUT/tests.txt; cat $OUT/tests.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err; grep "\[bench\] config 4:" $OUT/c4.err | cut -c1-170; python -c "
import json;d=json.load(open('$OUT/c4.json'));print(round(d['value'],1),'it/s')"
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 7) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz.txt; grep -c "<<<" $OUT/fuzz.txt; tail -1 $OUT/fuzz.txt
(timeout 600 python tools/guard_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/guard.txt; cat $OUT/guard.txt | cut -c1-250
