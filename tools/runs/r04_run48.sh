#!/bin/bash
OUT=gpurun_out/r04ah; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 600 python bench.py --no-configs --no-cpu-baseline > $OUT/h.json 2> $OUT/h.err; grep "\[bench\]" $OUT/h.err | head -3 | cut -c1-200
