#!/bin/bash
# round 4: merged clear of image / record and counter at few poses: tests, config 4, headline
OUT=gpurun_out/r04ak; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_config_4.json 2> $OUT/c4.err; grep "config 4:" $OUT/c4.err | cut -c1-150
timeout 600 python bench.py --no-configs --no-cpu-baseline > $OUT/h.json 2> $OUT/h.err; grep "config headline:" $OUT/h.err | cut -c1-150
