#!/bin/bash
# round 4, first GPU pass: the guarded 16-bit bricks + per-launch workspaces on the device
OUT=gpurun_out/r04a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_brick_storage.py -x -q 2>&1 | tail -15 > $OUT/guard_tests.txt; cat $OUT/guard_tests.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 600 python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -c 3000 $OUT/bench_headline.json
