#!/bin/bash
# round 4: channel volume gradient on the bricks: full GPU tests, channel timings, randomised sweep
OUT=gpurun_out/r04ae; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "VOLUME" $OUT/channels.txt | cut -c1-330
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 9) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz.txt; grep -c "<<<" $OUT/fuzz.txt; tail -1 $OUT/fuzz.txt
