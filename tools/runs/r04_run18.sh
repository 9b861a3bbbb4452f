#!/bin/bash
# round 4: randomised sweep of the brick kernels (incl. guarded bricks, channel backward) against the per-ray kernels
OUT=gpurun_out/r04r; mkdir -p $OUT
(timeout 900 python tools/fuzz_bricks.py --cases 64 --seed 4) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz.txt; tail -70 $OUT/fuzz.txt
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 5 --smooth) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_smooth.txt; tail -5 $OUT/fuzz_smooth.txt
(timeout 300 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "^B 8\|^B 1:" $OUT/channels.txt
