#!/bin/bash
# round 4: what the fp32 fallback of the 16-bit bricks costs; channel backward timings
OUT=gpurun_out/r04j; mkdir -p $OUT
timeout 900 python tools/guard_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/guard_bench.txt; cat $OUT/guard_bench.txt
timeout 600 python tools/channels_bench.py --real-mask 2>&1 | grep -v amdgpu.ids > $OUT/channels_real.txt; cat $OUT/channels_real.txt
