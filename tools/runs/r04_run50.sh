#!/bin/bash
# round 4: marcher channel backward on fp32 values + label map: tests, timings, smooth sweep
OUT=gpurun_out/r04aj; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "^trilinear" $OUT/channels.txt | cut -c1-300
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 23 --smooth) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_smooth.txt; echo "smooth flagged: $(grep -c '<<<' $OUT/fuzz_smooth.txt)"; tail -1 $OUT/fuzz_smooth.txt
