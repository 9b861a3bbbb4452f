#!/bin/bash
# round 4: the randomised sweep on the final tree (every D.z now on every brick storage)
OUT=gpurun_out/r04as; mkdir -p $OUT
(timeout 1200 python tools/fuzz_bricks.py --cases 128 --seed 11) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_bricks.txt; echo "flagged: $(grep -c "<<<" $OUT/fuzz_bricks.txt)"; grep "<<<" $OUT/fuzz_bricks.txt | cut -c1-330 | head; tail -2 $OUT/fuzz_bricks.txt | cut -c1-500
