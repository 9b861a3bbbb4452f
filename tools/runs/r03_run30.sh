#!/bin/bash
OUT=gpurun_out/r03ad; mkdir -p $OUT
timeout 900 python tools/fuzz_bricks.py --cases 60 --seed 7 2>&1 | grep -v amdgpu > $OUT/fuzz.txt; grep "<<<\|worst\|Error\|error" $OUT/fuzz.txt | head -20; tail -3 $OUT/fuzz.txt
timeout 600 python tools/fuzz_bricks.py --cases 40 --seed 8 --smooth 2>&1 | grep -v amdgpu > $OUT/fuzz_smooth.txt; grep "<<<\|worst\|Error\|error" $OUT/fuzz_smooth.txt | head -20
