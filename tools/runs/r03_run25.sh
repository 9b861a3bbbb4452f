#!/bin/bash
OUT=gpurun_out/r03y; mkdir -p $OUT
(for f in 0 16384 32768; do timeout 300 python tools/packed_bench.py $f; done) 2>&1 | grep -v "amdgpu\|f32 bricks   0\.\(2\|4\|5\)" > $OUT/units.txt; cat $OUT/units.txt
