#!/bin/bash
OUT=gpurun_out/r03aq; mkdir -p $OUT
(for f in 0 524288 786432 917504 1048576; do timeout 300 python tools/packed_bench.py $f; done) 2>&1 | grep -v amdgpu | grep "flags\|B   32\|B    8 f\|B  128 forward" > $OUT/early.txt; cat $OUT/early.txt
