#!/bin/bash
OUT=gpurun_out/r03ao; mkdir -p $OUT
(timeout 300 python tools/packed_bench.py product; timeout 300 python tools/packed_bench.py 0; timeout 300 python tools/sparse_bench.py) 2>&1 | grep -v amdgpu | grep "flags\|B   32\|body" > $OUT/cmp.txt; cat $OUT/cmp.txt
