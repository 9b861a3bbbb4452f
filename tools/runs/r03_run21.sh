#!/bin/bash
OUT=gpurun_out/r03u; mkdir -p $OUT
timeout 600 python tools/packed_bench.py 2>&1 | grep -v amdgpu > $OUT/packed.txt; cat $OUT/packed.txt
