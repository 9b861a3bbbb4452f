#!/bin/bash
OUT=gpurun_out/r03ap; mkdir -p $OUT
(timeout 300 python tools/sparse_bench.py; timeout 300 python tools/packed_bench.py product) 2>&1 | grep -v amdgpu > $OUT/tools.txt; cat $OUT/tools.txt
