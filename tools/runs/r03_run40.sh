#!/bin/bash
OUT=gpurun_out/r03an; mkdir -p $OUT
timeout 300 python tools/sparse_bench.py 2>&1 | grep -v amdgpu > $OUT/sparse.txt; cat $OUT/sparse.txt
