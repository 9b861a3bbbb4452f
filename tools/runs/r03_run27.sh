#!/bin/bash
OUT=gpurun_out/r03aa; mkdir -p $OUT
timeout 300 python tools/channels_bench.py --real-mask 2>&1 | grep -v amdgpu > $OUT/channels_real.txt; cat $OUT/channels_real.txt
