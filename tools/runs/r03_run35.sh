#!/bin/bash
OUT=gpurun_out/r03ai; mkdir -p $OUT
(for f in 0 8 1024 1032; do timeout 300 python tools/packed_bench.py $f; done) 2>&1 | grep -v amdgpu | grep "flags\|B   32 fwd+record\|B  128 fwd+record\|B    8 fwd+record" > $OUT/aux_grouping.txt; cat $OUT/aux_grouping.txt
