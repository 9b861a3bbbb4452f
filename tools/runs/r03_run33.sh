#!/bin/bash
OUT=gpurun_out/r03ag; mkdir -p $OUT
(timeout 300 python tools/packed_bench.py 0; timeout 300 python tools/packed_bench.py 4096) 2>&1 | grep -v amdgpu | grep "flags\|B   32\|B  128\|B    8\|B    1 " > $OUT/ticket_ahead.txt; cat $OUT/ticket_ahead.txt
