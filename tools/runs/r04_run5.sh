#!/bin/bash
# round 4: can better brick weights close the gap between the launch and sum / 256 ?
OUT=gpurun_out/r04e; mkdir -p $OUT
timeout 900 python tools/brick_weights.py 2>&1 | grep -v amdgpu.ids > $OUT/brick_weights.txt; cat $OUT/brick_weights.txt
