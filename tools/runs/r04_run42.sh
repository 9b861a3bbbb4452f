#!/bin/bash
# round 4: length-class thresholds of the forward + record kernel (grouped runs of 8 pixels) and the forward kernel
OUT=gpurun_out/r04ac; mkdir -p $OUT
(timeout 1200 python tools/brick_bench.py --cases pert32aux,pert32 --variants=-2 --storage q16p --order weight --classes 14:32,16:36,18:40,20:44,22:48,24:54,18:36,18:44,20:40,16:40) 2>&1 | grep -v amdgpu.ids | cut -c1-40,100-200 > $OUT/classes.txt; cat $OUT/classes.txt
