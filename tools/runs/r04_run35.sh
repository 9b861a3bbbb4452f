#!/bin/bash
# round 4: the look-ahead at large batches (claims held ahead against the balance of the launch's end)
OUT=gpurun_out/r04w; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert64,pert128,pert512,pert64aux,pert128aux --variants=-2 --storage q16p --order weight --dbg 0,4096) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/look_ahead_large.txt; cat $OUT/look_ahead_large.txt
