#!/bin/bash
# round 4: look-ahead with rows worked out ahead (few poses) and lazy staging: stage trace, timings
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 600 python tools/brick_times.py --cases pert1,pert1aux --variant=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/brick_times_few_poses.txt; cat $OUT/brick_times_few_poses.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,8192,4096) 2>&1 | grep -v amdgpu.ids > $OUT/lazy.txt; cat $OUT/lazy.txt
