#!/bin/bash
# round 4: per-brick durations of a single-pose launch (what a brick costs when almost nothing is walked)
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 600 python tools/brick_times.py --cases pert1,pert1aux,pert32aux --variant=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/brick_times_few_poses.txt; cat $OUT/brick_times_few_poses.txt
