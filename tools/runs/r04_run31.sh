#!/bin/bash
# round 4: the pooled end of a brick on / off (dbg 2048) by number of poses, look-ahead on
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2,pert2aux,pert4,pert4aux,pert8,pert8aux,pert16aux --variants=-2 --storage q16p --dbg 0,2048) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/pool_by_poses.txt; cat $OUT/pool_by_poses.txt
