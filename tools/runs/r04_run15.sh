#!/bin/bash
# round 4: the sorted walk, sort pass and batch entries software-pipelined
OUT=gpurun_out/r04o; mkdir -p $OUT
timeout 900 python tools/brick_bench.py --cases pert32,pert32aux,pert8,pert128,pert1 --order weight --sorted 0,8,5 2>&1 | grep -v amdgpu.ids > $OUT/sorted_walk.txt; cat $OUT/sorted_walk.txt
(DDRR_SORTED=8 timeout 600 python tools/brick_profile.py --cases pert32,pert32aux --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/sorted_profile.txt; cat $OUT/sorted_profile.txt
