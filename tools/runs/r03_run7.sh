#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=0,5 --dbg 0,256 > $OUT/no_ray_loads.txt 2>&1; grep -v amdgpu.ids $OUT/no_ray_loads.txt | cut -c1-175
DDRR_EXP_FLAGS="-DDDRR_WALK_CHECK4" timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=0,5 > $OUT/check4.txt 2>&1; grep -v amdgpu.ids $OUT/check4.txt | cut -c1-175
bash tools/prof_bench.sh $OUT/prof > $OUT/prof_bench.txt 2>&1; cat $OUT/prof_bench.txt | cut -c1-200
