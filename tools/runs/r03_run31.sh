#!/bin/bash
OUT=gpurun_out/r03ae; mkdir -p $OUT
timeout 600 python tools/brick_bench.py --variants 5,11,0,12 --order weight --cases pert32,pert32aux,pert8,pert128,base32 2>&1 | grep -v amdgpu > $OUT/c6.txt; cat $OUT/c6.txt
timeout 300 python tools/brick_profile.py --variants 5,11 --cases pert32,pert32aux 2>&1 | grep -v amdgpu > $OUT/prof.txt; grep "variant\|wave-steps\|hits per batch\|walk   \|barrier wait\|phase A \|unit pull\|batch pop" $OUT/prof.txt
