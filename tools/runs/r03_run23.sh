#!/bin/bash
OUT=gpurun_out/r03w; mkdir -p $OUT
timeout 300 python tools/brick_profile.py --variants -2 --storage q16p --cases pert32,pert32aux,pert1,pert8 2>&1 | grep -v amdgpu > $OUT/prof.txt; cat $OUT/prof.txt
