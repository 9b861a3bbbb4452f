#!/bin/bash
# round 4: phase profile of the sorted walk
OUT=gpurun_out/r04n; mkdir -p $OUT
(DDRR_SORTED=8 timeout 600 python tools/brick_profile.py --cases pert32,pert32aux --storage q16p; timeout 600 python tools/brick_profile.py --cases pert32 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/sorted_profile.txt; cat $OUT/sorted_profile.txt
