#!/bin/bash
OUT=gpurun_out/r03af; mkdir -p $OUT
timeout 300 python tools/brick_profile.py --variants 11 --cases pert32,pert32aux 2>&1 | grep -v amdgpu > $OUT/prof.txt; cat $OUT/prof.txt
