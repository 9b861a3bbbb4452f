#!/bin/bash
OUT=gpurun_out/r03k; mkdir -p $OUT
timeout 300 python tools/brick_profile.py --cases pert32,pert32aux,pert128 --variants=0,5 > $OUT/tail.txt 2>&1; grep -v amdgpu.ids $OUT/tail.txt | grep "variant\|lifetime\|barrier wait"
