#!/bin/bash
OUT=gpurun_out/r03n; mkdir -p $OUT
timeout 300 python tools/brick_profile.py --cases pert32,pert32aux,pert128 --variants=5 > $OUT/span.txt 2>&1; grep -v amdgpu.ids $OUT/span.txt | grep "variant\|launch span"
