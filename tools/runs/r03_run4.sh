#!/bin/bash
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 200 python tools/brick_bench.py --cases pert32,pert32aux,base32,pert1 --variants=0,16,32,18,20 --sqw 8 > $OUT/sq_variants.txt 2>&1; grep -v amdgpu.ids $OUT/sq_variants.txt | cut -c1-175
timeout 200 python tools/brick_bench.py --cases pert32,pert32aux --variants=16,20 --sqw 4,6,10,12 > $OUT/sq_widths.txt 2>&1; grep -v amdgpu.ids $OUT/sq_widths.txt | cut -c1-175
timeout 200 python tools/brick_profile.py --cases pert32,pert32aux --variants=16 > $OUT/phase_profile_sq.txt 2>&1; grep -v amdgpu.ids $OUT/phase_profile_sq.txt
