#!/bin/bash
# round 4: fp32 bricks -- the general kernel (quad staging) against the configurable kernel's 32^3 fp32 variant, 256^3 and 512^3
OUT=gpurun_out/r04ao; mkdir -p $OUT
for size in 256 512; do
(timeout 600 python tools/brick_bench.py --size $size --variants=-1,0 --storage f32 --cases pert1,pert1aux,pert8,pert8aux,pert32,pert32aux) 2>&1 | grep -v amdgpu.ids > $OUT/f32_general_vs_cfg_$size.txt; cut -c1-200 $OUT/f32_general_vs_cfg_$size.txt
done
