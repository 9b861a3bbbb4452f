#!/bin/bash
# round 4: few-pose launches (registration: B = 1): brick variants -- does a second workgroup per CU hide the staging latency?
OUT=gpurun_out/r04s; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux --variants=-2,10,2,0,3) 2>&1 | grep -v amdgpu.ids > $OUT/few_poses_variants.txt; cat $OUT/few_poses_variants.txt
