#!/bin/bash
# round 4: few-pose launches: the product's packed 16-bit bricks against two 32^3 16-bit workgroups per CU
OUT=gpurun_out/r04s; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32aux --variants=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/few_poses_q16p.txt; cat $OUT/few_poses_q16p.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32aux --variants=10,1) 2>&1 | grep -v amdgpu.ids > $OUT/few_poses_q16.txt; cat $OUT/few_poses_q16.txt
