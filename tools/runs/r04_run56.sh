#!/bin/bash
# round 4: one pose, fp32 bricks: where the general kernel's 7-12 % over the configurable kernel's 32^3 variant come from (phase profile), and the packed 16-bit bricks beside them
OUT=gpurun_out/r04ap; mkdir -p $OUT
(timeout 600 python tools/brick_profile.py --variants=-1,0 --storage f32 --cases pert1,pert1aux; timeout 600 python tools/brick_profile.py --variants=-2 --storage q16p --cases pert1,pert1aux) 2>&1 | grep -v amdgpu.ids > $OUT/phase_profile_one_pose.txt; cat $OUT/phase_profile_one_pose.txt
