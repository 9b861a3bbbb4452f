#!/bin/bash
# round 4: where a single-pose launch spends its wave time (registration, B = 1)
OUT=gpurun_out/r04s; mkdir -p $OUT
(timeout 600 python tools/brick_profile.py --cases pert1aux,pert1,pert4aux --storage q16p) 2>&1 | grep -v amdgpu.ids > $OUT/phase_profile_few_poses.txt; cat $OUT/phase_profile_few_poses.txt
