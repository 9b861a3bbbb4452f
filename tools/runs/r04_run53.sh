#!/bin/bash
# round 4: how many rounds before the launch's end the look-ahead stops claiming (2 / 3 / 4): 512^3 and the 133-slice CT
OUT=gpurun_out/r04am; mkdir -p $OUT
(timeout 600 python tools/brick_bench.py --variants -2 --storage q16p --cases pert1,pert1aux,pert8,pert8aux,pert32,pert32aux --dbg 0,8192,16384,4096) 2>&1 | grep -v amdgpu.ids > $OUT/look_rounds_512.txt; cat $OUT/look_rounds_512.txt | cut -c1-230
(timeout 600 python tools/channels_fwd_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels_fwd.txt; cat $OUT/channels_fwd.txt
(timeout 600 python tools/channels_fwd_bench.py --cube) 2>&1 | grep -v amdgpu.ids > $OUT/channels_fwd_cube.txt; cat $OUT/channels_fwd_cube.txt
