#!/bin/bash
# round 4: the next brick requested ahead (claim, lookups, image into registers): parity, then timings at 1..32 poses
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,8192,4096) 2>&1 | grep -v amdgpu.ids > $OUT/look_ahead.txt; cat $OUT/look_ahead.txt
