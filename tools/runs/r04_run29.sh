#!/bin/bash
# round 4: lazy staging debug
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 300 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert32aux --variants=-2 --storage q16p --dbg 8192) 2>&1 | grep -v amdgpu.ids | cut -c1-230 > $OUT/dbg_a.txt; cat $OUT/dbg_a.txt
(timeout 300 python tools/brick_bench.py --cases pert1 --variants=-2 --storage q16p --dbg 0) 2>&1 | grep -v amdgpu.ids | cut -c1-230 | tail -5 > $OUT/dbg_b.txt; cat $OUT/dbg_b.txt
