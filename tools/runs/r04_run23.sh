#!/bin/bash
# round 4: lazy staging of few-unit bricks on / off (dbg 8192) / no look-ahead at all (dbg 4096)
OUT=gpurun_out/r04t; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux --variants=-2 --storage q16p --dbg 0,8192,4096) 2>&1 | grep -v amdgpu.ids > $OUT/lazy.txt; cat $OUT/lazy.txt
(timeout 900 python -m pytest tests -m gpu -x -q -k "brick or registration or baseline" 2>&1 | tail -4) > $OUT/tests2.txt; cat $OUT/tests2.txt
