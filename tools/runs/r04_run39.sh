#!/bin/bash
# round 4: the next unit's image in shares taken by whichever waves finish first: tests, timings, config 4
OUT=gpurun_out/r04z; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert2aux,pert4aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,4096) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/shares.txt; cat $OUT/shares.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err; grep "\[bench\] config 4:" $OUT/c4.err | cut -c1-170; python -c "
import json;d=json.load(open('$OUT/c4.json'));print(round(d['value'],1),'it/s')"
