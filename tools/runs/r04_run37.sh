#!/bin/bash
# round 4: the registration config with every brick storage (the phantom has 644 of 2048 bricks on the fp32 path)
OUT=gpurun_out/r04x; mkdir -p $OUT
for st in q16p q16 f32; do timeout 600 python bench.py --config 4 --no-cpu-baseline --storage $st > $OUT/c4_$st.json 2> $OUT/c4_$st.err; grep "\[bench\] config 4:" $OUT/c4_$st.err | cut -c1-160; python -c "
import json;d=json.load(open('$OUT/c4_$st.json'));print('$st', round(d['value'],1),'it/s', d['config'].get('brick_storage_fallbacks'))"; done
