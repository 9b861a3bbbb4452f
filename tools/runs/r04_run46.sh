#!/bin/bash
# round 4: bench with events on the dominant kernel only inside the timed region
OUT=gpurun_out/r04af; mkdir -p $OUT
( time timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err ) 2>&1 | grep real; grep "\[bench\]" $OUT/bench_headline.err | head -12 | cut -c1-230
for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_$c.err; grep "\[bench\] config $c:" $OUT/bench_$c.err | cut -c1-200; done
