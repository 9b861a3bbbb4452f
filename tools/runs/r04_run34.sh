#!/bin/bash
# round 4: evidence pass 2 on the final kernels -- full GPU tests, smoke, every bench config
OUT=gpurun_out/r04w; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
( time timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err ) 2>&1 | grep real; grep "\[bench\]" $OUT/bench_headline.err | head -12
for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_$c.err; grep "\[bench\] config $c:" $OUT/bench_$c.err; done
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; tail -4 $OUT/channels.txt
ls $OUT
