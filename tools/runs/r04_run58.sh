#!/bin/bash
# round 4: evidence on the final tree -- GPU tests, smoke, every bench config, rocprofv3 stats + PMC traffic, channel timings (both label maps)
OUT=gpurun_out/r04ar; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
( time timeout 900 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err ) 2>&1 | grep real; grep "\[bench\]" $OUT/bench_headline.err | head -12 | cut -c1-230
for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_$c.err; grep "\[bench\] config $c:" $OUT/bench_$c.err | cut -c1-200; done
timeout 900 tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -14 $OUT/rocprof_bench.txt
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "^B \|^# " $OUT/channels.txt | cut -c1-330
ls $OUT $OUT/prof | head -30
