#!/bin/bash
# round 4: evidence pass -- full GPU tests, smoke, every bench config, rocprofv3 stats + PMC traffic, issue-bound counters
OUT=gpurun_out/r04l; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python bench.py > $OUT/bench_config_headline.json 2> $OUT/bench_headline.err; grep "\[bench\]" $OUT/bench_headline.err | head -3
for c in 2 3 4 5; do timeout 600 python bench.py --config $c > $OUT/bench_config_$c.json 2> $OUT/bench_$c.err; grep "\[bench\] config $c:" $OUT/bench_$c.err; done
timeout 900 tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -14 $OUT/rocprof_bench.txt
timeout 1500 python tools/issue_bound.py $OUT 2>&1 | grep -v amdgpu.ids | tail -3
ls $OUT
