#!/bin/bash
OUT=gpurun_out/r03aj; mkdir -p $OUT
tools/prof_bench.sh $OUT > $OUT/rocprof_bench.txt 2>&1; head -40 $OUT/rocprof_bench.txt
cat $OUT/bench_line_under_trace.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else d['roofline'])"
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_bench_kernel_stats.csv \;
rm -rf $OUT/bench_trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_HIT_sum
