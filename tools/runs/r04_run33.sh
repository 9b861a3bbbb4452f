#!/bin/bash
# round 4: evidence pass 1 on the final kernels -- rocprofv3 stats + PMC traffic of the bench command, issue-bound counters
OUT=gpurun_out/r04v; mkdir -p $OUT
timeout 900 tools/prof_bench.sh $OUT/prof > $OUT/rocprof_bench.txt 2>&1; head -14 $OUT/rocprof_bench.txt
timeout 1500 python tools/issue_bound.py $OUT 2>&1 | grep -v amdgpu.ids | tail -3
ls $OUT $OUT/prof | head -30
