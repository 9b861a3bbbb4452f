#!/bin/bash
# round 4: forward-only walks enter a brick with the plain quotients (n * rcp(d), floor cell): GPU tests, forward timings, randomised sweep, headline
OUT=gpurun_out/r04av; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $OUT/gpu_tests.txt; grep "^FAILED\|passed\|failed\|^E   *assert" $OUT/gpu_tests.txt | cut -c1-250
(timeout 600 python tools/brick_bench.py --variants=-2 --storage q16p --cases pert1,pert8,pert32,pert32aux,pert128) 2>&1 | grep -v amdgpu.ids > $OUT/fwd_plain_entry.txt; cut -c1-200 $OUT/fwd_plain_entry.txt
(timeout 1200 python tools/fuzz_bricks.py --cases 128 --seed 11) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_bricks.txt; echo "flagged: $(grep -c "<<<" $OUT/fuzz_bricks.txt)"; tail -1 $OUT/fuzz_bricks.txt | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline > $OUT/h.json 2> $OUT/h.err; grep "config headline\|config 2\|config 5" $OUT/h.err | cut -c1-200
