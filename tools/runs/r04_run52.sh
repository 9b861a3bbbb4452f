#!/bin/bash
# round 4: any D.z on the forward kernel's bricks + its channel mode: new tests, channel tests, A/B timings, headline check
OUT=gpurun_out/r04al; mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -k "any_depth or channel or storage or look_ahead or bricks" 2>&1 | tail -15) > $OUT/gpu_tests_subset.txt; cat $OUT/gpu_tests_subset.txt
(timeout 600 python tools/channels_fwd_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels_fwd.txt; cat $OUT/channels_fwd.txt
timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 20 --warmup 5 > $OUT/h.json 2> $OUT/h.err; grep "config headline:" $OUT/h.err | cut -c1-200
