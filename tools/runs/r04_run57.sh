#!/bin/bash
# round 4: bricks out of every pose's view passed over where they are claimed (dbg 32768: off): tests, 512^3 at 1 / 8 / 32 poses, the example CT, registration, headline
OUT=gpurun_out/r04aq; mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -q -x -k "look_ahead or any_depth or channel or storage or bricks or registration" 2>&1 | tail -5) > $OUT/gpu_tests_subset.txt; cat $OUT/gpu_tests_subset.txt
(timeout 600 python tools/brick_bench.py --variants=-2 --storage q16p --cases pert1,pert1aux,pert8,pert8aux,pert32,pert32aux --dbg 0,32768) 2>&1 | grep -v amdgpu.ids > $OUT/in_view_512.txt; cut -c1-200 $OUT/in_view_512.txt
(timeout 600 python tools/brick_bench.py --variants=-1 --storage f32 --cases pert1,pert1aux --dbg 0,32768) 2>&1 | grep -v amdgpu.ids > $OUT/in_view_512_general.txt; cut -c1-200 $OUT/in_view_512_general.txt
timeout 600 python bench.py --config 4 --no-cpu-baseline > $OUT/bench_config_4.json 2> $OUT/c4.err; grep "config 4:" $OUT/c4.err | cut -c1-200
timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 20 --warmup 5 > $OUT/h.json 2> $OUT/h.err; grep "config headline:" $OUT/h.err | cut -c1-200
