#!/bin/bash
# round 4: any D.z staged as quads (both brick kernels), look-ahead stops 3 rounds before the end: GPU tests, the 133-slice CT, headline
OUT=gpurun_out/r04an; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
(timeout 600 python tools/channels_fwd_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/example_ct_kernels.txt; cat $OUT/example_ct_kernels.txt
timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 20 --warmup 5 > $OUT/h.json 2> $OUT/h.err; grep "config headline:" $OUT/h.err | cut -c1-200
