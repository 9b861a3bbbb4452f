#!/bin/bash
# round 4: labels staged with aligned dword loads on the unaligned path
OUT=gpurun_out/r04i; mkdir -p $OUT
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; cat $OUT/channels.txt
timeout 600 python tools/channels_profile.py --real-mask --poses 8 2>&1 | grep -v amdgpu.ids > $OUT/channels_profile.txt; cat $OUT/channels_profile.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.txt; tail -5 $OUT/gpu_tests.txt
