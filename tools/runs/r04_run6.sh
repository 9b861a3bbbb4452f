#!/bin/bash
# round 4: channel walk on the accumulating scheme + lean flush; the general kernel's forward on the accumulating walk
OUT=gpurun_out/r04f; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.txt; tail -8 $OUT/gpu_tests.txt
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; cat $OUT/channels.txt
