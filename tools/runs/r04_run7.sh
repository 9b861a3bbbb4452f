#!/bin/bash
# round 4: channel candidates grouped in runs of 8 adjacent pixels (coherent label changes)
OUT=gpurun_out/r04g; mkdir -p $OUT
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; cat $OUT/channels.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.txt; tail -8 $OUT/gpu_tests.txt
