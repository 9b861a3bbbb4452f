#!/bin/bash
# round 4: the channel render's ray backward on the bricks
OUT=gpurun_out/r04k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_brick_storage.py -x -q -k "channel" 2>&1 | tail -15 > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 600 python tools/channels_bench.py; timeout 600 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "backward\|entry points\|^#" $OUT/channels.txt
