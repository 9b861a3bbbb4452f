#!/bin/bash
# round 4: the marcher's channel backward on the bricks: parity on the device, timings
OUT=gpurun_out/r04q; mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -x -q -k "trilinear or channel" 2>&1 | tail -8) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; cat $OUT/channels.txt
