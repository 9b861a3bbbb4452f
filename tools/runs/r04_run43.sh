#!/bin/bash
# round 4: the channel render's volume gradient on the bricks: tests, timings, fuzz
OUT=gpurun_out/r04ad; mkdir -p $OUT
(timeout 900 python -m pytest tests -m gpu -x -q -k "channel" 2>&1 | tail -8) > $OUT/tests.txt; cat $OUT/tests.txt
(timeout 600 python tools/channels_bench.py) 2>&1 | grep -v amdgpu.ids > $OUT/channels.txt; grep "VOLUME\|channel backward, rays" $OUT/channels.txt | cut -c1-330
