#!/bin/bash
# round 4: where does the channel render's time go (phase profile, plain vs channels, same kernel)
OUT=gpurun_out/r04h; mkdir -p $OUT
(timeout 600 python tools/channels_profile.py --real-mask --poses 8; timeout 600 python tools/channels_profile.py --real-mask --poses 1) 2>&1 | grep -v amdgpu.ids > $OUT/channels_profile.txt; cat $OUT/channels_profile.txt
