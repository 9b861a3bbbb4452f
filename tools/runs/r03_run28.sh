#!/bin/bash
OUT=gpurun_out/r03ab; mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "channels or mask or general" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 300 python tools/channels_bench.py --real-mask 2>&1 | grep -v amdgpu > $OUT/channels_real.txt; cat $OUT/channels_real.txt
