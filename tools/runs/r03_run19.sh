#!/bin/bash
OUT=gpurun_out/r03s; mkdir -p $OUT
(echo "## heaviest first"; timeout 300 python tools/volgrad_bench.py 0; echo "## id order"; timeout 300 python tools/volgrad_bench.py 512) 2>&1 | grep -v amdgpu > $OUT/volgrad.txt; cat $OUT/volgrad.txt
(echo "## heaviest first"; timeout 300 python tools/channels_bench.py; echo "## real mask"; timeout 300 python tools/channels_bench.py --real-mask) 2>&1 | grep -v amdgpu > $OUT/channels.txt; cat $OUT/channels.txt
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
