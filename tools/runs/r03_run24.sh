#!/bin/bash
OUT=gpurun_out/r03x; mkdir -p $OUT
(timeout 300 python tools/packed_bench.py 0; timeout 300 python tools/packed_bench.py 4096) 2>&1 | grep -v amdgpu > $OUT/prefetch.txt; cat $OUT/prefetch.txt
python -m pytest tests -m gpu -x -q -k "brick or baseline or config or q16 or sweep" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
