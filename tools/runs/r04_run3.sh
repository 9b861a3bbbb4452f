#!/bin/bash
# round 4: why the empty-batch test failed on the device; counters for the issue_bound blocks
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_brick_storage.py -x -q -k "empty" 2>&1 | tail -40 > $OUT/empty.txt; cat $OUT/empty.txt
timeout 1500 python tools/issue_bound.py gpurun_out/r04c 2>&1 | grep -v amdgpu.ids | tail -20
