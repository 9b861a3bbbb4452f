#!/bin/bash
OUT=gpurun_out/r04x; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_brick_storage.py -m gpu -x -q -k look_ahead 2>&1 | tail -15) > $OUT/t.txt; cat $OUT/t.txt
