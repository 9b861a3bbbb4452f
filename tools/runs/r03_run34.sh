#!/bin/bash
OUT=gpurun_out/r03ah; mkdir -p $OUT
python -m pytest tests/test_gpu_baseline_sizes.py -m gpu -x -q -k "properties or published" > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
