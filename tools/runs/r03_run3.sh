#!/bin/bash
OUT=gpurun_out/r03c; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/brick_bench.py --cases pert32,pert32aux,base32,pert1,pert128 --variants=0,1,2,4,5,6 > $OUT/brick_variants.txt 2>&1; grep -v amdgpu.ids $OUT/brick_variants.txt
python tools/brick_profile.py --cases pert32,pert32aux --variants=0,1,2,4 > $OUT/phase_profile.txt 2>&1; grep -v amdgpu.ids $OUT/phase_profile.txt
