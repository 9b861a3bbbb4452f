#!/bin/bash
# round 3, GPU call 2: brick variants (storage, shape, workgroups per CU)
OUT=gpurun_out/r03b; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/brick_bench.py --cases pert32,pert32aux,base32,pert1 --variants=-1,0,1,2,3,4,5,6,7,8,9 > $OUT/brick_variants.txt 2>&1; grep -v amdgpu.ids $OUT/brick_variants.txt
