#!/bin/bash
# round 4: the sorted walk without any delivery (timing experiment): are the delivery atomics what its batches wait for?
OUT=gpurun_out/r04p; mkdir -p $OUT
(DDRR_EXP_FLAGS="-DDDRR_NO_DELIVER" timeout 600 python tools/brick_bench.py --cases pert32,pert32aux --order weight --sorted 0,8) 2>&1 | grep -v amdgpu.ids > $OUT/no_deliver.txt; cat $OUT/no_deliver.txt
