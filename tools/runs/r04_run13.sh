#!/bin/bash
# round 4: the sorted walk (hits of a round counting-sorted by length class through global scratch) against the queues
OUT=gpurun_out/r04m; mkdir -p $OUT
timeout 900 python tools/brick_bench.py --cases pert32,pert32aux,pert8,pert128,pert1 --order weight --sorted 0,8,6,12 2>&1 | grep -v amdgpu.ids > $OUT/sorted_walk.txt; cat $OUT/sorted_walk.txt
