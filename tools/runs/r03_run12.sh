#!/bin/bash
OUT=gpurun_out/r03l; mkdir -p $OUT
timeout 400 python tools/brick_bench.py --cases pert32,pert32aux,pert128 --variants=5 --order id,center,weight --split 0:1,256:2,512:4,1024:4 > $OUT/order_split.txt 2>&1; grep -v amdgpu.ids $OUT/order_split.txt | cut -c1-190
