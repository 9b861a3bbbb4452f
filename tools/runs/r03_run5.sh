#!/bin/bash
OUT=gpurun_out/r03e; mkdir -p $OUT
timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=5 --classes 18:40,22:48,26:56,30:64,22:56 > $OUT/classes_z64.txt 2>&1; grep -v amdgpu.ids $OUT/classes_z64.txt | cut -c1-175
