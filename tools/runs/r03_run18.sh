#!/bin/bash
OUT=gpurun_out/r03r; mkdir -p $OUT
timeout 600 python tools/brick_bench.py --variants 0,1 --order weight --dbg 0,4096,8192,12288 --cases pert32,pert32aux,pert8,pert128 2>&1 | grep -v amdgpu > $OUT/pool.txt; cat $OUT/pool.txt
timeout 300 python tools/brick_profile.py --variants 1 --cases pert32,pert32aux 2>&1 | grep -v amdgpu > $OUT/prof.txt; cat $OUT/prof.txt
python -m pytest tests -m gpu -x -q -k "brick or baseline or config or q16" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
