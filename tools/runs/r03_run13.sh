#!/bin/bash
OUT=gpurun_out/r03m; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 400 python tools/brick_bench.py --cases pert32,pert32aux,pert128,pert1,base32,pert8 --variants=0,5 --dbg 512,0 > $OUT/order_product.txt 2>&1; grep -v amdgpu.ids $OUT/order_product.txt | cut -c1-190
timeout 300 python tools/brick_profile.py --cases pert32,pert32aux,pert128 --variants=5 > $OUT/tail.txt 2>&1; grep -v amdgpu.ids $OUT/tail.txt | grep "variant\|lifetime\|barrier wait"
python bench.py > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -3 $OUT/bench_headline.err | grep -v amdgpu
