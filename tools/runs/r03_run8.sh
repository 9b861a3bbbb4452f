#!/bin/bash
OUT=gpurun_out/r03h; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
DDRR_EXP_FLAGS="-DDDRR_WALK_CHECK4" timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=0,5 > $OUT/check4.txt 2>&1; grep -v amdgpu.ids $OUT/check4.txt | cut -c1-175
timeout 300 python tools/brick_bench.py --cases pert32,pert32aux,pert1 --variants=0,5 > $OUT/check2.txt 2>&1; grep -v amdgpu.ids $OUT/check2.txt | cut -c1-175
timeout 300 python tools/channels_bench.py > $OUT/channels_synthetic.txt 2>&1; grep -v amdgpu.ids $OUT/channels_synthetic.txt
timeout 300 python tools/channels_bench.py --real-mask > $OUT/channels_real_mask.txt 2>&1; grep -v amdgpu.ids $OUT/channels_real_mask.txt
