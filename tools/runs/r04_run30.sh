#!/bin/bash
# round 4: look-ahead x {pool, second row table}: which combination is fastest at 1 / 8 / 32 poses
OUT=gpurun_out/r04t; mkdir -p $OUT; : > $OUT/matrix.txt
for flags in "" "-DDDRR_NO_POOL" "-DDDRR_NO_ROWS2" "-DDDRR_NO_POOL -DDDRR_NO_ROWS2"; do
  echo "### build flags: [$flags]" >> $OUT/matrix.txt
  (DDRR_EXP_FLAGS="$flags" timeout 900 python tools/brick_bench.py --cases pert1,pert1aux,pert8aux,pert32,pert32aux --variants=-2 --storage q16p --dbg 0,4096) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 >> $OUT/matrix.txt
done
cat $OUT/matrix.txt
