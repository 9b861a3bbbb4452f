#!/bin/bash
# round 4: the record on accumulated alphas with a tie window (24 instructions per step) against plane counters (27)
OUT=gpurun_out/r04aa; mkdir -p $OUT
(timeout 900 python tools/brick_bench.py --cases pert32aux,pert1aux,pert8aux --variants=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/rec_new.txt; cat $OUT/rec_new.txt
(DDRR_EXP_FLAGS="-DDDRR_RECORD_PLANE_COUNTERS" timeout 900 python tools/brick_bench.py --cases pert32aux,pert1aux,pert8aux --variants=-2 --storage q16p) 2>&1 | grep -v amdgpu.ids | cut -c1-60,100-230 > $OUT/rec_old.txt; cat $OUT/rec_old.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $OUT/tests.txt; cat $OUT/tests.txt
