#!/bin/bash
# round 4: final randomised sweeps of every brick kernel against the per-ray kernels
OUT=gpurun_out/r04ai; mkdir -p $OUT
for seed in 22; do (timeout 600 python tools/fuzz_bricks.py --cases 24 --seed $seed) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_$seed.txt; echo "seed $seed flagged: $(grep -c '<<<' $OUT/fuzz_$seed.txt)"; tail -1 $OUT/fuzz_$seed.txt; done
(timeout 600 python tools/fuzz_bricks.py --cases 32 --seed 23 --smooth) 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_smooth.txt; echo "smooth flagged: $(grep -c '<<<' $OUT/fuzz_smooth.txt)"; tail -1 $OUT/fuzz_smooth.txt
grep "<<<" $OUT/fuzz_2*.txt $OUT/fuzz_smooth.txt | cut -c1-330 | head -20
