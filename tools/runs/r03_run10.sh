#!/bin/bash
OUT=gpurun_out/r03j; mkdir -p $OUT
bash tools/pmc_quick.sh $OUT/pmc_fwd --case pert32 --kernel brick --aux 0 --reps 5 > /dev/null 2>&1; cat $OUT/pmc_fwd/summary.txt | head -24
bash tools/pmc_quick.sh $OUT/pmc_aux --case pert32 --kernel brick --aux 1 --reps 5 > /dev/null 2>&1; cat $OUT/pmc_aux/summary.txt | head -24
python bench.py --config 2 > $OUT/bench_config_2.json 2> $OUT/bench_config_2.err; tail -2 $OUT/bench_config_2.err
