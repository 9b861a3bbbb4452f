#!/bin/bash
# round 3, GPU call 1: blocked record (tests, kernel timing with / without the lane swap, PMC, bench)
OUT=gpurun_out/r03a; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/brick_bench.py --cases pert32,pert32aux --dbg 0,128,64 --classes 18:40 > $OUT/brick_bench.txt 2>&1; cat $OUT/brick_bench.txt | grep -v amdgpu.ids
bash tools/pmc_run.sh $OUT/pmc_aux --case pert32 --kernel brick --aux 1 > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pmc_aux > $OUT/pmc_aux_summary.txt 2>&1; head -40 $OUT/pmc_aux_summary.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
