#!/bin/bash
OUT=gpurun_out/r03z; mkdir -p $OUT
tools/pmc_quick.sh $OUT/fwd --case pert32 --kernel brick --aux 0 --storage q16p > $OUT/pmc_fwd_q16p.txt 2>&1; cat $OUT/pmc_fwd_q16p.txt
tools/pmc_quick.sh $OUT/aux --case pert32 --kernel brick --aux 1 --storage q16p > $OUT/pmc_fwd_record_q16p.txt 2>&1; cat $OUT/pmc_fwd_record_q16p.txt
rm -rf $OUT/fwd/trace $OUT/fwd/pmc_valu $OUT/fwd/pmc_lds $OUT/aux/trace $OUT/aux/pmc_valu $OUT/aux/pmc_lds
