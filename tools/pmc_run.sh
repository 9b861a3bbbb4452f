#!/bin/bash
# rocprofv3 evidence for one prof_cases.py configuration: kernel-trace stats + separate PMC passes.
# usage: tools/pmc_run.sh <outdir> <prof_cases args...>
set -u
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
run() {  # name, rocprof args...
  local name=$1; shift
  (cd /tmp && timeout 60 rocprofv3 "$@" -d "$ROOT/$OUT/$name" -o "$name" --output-format csv -- python "$ROOT/tools/prof_cases.py" "${PCARGS[@]}") > "$OUT/$name.log" 2>&1
}
PCARGS=("$@")
run trace --kernel-trace --stats
run pmc_valu --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run pmc_lds --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
run pmc_write --pmc WRITE_SIZE
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
find "$OUT" -name "*.csv" | head -50
