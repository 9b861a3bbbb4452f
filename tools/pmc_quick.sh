#!/bin/bash
# one PMC pass (VALU counters) + kernel trace for a prof_cases.py configuration
OUT=$1; shift
mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$(pwd)
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace" -o trace --output-format csv -- python "$ROOT/tools/prof_cases.py" "$@") > "$OUT/trace.log" 2>&1
(cd /tmp && timeout 60 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$ROOT/$OUT/pmc_valu" -o pmc_valu --output-format csv -- python "$ROOT/tools/prof_cases.py" "$@") > "$OUT/pmc_valu.log" 2>&1
(cd /tmp && timeout 60 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$ROOT/$OUT/pmc_lds" -o pmc_lds --output-format csv -- python "$ROOT/tools/prof_cases.py" "$@") > "$OUT/pmc_lds.log" 2>&1
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
head -22 "$OUT/summary.txt"
