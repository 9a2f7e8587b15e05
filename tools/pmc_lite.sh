#!/bin/bash
# Two quick PMC passes (SQ activity + LDS) for a command:  tools/pmc_lite.sh <outdir> -- <command...>
set -u
OUT=$1; shift; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD=("$@")
cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" --output-format csv -- "${CMD[@]}" > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU
