#!/bin/bash
# Collect rocprofv3 PMC counters for a command, one counter group per pass (gfx950 slot limits:
# SQ 8, TCC 4, FETCH_SIZE costs 3, WRITE_SIZE 2 - MI355X_MICROARCH.md).  Counter passes are run
# WITHOUT any trace option other than --kernel-trace.
#   tools/pmc.sh <outdir> -- <command...>
set -u
OUT=$1; shift; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" --output-format csv -- "${CMD[@]}" > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
CMD=("$@")
cd "$GRAFT_REPO_ROOT"
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr
find "$OUT" -name "*counter_collection.csv" | head
