#!/bin/bash
# Memory-system PMC passes (L2 <-> fabric request counts, queue levels, stalls) for a command:
#   tools/pmc_mem.sh <outdir> -- <command...>
# RDREQ_LEVEL / RDREQ = average read latency in cycles; LEVEL / busy cycles = requests in flight.
set -u
OUT=$1; shift; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD=("$@")
cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" --output-format csv -- "${CMD[@]}" > "$OUT/$name.log" 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
