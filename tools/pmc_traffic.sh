#!/bin/bash
# HBM traffic of one bench forward: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on
# gfx950), counters only + --kernel-trace.   tools/pmc_traffic.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d "$OUT/$c" --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16-leg --no-graph --sustained-steps 0 --no-parity "$@" > "$OUT/$c.log" 2>&1
  echo "$c rc=$?"
done
rm -rf "$OUT"/*/*/*.db
