#!/bin/bash
# PMC counters of the 128-channel Bottleneck kernel (ring kernel = default, and the 16-KiB-K-tile kernel)
mkdir -p gpurun_out/r2p
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2p
timeout 900 bash tools/pmc.sh $OUT/c -- python tools/bneck_bench.py 128 0 > $OUT/pmc_c.log 2>&1
python tools/pmc_summary.py $OUT/c bottleneck128 > $OUT/pmc_c_summary.txt 2>&1
cat $OUT/pmc_c_summary.txt
rm -rf $OUT/c/*/  # raw csv dirs are large
