#!/bin/bash
# PMC counters of the 256x256 GEMM kernel on its two largest line items (3x3 256->256 @40, 1x1 256->256 @40)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2w; mkdir -p $OUT
timeout 600 bash tools/pmc.sh $OUT/k3 -- python tools/gemm_bench.py --variants 0 --iters 10 --only "bneck 3x3 256->256 @40" --out r2w_k3.json > $OUT/pmc_k3.log 2>&1
python tools/pmc_summary.py $OUT/k3 conv_gemm > $OUT/pmc_k3_summary.txt 2>&1
timeout 600 bash tools/pmc.sh $OUT/k1 -- python tools/gemm_bench.py --variants 0 --iters 10 --only "bneck 1x1 256->256 @40" --out r2w_k1.json > $OUT/pmc_k1.log 2>&1
python tools/pmc_summary.py $OUT/k1 conv_gemm > $OUT/pmc_k1_summary.txt 2>&1
rm -rf $OUT/k3/*/ $OUT/k1/*/
cat $OUT/pmc_k3_summary.txt $OUT/pmc_k1_summary.txt
