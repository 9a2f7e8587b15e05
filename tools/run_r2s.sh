#!/bin/bash
# 64-channel ring kernel: bit-identity tests, timing vs the weights-resident kernel
mkdir -p gpurun_out/r2s
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "bottleneck" > gpurun_out/r2s/tests_bneck.log 2>&1
echo "tests rc=$?" | tee gpurun_out/r2s/summary.txt
tail -5 gpurun_out/r2s/tests_bneck.log
timeout 300 python tools/bneck_bench.py 64 ${1:-9640,0,9640,0,9601,9602,9604,9608} > gpurun_out/r2s/bneck64r.log 2>&1
cat gpurun_out/r2s/bneck64r.log
