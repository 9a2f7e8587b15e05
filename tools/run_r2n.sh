#!/bin/bash
# bottleneck128c (4-slot ring of 32-k stages): bit-identity tests, then timings against the shipped kernel
mkdir -p gpurun_out/r2n
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "bottleneck" > gpurun_out/r2n/tests_bneck.log 2>&1
echo "tests rc=$?" | tee gpurun_out/r2n/summary.txt
tail -3 gpurun_out/r2n/tests_bneck.log
timeout 300 python tools/bneck_bench.py 128 ${1:-0,9200,0,9200,9264,9328} > gpurun_out/r2n/bneck128c.log 2>&1
echo "bench rc=$?" | tee -a gpurun_out/r2n/summary.txt
cat gpurun_out/r2n/bneck128c.log
