#!/bin/bash
# chunk-major K walk for wide 3x3 convs: op tests, A/B timing (variant 6427 = tap-major 256x256), PMC hit rate
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$GRAFT_REPO_ROOT/gpurun_out/r2x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "conv or bottleneck or variant or tile" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 300 python tools/gemm_bench.py --variants 6427,0,6427,0 --iters 20 --only "3x3" --out r2x_gemm.json 2>&1 | grep -v amdgpu.ids | cut -c1-400
