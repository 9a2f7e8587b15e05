#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout=300 -k "bottleneck_fused" > $O/tests_bneck.log 2>&1; rc=$?; echo "bneck tests rc=$rc" | tee -a $O/summary.txt; tail -3 $O/tests_bneck.log
if [ $rc -eq 0 ]; then
  timeout 300 python tools/bneck_bench.py 128 0,9004,0,9004,9014,9024,9044,9084 2>&1 | grep -v amdgpu.ids | tee $O/bneck128_nw4.log
fi
