#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout=300 -k "bottleneck_fused" > $O/tests_bneck.log 2>&1; rc=$?; echo "bneck tests rc=$rc" | tee -a $O/summary.txt; tail -3 $O/tests_bneck.log
if [ $rc -eq 0 ]; then
  timeout 300 python tools/bneck_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe.log
  timeout 300 python tools/bneck_bench.py 128 2>&1 | grep -v amdgpu.ids | tee $O/bneck128.log
  timeout 400 python bench.py --no-cpu-baseline --no-f16-leg > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt
  cp gpurun_out/bench_families.json $O/families.json
  python - <<'P'
import json
d=json.loads(open("gpurun_out/r2f/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["top_shapes"][:3])
P
fi
