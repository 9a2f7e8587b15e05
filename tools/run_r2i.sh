#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout=300 -k "bottleneck_fused" > $O/tests_bneck.log 2>&1; rc=$?; echo "bneck tests rc=$rc" | tee -a $O/summary.txt; tail -3 $O/tests_bneck.log
if [ $rc -eq 0 ]; then
  timeout 300 python tools/bneck_bench.py 128 2>&1 | grep -v amdgpu.ids | tee $O/bneck128_two_per_cu.log
  CFT_BNECK128=persistent timeout 300 python tools/bneck_bench.py 128 0 2>&1 | grep -v amdgpu.ids | tee $O/bneck128_persistent.log
  CFT_FUSE128=1 timeout 400 python bench.py --no-cpu-baseline --no-f16-leg > $O/bench_fuse128.json 2> $O/bench1.log; echo "bench fuse128 rc=$?" | tee -a $O/summary.txt
  cp gpurun_out/bench_families.json $O/families_fuse128.json
  timeout 400 python bench.py --no-cpu-baseline --no-f16-leg > $O/bench_default.json 2> $O/bench0.log; echo "bench default rc=$?" | tee -a $O/summary.txt
  python - <<'P'
import json
for f in ("bench_fuse128","bench_default"):
    d=json.loads(open(f"gpurun_out/r2i/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"], [(t["shape"],t["ms"]) for t in d["roofline"]["top_shapes"][:3]])
P
fi
