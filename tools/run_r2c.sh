#!/bin/bash
# GPU run C of round 2: fused 128-channel Bottleneck (parity, timing, A/B in the forward), NMS v2, fixed tests.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout=300 -k "bottleneck_fused or plain_nchw" > $O/tests_bneck.log 2>&1; rc=$?; echo "bneck tests rc=$rc" | tee -a $O/summary.txt; tail -3 $O/tests_bneck.log
timeout 600 python -m pytest tests/test_nms.py -q --timeout=300 > $O/tests_nms.log 2>&1; echo "nms tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/tests_nms.log
timeout 300 python tools/nms_bench.py > $O/nms.log 2>&1; echo "nms bench rc=$?" | tee -a $O/summary.txt; tail -3 $O/nms.log
if [ $rc -eq 0 ]; then
  timeout 300 python tools/bneck_bench.py 128 > $O/bneck128.log 2>&1; echo "bneck bench rc=$?" | tee -a $O/summary.txt; cat $O/bneck128.log
  timeout 400 python bench.py --no-cpu-baseline --no-f16-leg > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt
  cp gpurun_out/bench_families.json $O/families.json
fi
timeout 900 python -m pytest tests/test_gpu_model.py -q --timeout=600 > $O/tests_model.log 2>&1; echo "model tests rc=$?" | tee -a $O/summary.txt; tail -5 $O/tests_model.log
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r2c/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["top_shapes"])
except Exception as e: print("ERR", e)
P
