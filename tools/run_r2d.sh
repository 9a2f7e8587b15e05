#!/bin/bash
# GPU run D of round 2: persistent fused 128-channel Bottleneck (parity, timing, forward), training-mode forward tests.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout=300 -k "bottleneck_fused or plain_nchw" > $O/tests_bneck.log 2>&1; rc=$?; echo "bneck tests rc=$rc" | tee -a $O/summary.txt; tail -3 $O/tests_bneck.log
if [ $rc -eq 0 ]; then
  timeout 300 python tools/bneck_bench.py 128 > $O/bneck128.log 2>&1; echo "bneck bench rc=$?" | tee -a $O/summary.txt; cat $O/bneck128.log
  timeout 400 python bench.py --no-cpu-baseline --no-f16-leg > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt
  cp gpurun_out/bench_families.json $O/families.json
fi
timeout 900 python -m pytest tests/test_train.py tests/test_gpu_model.py -q --timeout=600 -k "train or dropout or foreign or checkpoint or cfg3_full" > $O/tests_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt; tail -15 $O/tests_new.log | cut -c1-300
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r2d/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["top_shapes"])
except Exception as e: print("ERR", e)
P
