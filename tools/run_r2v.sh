#!/bin/bash
# XCD-aware tile assignment in the tiled Bottleneck kernels: tests, micro timings, one bench
mkdir -p gpurun_out/r2v
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "bottleneck" > gpurun_out/r2v/tests_bneck.log 2>&1
echo "tests rc=$?" | tee gpurun_out/r2v/summary.txt
tail -3 gpurun_out/r2v/tests_bneck.log
timeout 300 python tools/bneck_bench.py 128 0,0,0 > gpurun_out/r2v/b128.log 2>&1; cat gpurun_out/r2v/b128.log
timeout 300 python tools/bneck_bench.py 64 0,0,0 > gpurun_out/r2v/b64.log 2>&1; cat gpurun_out/r2v/b64.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg > gpurun_out/r2v/bench.json 2> gpurun_out/r2v/bench.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2v/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [s for s in d["roofline"]["top_shapes"] if "bneck" in s["shape"]])
PY
