#!/bin/bash
# bottleneck128c as the default: op tests, micro timing, A/B of the whole forward (CFT_BNECK128=b = the previous kernel)
mkdir -p gpurun_out/r2o
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "bottleneck" > gpurun_out/r2o/tests_bneck.log 2>&1
echo "tests rc=$?" | tee gpurun_out/r2o/summary.txt
tail -3 gpurun_out/r2o/tests_bneck.log
timeout 300 python tools/bneck_bench.py 128 9100,0,9100,0,9201,9202,9204,9208,9456,9712 > gpurun_out/r2o/bneck128c.log 2>&1
cat gpurun_out/r2o/bneck128c.log
for rep in 1 2; do
  CFT_BNECK128=b timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg > gpurun_out/r2o/bench_b_$rep.json 2> gpurun_out/r2o/bench_b_$rep.log
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg > gpurun_out/r2o/bench_c_$rep.json 2> gpurun_out/r2o/bench_c_$rep.log
done
python - <<'PY'
import json
for n in ("b_1","c_1","b_2","c_2"):
    try:
        d=json.load(open(f"gpurun_out/r2o/bench_{n}.json"))
        t=[s for s in d["roofline"]["top_shapes"] if "bneck_c128" in s["shape"]]
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], t)
    except Exception as e:
        print(n, "failed", e)
PY
