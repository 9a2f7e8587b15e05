#!/bin/bash
mkdir -p gpurun_out/r2u
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "bottleneck" > gpurun_out/r2u/tests_bneck.log 2>&1
echo "tests rc=$?" | tee gpurun_out/r2u/summary.txt
tail -3 gpurun_out/r2u/tests_bneck.log
timeout 300 python tools/bneck_bench.py 128 0,9248,0,9248,0,9248 > gpurun_out/r2u/bneck128x.log 2>&1
cat gpurun_out/r2u/bneck128x.log
for rep in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg > gpurun_out/r2u/bench_c_$rep.json 2> gpurun_out/r2u/bench_c_$rep.log
  CFT_BNECK128=x timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg > gpurun_out/r2u/bench_x_$rep.json 2> gpurun_out/r2u/bench_x_$rep.log
done
python - <<'PY'
import json
for n in ("c_1","x_1","c_2","x_2"):
    try:
        d=json.load(open(f"gpurun_out/r2u/bench_{n}.json"))
        t=[s for s in d["roofline"]["top_shapes"] if "bneck_c128" in s["shape"]]
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], t)
    except Exception as e:
        print(n, "failed", e)
PY
