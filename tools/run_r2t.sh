#!/bin/bash
# A/B of the whole forward: 64-channel ring kernel (default) vs the weights-resident kernel (CFT_BNECK64=resident)
mkdir -p gpurun_out/r2t
for rep in 1 2; do
  CFT_BNECK64=resident timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg > gpurun_out/r2t/bench_res_$rep.json 2> gpurun_out/r2t/bench_res_$rep.log
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg > gpurun_out/r2t/bench_ring_$rep.json 2> gpurun_out/r2t/bench_ring_$rep.log
done
python - <<'PY'
import json
for n in ("res_1","ring_1","res_2","ring_2"):
    try:
        d=json.load(open(f"gpurun_out/r2t/bench_{n}.json"))
        t=[s for s in d["roofline"]["top_shapes"] if "bneck_c64" in s["shape"]]
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], t)
    except Exception as e:
        print(n, "failed", e)
PY
