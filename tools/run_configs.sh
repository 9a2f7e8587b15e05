#!/bin/bash
# Bench lines of the other BASELINE configurations (single GPU share), round 2
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2cfg; mkdir -p $O
timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-f16-leg > $O/bench_bs8.json 2> $O/bs8.log; echo "bs8 rc=$?"
timeout 300 python bench.py --config cfg2 --batch 16 --dtype f32 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/cfg2.log; echo "cfg2 rc=$?"
timeout 300 python bench.py --config cfg4 --batch 64 --no-cpu-baseline --no-f16-leg > $O/bench_cfg4.json 2> $O/cfg4.log; echo "cfg4 rc=$?"
timeout 400 python bench.py --config cfg5 --batch 16 --size 1280 --no-cpu-baseline --no-f16-leg > $O/bench_cfg5.json 2> $O/cfg5.log; echo "cfg5 rc=$?"
python - <<'PY'
import json
for n in ("bs8","cfg2","cfg4","cfg5"):
    try:
        d=json.load(open(f"gpurun_out/r2cfg/bench_{n}.json"))
        print(n, d["dtype"], d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e:
        print(n, "failed", e)
PY
