#!/bin/bash
# full GPU suite + smoke + bench after: bottleneck128c default (ABI v3), fast exact-erf GELU
mkdir -p gpurun_out/r2q
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r2q/tests_all.log 2>&1
echo "tests rc=$?" | tee gpurun_out/r2q/summary.txt
tail -5 gpurun_out/r2q/tests_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2q/smoke.log 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/r2q/summary.txt
tail -3 gpurun_out/r2q/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg > gpurun_out/r2q/bench.json 2> gpurun_out/r2q/bench.log
echo "bench rc=$?" | tee -a gpurun_out/r2q/summary.txt
cat gpurun_out/r2q/bench.json | cut -c1-1500
cp gpurun_out/bench_families.json gpurun_out/r2q/families.json 2>/dev/null
