#!/bin/bash
# GPU run A of round 2: full GPU test suite, smoke, parity diagnostics, bench, NMS timing, copyBuffer probes.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -x --timeout=600 > gpurun_out/r2a/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2a/summary.txt
tail -5 gpurun_out/r2a/tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r2a/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r2a/summary.txt
timeout 600 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.log; echo "bench rc=$?" | tee -a gpurun_out/r2a/summary.txt
cp gpurun_out/bench_families.json gpurun_out/r2a/ 2>/dev/null
timeout 300 python tools/nms_bench.py > gpurun_out/r2a/nms.log 2>&1; echo "nms rc=$?" | tee -a gpurun_out/r2a/summary.txt
timeout 600 python tests/gpu_diag.py > gpurun_out/r2a/diag.log 2>&1; echo "diag rc=$?" | tee -a gpurun_out/r2a/summary.txt
cp gpurun_out/diag.json gpurun_out/r2a/ 2>/dev/null
for m in plain events streams graph; do
  timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/r2a/cb_$m --output-format csv -- python tools/copybuffer_probe.py $m > gpurun_out/r2a/cb_$m.log 2>&1
  f=$(ls gpurun_out/r2a/cb_$m/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $m" >> gpurun_out/r2a/summary.txt; [ -n "$f" ] && cut -d, -f1,2 "$f" | head -8 >> gpurun_out/r2a/summary.txt
  rm -rf gpurun_out/r2a/cb_$m
done
cat gpurun_out/r2a/summary.txt; cat gpurun_out/r2a/bench.json | head -c 3000
