#!/bin/bash
# GPU run B of round 2: staggered-group GEMM kernels (variants 80 / 81): parity, micro-benchmarks, A/B in the forward; full suite.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout=300 -k "staggered or tile_configuration" > $O/tests_gemm.log 2>&1; rc=$?; echo "gemm tests rc=$rc" | tee -a $O/summary.txt; tail -3 $O/tests_gemm.log
if [ $rc -eq 0 ]; then
  timeout 400 python tools/gemm_bench.py --variants 27,80,180,280,1680 --iters 10 --out r2b_gemm_wide.json --only "3x3s2 128->256|3x3s2 256->512|bneck 3x3 256->256|3x3s2 512->1024|bneck 3x3 512->512|1x1 1024->1024|SPP cv2|C3 1x1 512->512|GPT qkv 1024|GPT fc1 1024|GPT out 1024|GPT qkv 512|GPT fc1 512" > $O/gemm_wide.log 2>&1; echo "gemm wide rc=$?" | tee -a $O/summary.txt
  timeout 400 python tools/gemm_bench.py --variants 51,30,81,181,281,1681 --iters 10 --out r2b_gemm_n128.json --only "bneck 3x3 128->128|3x3s2 64->128|C3 1x1 128->128" > $O/gemm_n128.log 2>&1; echo "gemm n128 rc=$?" | tee -a $O/summary.txt
  cp gpurun_out/r2b_gemm_*.json $O/
  CFT_AUTO8=0 timeout 400 python bench.py --no-cpu-baseline --no-f16-leg > $O/bench_auto8_off.json 2> $O/bench_off.log; echo "bench off rc=$?" | tee -a $O/summary.txt
  cp gpurun_out/bench_families.json $O/families_off.json
  CFT_AUTO8=1 timeout 400 python bench.py --no-cpu-baseline --no-f16-leg > $O/bench_auto8_on.json 2> $O/bench_on.log; echo "bench on rc=$?" | tee -a $O/summary.txt
  cp gpurun_out/bench_families.json $O/families_on.json
fi
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > $O/tests_all.log 2>&1; echo "all tests rc=$?" | tee -a $O/summary.txt; tail -15 $O/tests_all.log
grep -h "variant\|best-of" $O/gemm_wide.log $O/gemm_n128.log
python - <<'P'
import json
for f in ("gpurun_out/r2b/bench_auto8_off.json","gpurun_out/r2b/bench_auto8_on.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["by_block"])
    except Exception as e: print(f, "ERR", e)
P
