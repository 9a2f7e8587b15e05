#!/bin/bash
# after the chunk-major K walk: full GPU suite, smoke, headline bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2y; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/tests_all.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
tail -3 $O/tests_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_bs64.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt
cp gpurun_out/bench_families.json $O/gemm_families_cfg3.json
head -c 600 $O/bench_bs64.json
