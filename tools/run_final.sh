#!/bin/bash
# Round-end validation on the GPU box: full -m gpu suite, smoke, parity diagnostics, then the evidence run.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-final}; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/tests_all.log 2>&1; echo "tests rc=$?" | tee $O/summary_tests.txt
tail -3 $O/tests_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary_tests.txt
tail -3 $O/smoke.log
timeout 700 python tests/gpu_diag.py > $O/diag.log 2>&1; echo "diag rc=$?" | tee -a $O/summary_tests.txt
cp gpurun_out/diag.json $O/diag.json 2>/dev/null
bash tools/run_evidence.sh ${1:-final}
