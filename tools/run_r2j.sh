#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2j; mkdir -p $O
timeout 300 python tools/bneck_bench.py 128 0,916,0,916 2>&1 | grep -v amdgpu.ids | tee $O/bneck128_prio.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/tests_all.log 2>&1; echo "all gpu tests rc=$?" | tee -a $O/summary.txt; tail -6 $O/tests_all.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4
