#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2e; mkdir -p $O
timeout 300 python tools/bneck_probe.py > $O/probe.log 2>&1; echo "probe rc=$?" | tee -a $O/summary.txt; cat $O/probe.log | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_letterbox.py tests/test_train.py -q --timeout=300 > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -5 $O/tests.log | cut -c1-300
