#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2l; mkdir -p $O
timeout 600 python tools/gemm_bench.py --variants 0,23,27,30,33,51,60,6,63,7 --iters 20 --out r2l_linears.json --only "GPT" 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(r['shape'], {k:v.get('us') for k,v in r['variants'].items()})
"
