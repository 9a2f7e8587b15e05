#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r2g; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x --timeout=300 -k "conv2d or tile_configuration" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/tests.log
timeout 400 python tools/gemm_bench.py --variants 23,3223,51,3251,27,3227 --iters 20 --out r2g_bias.json --only "bneck 1x1 128->128|C3 1x1 256->256|bneck 1x1 256->256|bneck 3x3 128->128|bneck 3x3 256->256|C3 1x1 128->128|GPT out 512|GPT fc1 256" 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(r['shape'], {k:(v.get('us'),v.get('tflops')) for k,v in r['variants'].items()})
"
timeout 400 python bench.py --no-cpu-baseline --no-f16-leg > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/r2g/bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["top_shapes"][:4])
P
