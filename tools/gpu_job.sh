#!/bin/bash
# One entry point for every GPU-box job of a round (replaces the per-experiment run_r2*.sh scripts of round 2):
#   gpurun --timeout N -- bash tools/gpu_job.sh <job> [args]
# Every job writes under gpurun_out/<job>/ ; summaries worth keeping are copied to profiles/ by hand afterwards.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
JOB=${1:-help}; shift
O=gpurun_out/$JOB; mkdir -p $O
case $JOB in
  parity)      # the round-3 parity tests + the bench line with parity_at_bench_shape / sustained
    timeout 1500 python -m pytest tests -q -m gpu -x -k "references_own_bf16 or reference_constructor or benchmarked or cfg5_bf16 or channel_slices_of_a_six or fresh_model or train_forward or profile_flag" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
    tail -5 $O/tests.log
    timeout 900 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt
    tail -3 $O/bench.log; head -c 600 $O/bench.json ;;
  tests)       # full -m gpu suite + smoke
    timeout 2400 python -m pytest tests -q -m gpu -x "$@" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
    tail -5 $O/tests.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -3 $O/smoke.log ;;
  gemm)        # tools/gemm_bench.py A/B: gpu_job.sh gemm <variants> [only-filter]
    timeout 1200 python tools/gemm_bench.py --variants "$1" ${2:+--only "$2"} --out $JOB/gemm.json > $O/gemm.log 2>&1; echo "gemm rc=$?" | tee $O/summary.txt
    grep -E "^variant|^best" $O/gemm.log ;;
  ab900)       # A/B of the automatic choice (variant 0) against variant 900: bit-identity tests, per-shape A/B, whole-forward A/B
    timeout 1200 python -m pytest tests -q -m gpu -x -k "alternative_gemm or test_conv2d or bottleneck or channel_slices or fresh_model or train_forward or profile_flag or wide_layers" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
    tail -4 $O/tests.log
    timeout 900 python tools/gemm_bench.py --variants 900,0 --out $JOB/gemm.json > $O/gemm.log 2>&1; echo "gemm rc=$?" | tee -a $O/summary.txt
    grep -E "^variant|^best" $O/gemm.log
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    for v in 900 0 900 0; do timeout 300 python bench.py $X --conv-variant $v > $O/bench_v$v.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/bench_v$v.json'));print('variant $v', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/summary.txt; done ;;
  batches)     # pairs/s against the batch per GPU (tile-count quantisation on 256 CUs: 64 pairs = 400 tiles of 256 rows at 40x40)
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    for b in 64 80 48 96 64 80; do timeout 300 python bench.py $X --batch $b > $O/bench_b$b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/bench_b$b.json'));print('batch $b', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/summary.txt; done ;;
  inflight)    # forwards in flight x intra-forward stream overlap
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    for cfgline in "--in-flight 2" "--in-flight 3" "--in-flight 3 --no-overlap" "--in-flight 4" "--in-flight 2" "--in-flight 3"; do
      timeout 400 python bench.py $X $cfgline > $O/b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/b.json'));print('$cfgline', d['value'], d['ms_per_step'], d.get('single_in_flight'))" | tee -a $O/summary.txt; done ;;
  nms)         # NMS tests + timing on the bench workload
    timeout 900 python -m pytest tests/test_nms.py tests/test_letterbox.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -3 $O/tests.log
    timeout 600 python tools/nms_bench.py > $O/nms_bench.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt; cat $O/nms_bench.log | tail -4; cp gpurun_out/nms_bench.json $O/
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python tools/nms_bench.py > $O/prof.log 2>&1
    python - <<PY
import csv, glob
f = glob.glob("$O/prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "nms" in r["Name"]:
        print(r["Name"][:40], r["Calls"], r["AverageNs"], r["TotalDurationNs"])
PY
    rm -rf $O/prof ;;
  diag)        # parity diagnostics (measured errors per configuration / precision, incl. vs the reference's own bf16 forward)
    timeout 900 python tests/gpu_diag.py > $O/diag.log 2>&1; echo "diag rc=$?" | tee $O/summary.txt; cp gpurun_out/diag.json $O/diag.json; grep lowp_case $O/diag.log | cut -c1-400 ;;
  prof1)       # rocprofv3 kernel stats of the single-stream, one-forward-in-flight run (per-kernel durations are meaningful)
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof1 --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg --no-overlap --in-flight 1 --sustained-steps 0 --no-parity > $O/prof1.log 2>&1; echo "prof single-stream rc=$?" | tee $O/summary.txt
    cp $(ls $O/prof1/*/*kernel_stats.csv | head -1) $O/bench_bs64_kernel_stats.csv; rm -rf $O/prof1; tail -2 $O/prof1.log | cut -c1-300 ;;
  pmc)         # PMC counters of the 16-wave GEMM kernel on 3x3 256->256 @40 (+res), generic (900) and uniform-K-walk (0) address path
    for v in 900 0; do
      bash tools/pmc.sh $O/v$v -- python tools/gemm_bench.py --variants $v --iters 10 --only "bneck 3x3 256->256" --out $JOB/g$v.json > $O/pmc_v$v.log 2>&1
      python tools/pmc_summary.py $O/v$v conv_gemm > $O/pmc_3x3_256ch_40x40_variant$v.txt; rm -rf $O/v$v
      head -30 $O/pmc_3x3_256ch_40x40_variant$v.txt
    done ;;
  kg4)         # (needs tools/experimental/r03_half_kstep.patch applied + rebuilt) half-K-step tiles (variants 61 / 62): tests, per-shape A/B against the automatic choice; small-batch in-flight sweep
    timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "alternative_gemm or every_tile" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -3 $O/tests.log
    timeout 900 python tools/gemm_bench.py --variants 0,61,62 --out $JOB/gemm.json > $O/gemm.log 2>&1; echo "gemm rc=$?" | tee -a $O/summary.txt
    grep -E "^variant|^best" $O/gemm.log
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    for cfgline in "--batch 8 --in-flight 2" "--batch 8 --in-flight 4" "--batch 8 --in-flight 8" "--batch 16 --in-flight 4"; do
      timeout 400 python bench.py $X $cfgline > $O/b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/b.json'));print('$cfgline', d['value'], d['ms_per_step'], d.get('single_in_flight',{}).get('value'))" | tee -a $O/summary.txt; done ;;
  persist)     # (needs tools/experimental/r03_persistent_tiles.patch applied + rebuilt) persistent tile loop of the GEMM family (variant 0) against one workgroup per tile (variant 901): tests, per-shape A/B, whole forward A/B
    timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv or gemm or bottleneck or focus or linear or tile" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -3 $O/tests.log
    timeout 900 python tools/gemm_bench.py --variants 901,0 --out $JOB/gemm.json > $O/gemm.log 2>&1; echo "gemm rc=$?" | tee -a $O/summary.txt
    grep -E "^variant|^best" $O/gemm.log
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    for v in 901 0 901 0; do timeout 300 python bench.py $X --conv-variant $v > $O/bench_v$v.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/bench_v$v.json'));print('variant $v', d['value'], d['ms_per_step'], d['single_in_flight']['value'], d['roofline']['frac'])" | tee -a $O/summary.txt; done ;;
  tiles)       # tile choice re-check with the interleaved, warmed per-shape timing: automatic choice against the forced tile variants
    timeout 1200 python tools/gemm_bench.py --used --variants ${1:-0,27,60,51,23,33,30,6,63,2} --rounds 3 --out $JOB/gemm.json > $O/gemm.log 2>&1; echo "gemm rc=$?" | tee $O/summary.txt
    grep -E "^variant|^best" $O/gemm.log ;;
  r4a)         # round 4, first call: L2->LDS stream microbenchmark, ring GEMM kernel (variant 90) bit-identity + per-shape A/B + probes, runtime knobs
    timeout 120 tools/micro/dma_ring > $O/dma_ring.txt 2>&1; echo "dma rc=$?" | tee $O/summary.txt; tail -64 $O/dma_ring.txt
    timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "alternative_gemm or wide_layers" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -4 $O/tests.log
    WIDE="3x3s2 128->256|C3 1x1 256->256|3x3s2 256->512|C3 1x1 512->512|bneck 1x1 256->256|bneck 3x3 256->256|3x3s2 512->1024|bneck 3x3 512->512|1x1 1024->1024|SPP cv2|GPT|quant"
    timeout 600 python tools/gemm_bench.py --variants 0,91 --only "$WIDE" --rounds 3 --out $JOB/gemm_ring_ab.json > $O/gemm.log 2>&1; echo "gemm rc=$?" | tee -a $O/summary.txt
    python - <<PY
import json
for r in json.load(open("gpurun_out/$JOB/gemm_ring_ab.json")):
    v = r["variants"]
    print(f'{r["shape"]:42s} auto {v["0"]["us"]:7.1f} us {v["0"]["tflops"]:7.1f} TF | ring {v["91"]["us"]:7.1f} us {v["91"]["tflops"]:7.1f} TF  diff {v["91"]["maxdiff_vs_first"]}')
PY
    grep -E "^variant|^best" $O/gemm.log
    if [ -f multispectral-object-detection_amd/libcft_hip_probes.so ]; then
      timeout 600 python tools/gemm_bench.py --lib multispectral-object-detection_amd/libcft_hip_probes.so --variants 27,127,227,1627,91,190,290,1690 --only "bneck 3x3 256->256|GPT fc1 1024|C3 1x1 512->512|quant 3x3" --rounds 3 --out $JOB/gemm_ring_probes.json > $O/probes.log 2>&1; echo "probes rc=$?" | tee -a $O/summary.txt
      python - <<PY
import json
for r in json.load(open("gpurun_out/$JOB/gemm_ring_probes.json")):
    print(r["shape"], {k: v.get("us") for k, v in r["variants"].items()})
PY
    fi
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    run() { tag=$1; shift; timeout 300 env "$@" python bench.py $X $ARGS > $O/b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/b.json'));print('$tag', d['value'], d['ms_per_step'], (d.get('single_in_flight') or {}).get('value'), d['config'].get('stream_group_probe_ms_per_step'))" | tee -a $O/summary.txt; }
    ARGS="--in-flight 2"; run "default q, 2 in flight" A=1
    ARGS="--in-flight 2"; run "GPU_MAX_HW_QUEUES=8, 2 in flight" GPU_MAX_HW_QUEUES=8
    ARGS="--in-flight 3"; run "GPU_MAX_HW_QUEUES=8, 3 in flight" GPU_MAX_HW_QUEUES=8
    ARGS="--in-flight 3"; run "default q, 3 in flight" A=1
    ARGS="--in-flight 2 --steps 200"; run "kernarg=0" HIP_FORCE_DEV_KERNARG=0
    ARGS="--in-flight 2 --steps 200"; run "kernarg=1" HIP_FORCE_DEV_KERNARG=1
    ARGS="--in-flight 2 --steps 200"; run "kernarg=0" HIP_FORCE_DEV_KERNARG=0
    ARGS="--in-flight 2 --steps 200"; run "kernarg=1" HIP_FORCE_DEV_KERNARG=1
    ARGS="--in-flight 2 --conv-variant 90"; run "ring kernel (variant 90) on the wide layers, 2 in flight" A=1
    ARGS="--in-flight 2"; run "default again" A=1
    timeout 600 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x > $O/dist_tests.log 2>&1; echo "rccl world-size-1 tests rc=$?" | tee -a $O/summary.txt; tail -4 $O/dist_tests.log
    timeout 300 python bench.py $X --in-flight 2 --force-gather > $O/bench_force_gather.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/bench_force_gather.json'));print('force-gather', d['value'], d['ms_per_step'], d['multi_gpu_selfcheck'])" | tee -a $O/summary.txt ;;
  r4b)         # round 4: the 8-wave full-line GEMM kernel (variants 90 / 91): bit-identity, per-shape A/B, probes, whole-forward A/B
    timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "alternative_gemm or wide_layers" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -4 $O/tests.log
    WIDE="3x3s2 128->256|C3 1x1 256->256|3x3s2 256->512|C3 1x1 512->512|bneck 1x1 256->256|bneck 3x3 256->256|3x3s2 512->1024|bneck 3x3 512->512|1x1 1024->1024|SPP cv2|GPT|quant"
    timeout 600 python tools/gemm_bench.py --variants 0,91 --only "$WIDE" --rounds 3 --out $JOB/gemm_ab.json > $O/gemm.log 2>&1; echo "gemm rc=$?" | tee -a $O/summary.txt
    python - <<PY
import json
for r in json.load(open("gpurun_out/$JOB/gemm_ab.json")):
    v = r["variants"]
    print(f'{r["shape"]:42s} auto {v["0"]["us"]:7.1f} us {v["0"]["tflops"]:7.1f} TF | 8-wave {v["91"]["us"]:7.1f} us {v["91"]["tflops"]:7.1f} TF  diff {v["91"]["maxdiff_vs_first"]}')
PY
    if [ -f multispectral-object-detection_amd/libcft_hip_probes.so ]; then
      timeout 600 python tools/gemm_bench.py --lib multispectral-object-detection_amd/libcft_hip_probes.so --variants 27,127,227,1627,91,190,290,1690 --only "bneck 3x3 256->256|GPT fc1 1024|C3 1x1 512->512|quant 3x3" --rounds 3 --out $JOB/gemm_probes.json > $O/probes.log 2>&1; echo "probes rc=$?" | tee -a $O/summary.txt
      python - <<PY
import json
for r in json.load(open("gpurun_out/$JOB/gemm_probes.json")):
    print(r["shape"], {k: v.get("us") for k, v in r["variants"].items()})
PY
    fi
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    for v in 0 90 0 90; do timeout 300 python bench.py $X --conv-variant $v > $O/bench_v$v.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/bench_v$v.json'));print('variant $v', d['value'], d['ms_per_step'], d['single_in_flight']['value'], d['roofline']['frac'])" | tee -a $O/summary.txt; done ;;
  r4c)         # round 4: new tests (dual de-tokeniser, fusion plan, NMS module / autoShape, RCCL world size 1, probe-build 8-wave kernel), fusion A/B, stream priorities, small-batch in-flight sweep, cfg5
    timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_letterbox.py tests/test_gpu_distributed.py -q -m gpu -x -k "upsample_add_dual or probe_build or cft_output_fusion or nms_module or rccl or sharded_detect or tokenize or two_forwards" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -5 $O/tests.log
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    run() { tag=$1; shift; timeout 300 python bench.py $X "$@" > $O/b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/b.json'));print('$tag', d['value'], d['ms_per_step'], (d.get('single_in_flight') or {}).get('value'), d['roofline']['whole_step']['frac'], d['config'].get('stream_group_probe_ms_per_step'))" | tee -a $O/summary.txt; }
    run "fused CFT outputs (default)"
    run "three launches (--no-cft-fusion)" --no-cft-fusion
    run "fused CFT outputs (default)"
    run "three launches (--no-cft-fusion)" --no-cft-fusion
    run "stream priorities -1,0" --stream-priorities=-1,0
    run "stream priorities -1,-1" --stream-priorities=-1,-1
    run "bs8 in-flight 2" --batch 8 --in-flight 2
    run "bs8 in-flight 4" --batch 8 --in-flight 4
    run "bs8 in-flight 6" --batch 8 --in-flight 6
    run "bs8 in-flight 8" --batch 8 --in-flight 8
    run "bs16 in-flight 4" --batch 16 --in-flight 4
    run "cfg5 16 pairs 1280 in-flight 2" --config cfg5 --batch 16 --size 1280 --in-flight 2
    run "cfg5 16 pairs 1280 in-flight 3" --config cfg5 --batch 16 --size 1280 --in-flight 3 ;;
  r4d)         # round 4: full -m gpu suite + smoke + the default bench line (format check of the new fields)
    timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -5 $O/tests.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -3 $O/smoke.log
    timeout 900 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt; tail -2 $O/bench.log; head -c 3000 $O/bench.json ;;
  r4e)         # round 4: Conv + C3.cv1|cv2 chained kernel: bit-identity tests, per-pair timing, whole-forward A/B (interleaved)
    timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x -k "chain" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -5 $O/tests.log
    timeout 300 python tools/chain_bench.py > $O/chain_pairs.txt 2>&1; cat $O/chain_pairs.txt
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    run() { tag=$1; shift; timeout 300 python bench.py $X "$@" > $O/b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/b.json'));print('$tag', d['value'], d['ms_per_step'], (d.get('single_in_flight') or {}).get('value'), d['roofline']['whole_step']['frac'], d['config'].get('stream_group_probe_ms_per_step'))" | tee -a $O/summary.txt; }
    run "chained Conv+C3 (default)"
    run "two launches (--no-conv-chain)" --no-conv-chain
    run "chained Conv+C3 (default)"
    run "two launches (--no-conv-chain)" --no-conv-chain
    run "bs8 in-flight 4 chained" --batch 8 --in-flight 4
    run "bs8 in-flight 4 two launches" --batch 8 --in-flight 4 --no-conv-chain ;;
  pmc4)        # round 4: PMC counters of the shipped 16-wave kernel (27) and the probe build's 8-wave kernel (91) on 3x3 256->256 @40 (+res)
    for v in 27 91; do
      bash tools/pmc.sh $O/v$v -- python tools/gemm_bench.py --lib multispectral-object-detection_amd/libcft_hip_probes.so --variants $v --iters 10 --rounds 1 --only "bneck 3x3 256->256" --out $JOB/g$v.json > $O/pmc_v$v.log 2>&1
      python tools/pmc_summary.py $O/v$v conv_gemm > $O/pmc_3x3_256ch_40x40_variant$v.txt; rm -rf $O/v$v
      head -32 $O/pmc_3x3_256ch_40x40_variant$v.txt
    done ;;
  micro4)      # round 4 micro-benchmarks: HBM read / write / copy ceilings; DMA stream coupled with fragment reads / MFMAs; power coupling
    timeout 120 tools/micro/hbm_rw > $O/hbm_rw.txt 2>&1; cat $O/hbm_rw.txt
    timeout 200 tools/micro/dma_ring > $O/dma_ring.txt 2>&1; grep -A20 "coupling of the DMA" $O/dma_ring.txt
    timeout 120 tools/micro/power_coupling > $O/power_coupling.txt 2>&1; cat $O/power_coupling.txt ;;
  bench)       # headline bench line (+ extra args)
    timeout 900 python bench.py "$@" > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee $O/summary.txt
    tail -4 $O/bench.log; head -c 400 $O/bench.json ;;
  evidence)    # PMC traffic passes, bench line, rocprofv3 kernel stats (single- and two-stream)
    timeout 120 tools/micro/power_coupling > $O/power_coupling.txt 2>&1; cat $O/power_coupling.txt
    bash tools/pmc_traffic.sh $O/traffic > $O/traffic.log 2>&1
    python tools/traffic_summary.py $O/traffic $O/traffic.json config=cfg3 batch=64 size=640 dtype=bf16 | tee $O/summary.txt
    rm -rf $O/traffic; cp $O/traffic.json profiles/r04_traffic.json
    timeout 900 python bench.py > $O/bench_bs64.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt
    cp gpurun_out/bench_families.json $O/gemm_families_cfg3.json
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof1 --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg --no-overlap --in-flight 1 --sustained-steps 0 --no-parity > $O/prof1.log 2>&1; echo "prof single-stream rc=$?" | tee -a $O/summary.txt
    cp $(ls $O/prof1/*/*kernel_stats.csv | head -1) $O/bench_bs64_kernel_stats.csv; rm -rf $O/prof1
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof2 --output-format csv -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity > $O/prof2.log 2>&1; echo "prof two-stream rc=$?" | tee -a $O/summary.txt
    cp $(ls $O/prof2/*/*kernel_stats.csv | head -1) $O/bench_bs64_kernel_stats_two_streams_40steps.csv; rm -rf $O/prof2
    head -c 1500 $O/bench_bs64.json ;;
  configs)     # bench lines of the other BASELINE configurations (single-GPU share)
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0"
    timeout 300 python bench.py --batch 8 $X > $O/bench_bs8.json 2> $O/bs8.log; echo "bs8 rc=$?"
    timeout 300 python bench.py --config cfg2 --batch 16 --dtype f32 --no-cpu-baseline --sustained-steps 0 > $O/bench_cfg2.json 2> $O/cfg2.log; echo "cfg2 rc=$?"
    timeout 300 python bench.py --config cfg4 --batch 64 $X > $O/bench_cfg4.json 2> $O/cfg4.log; echo "cfg4 rc=$?"
    timeout 400 python bench.py --config cfg5 --batch 16 --size 1280 $X > $O/bench_cfg5.json 2> $O/cfg5.log; echo "cfg5 rc=$?"
    python - <<'PY'
import json
for n in ("bs8", "cfg2", "cfg4", "cfg5"):
    try:
        d = json.load(open(f"gpurun_out/configs/bench_{n}.json"))
        print(n, d["dtype"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("parity_at_bench_shape"))
    except Exception as e:
        print(n, "failed", e)
PY
    ;;
  *) echo "jobs: parity | tests [pytest args] | gemm <variants> [filter] | bench [args] | evidence | configs" ;;
esac
