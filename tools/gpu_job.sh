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
    timeout 2400 python -m pytest tests -q -m gpu -x --durations=40 "$@" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt
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
  tiles)       # tile choice re-check with the interleaved, warmed per-shape timing: automatic choice against the forced tile variants
    timeout 1200 python tools/gemm_bench.py --used --variants ${1:-0,27,60,51,23,33,30,6,63,2} --rounds 3 --out $JOB/gemm.json > $O/gemm.log 2>&1; echo "gemm rc=$?" | tee $O/summary.txt
    grep -E "^variant|^best" $O/gemm.log ;;
  r6a)         # round 6: the asm-K-loop kernel: bit-identity tests, per-shape A/B against the round-5 choice (variant 97), whole forward A/B
    timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "asm_gemm" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -5 $O/tests.log
    timeout 1200 python tools/gemm_bench.py --used --variants ${1:-97,0} --rounds 3 --out $JOB/gemm.json > $O/gemm.log 2>&1; echo "gemm rc=$?" | tee -a $O/summary.txt
    grep -E "^variant|^best|us " $O/gemm.log | tail -80
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    for v in 97 0 97 0; do timeout 300 python bench.py $X --conv-variant $v > $O/bench_v$v.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/bench_v$v.json'));print('variant $v', d['value'], d['ms_per_step'], d['roofline']['frac'], (d.get('single_in_flight') or {}).get('value'))" | tee -a $O/summary.txt; done ;;
  bneck)       # the fused Bottleneck kernels against the two launches + ablation probes (probe build)
    CFT_BENCH_LIB=multispectral-object-detection_amd/libcft_hip_probes.so timeout 600 python tools/bneck_bench.py 128 > $O/bneck128.log 2>&1; echo "bneck128 rc=$?"; cat $O/bneck128.log | tail -12 ;;
  bneck6)      # round 6: the 16 x 16-tile asm Bottleneck-128 kernel (variant 0) against the round-5 kernel (97): tests, then timing
    timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "bottleneck" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
    timeout 600 python tools/bneck_bench.py 128 > $O/bneck128.log 2>&1; echo "bneck128 rc=$?"; cat $O/bneck128.log | tail -8 ;;
  r6b)         # round 6: whole-forward A/B (variant 97 = round-5 choice, 0 = asm kernels) with the board power sampled during the sustained leg
    rocm-smi --showmaxpower --showpower 2>&1 | grep -i -E "power|watt" | head -4 | tee $O/summary.txt
    X="--no-cpu-baseline --no-f16-leg --no-parity --sustained-steps 400"
    for v in 97 0 97 0; do
      (timeout 300 python bench.py $X --conv-variant $v > $O/b_$v.json 2>> $O/bench.log &)
      sleep 50; for i in 1 2 3 4 5 6; do cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_input 2>/dev/null | tr '\n' ' '; echo; sleep 1; done | tee -a $O/power_$v.txt   # microwatts, every card of the node (ours is the one that moves)
      wait; sleep 25
      python -c "import json;d=json.load(open('$O/b_$v.json'));s=d.get('sustained') or {};print('variant $v', d['value'], d['ms_per_step'], (d.get('single_in_flight') or {}).get('value'), s.get('value'), (s.get('shader_clock_under_load') or {}).get('s_memtime_mhz'))" | tee -a $O/summary.txt
    done ;;
  pmc6)        # round 6: PMC counters of the asm-K-loop kernel (variant 96) and of the 16-wave kernel (97) on 3x3 512->512 @20 (72 K steps, 200 tiles)
    for v in 96 97; do
      bash tools/pmc.sh $O/v$v -- python tools/gemm_bench.py --variants $v --iters 10 --rounds 1 --only "bneck 3x3 512->512" --out $JOB/g$v.json > $O/pmc_v$v.log 2>&1
      python tools/pmc_summary.py $O/v$v conv_gemm > $O/pmc_3x3_512ch_20x20_variant$v.txt; rm -rf $O/v$v
      head -40 $O/pmc_3x3_512ch_20x20_variant$v.txt
    done ;;
  pmc6b)       # round 6: counters VERDICT r5 listed as evidence gaps - the shipped bottleneck128c_kernel, and the d = 1024 CFT linears at M = 8 192 (asm kernel, 128-row tiles)
    bash tools/pmc.sh $O/b128 -- python tools/bneck_bench.py 128 0 > $O/pmc_b128.log 2>&1
    python tools/pmc_summary.py $O/b128 bottleneck128c > $O/pmc_bottleneck128c.txt; rm -rf $O/b128; head -40 $O/pmc_bottleneck128c.txt
    for shape in "GPT fc2 4096->1024" "GPT qkv 1024->3072"; do
      tag=$(echo "$shape" | sed -e 's/[^A-Za-z0-9]\+/_/g')
      bash tools/pmc.sh $O/lin -- python tools/gemm_bench.py --variants 0 --iters 10 --rounds 1 --only "$shape" --out $JOB/lin.json > $O/pmc_$tag.log 2>&1
      python tools/pmc_summary.py $O/lin conv_gemm > $O/pmc_$tag.txt; rm -rf $O/lin; head -40 $O/pmc_$tag.txt
    done ;;
  chain6)      # round 6: the chained pairs on the asm K loop (variant 0) against the 16-wave chained kernel (97): tests, per-pair timing, whole forward
    timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x -k "chain" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -4 $O/tests.log
    for v in 97 0 97 0; do echo "variant $v"; CFT_CHAIN_BENCH_VARIANT=$v timeout 300 python tools/chain_bench.py --res 2>&1 | tail -3; CFT_CHAIN_BENCH_VARIANT=$v timeout 300 python tools/chain_bench.py 2>&1 | grep -E "256|bf16" | tail -4; done | tee $O/chain_bench.txt
    X="--no-cpu-baseline --no-f16-leg --no-parity --sustained-steps 0"
    for v in 97 0 97 0; do timeout 300 python bench.py $X --conv-variant $v > $O/bench_v$v.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/bench_v$v.json'));print('variant $v', d['value'], d['ms_per_step'], d['roofline']['frac'], (d.get('single_in_flight') or {}).get('value'))" | tee -a $O/summary.txt; done ;;
  micro)       # micro-benchmarks: HBM read / write / copy ceilings; Infinity-Cache producer -> consumer; DMA stream coupling; power coupling
    for m in ${@:-hbm_rw mall_probe dma_ring power_coupling}; do timeout 200 tools/micro/$m > $O/$m.txt 2>&1; echo "$m rc=$?"; tail -40 $O/$m.txt; done ;;
  r5a)         # round 5, first call: Infinity-Cache probe; new tests (depth-first prefix, survey weights, ADVICE fixes); depth-first A/B on the forward
    timeout 200 tools/micro/mall_probe > $O/mall_probe.txt 2>&1; echo "mall rc=$?" | tee $O/summary.txt; cat $O/mall_probe.txt
    timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -x -k "depth_first or survey or chain or cft_output_fusion or profile_flag or upsample_add_dual" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -5 $O/tests.log
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    run() { tag=$1; shift; timeout 400 python bench.py $X "$@" > $O/b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/b.json'));print('$tag', d['value'], d['ms_per_step'], (d.get('single_in_flight') or {}).get('value'), d['roofline']['whole_step']['frac'], d['config'].get('stream_group_probe_ms_per_step'))" | tee -a $O/summary.txt; }
    run "layer by layer (default)"; cp gpurun_out/bench_families.json $O/families_default.json
    run "depth-first 8 chunks, rows 0-4" --depth-first 8; cp gpurun_out/bench_families.json $O/families_df8.json
    run "depth-first 16 chunks, rows 0-4" --depth-first 16
    run "depth-first 4 chunks, rows 0-4" --depth-first 4
    run "depth-first 8 chunks, rows 0-2" --depth-first 8,3
    run "layer by layer (default)"
    run "depth-first 8 chunks, rows 0-4" --depth-first 8
    run "depth-first 8, one forward in flight" --depth-first 8 --in-flight 1
    run "layer by layer, one forward in flight" --in-flight 1 ;;
  r5b)         # round 5: chained 3x3 (+ shortcut) + 1x1 kernel, split-K linears + LayerNorm-reduce: op / model tests, A/B on the forward
    timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -x -k "chain or splitk or layernorm or attention or cft_output_fusion or graph_replay or train_forward" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -5 $O/tests.log
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    run() { tag=$1; shift; timeout 400 python bench.py $X "$@" > $O/b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/b.json'));r=d['roofline'];print('$tag', d['value'], d['ms_per_step'], (d.get('single_in_flight') or {}).get('value'), r['whole_step']['frac'], r['by_block']['cft_block_whole']['ms'], r['by_block']['cft_block_whole']['frac'], d['config'].get('stream_group_probe_ms_per_step'))" | tee -a $O/summary.txt; }
    run "default (chain + splitk)"; cp gpurun_out/bench_families.json $O/families_default.json
    run "no splitk" --no-splitk; cp gpurun_out/bench_families.json $O/families_nosplitk.json
    run "no conv chain" --no-conv-chain; cp gpurun_out/bench_families.json $O/families_nochain.json
    run "default (chain + splitk)"
    run "no splitk" --no-splitk
    run "no conv chain" --no-conv-chain
    run "bs8 default" --batch 8
    run "bs8 no splitk" --batch 8 --no-splitk ;;
  r5c)         # round 5: the one-kernel stem (Focus + Conv s2 + C3 cv1|cv2): op / model bit-identity tests, A/B on the forward
    timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -x -k "stem or chain or focus or depth_first or graph_replay or uint8 or six_channel" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -5 $O/tests.log
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    run() { tag=$1; shift; timeout 400 python bench.py $X "$@" > $O/b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/b.json'));r=d['roofline'];print('$tag', d['value'], d['ms_per_step'], (d.get('single_in_flight') or {}).get('value'), r['whole_step']['frac'], r['floors']['algorithmic_gbytes_per_step'], d['config'].get('stream_group_probe_ms_per_step'))" | tee -a $O/summary.txt; }
    run "stem" --stem; cp gpurun_out/bench_families.json $O/families_stem.json
    run "default (Focus, then chained Conv + C3)"; cp gpurun_out/bench_families.json $O/families_default.json
    run "stem" --stem
    run "default (Focus, then chained Conv + C3)"
    timeout 300 python tools/stem_bench.py > $O/stem_bench.txt 2>&1; cat $O/stem_bench.txt ;;
  r5d)         # round 5: small-batch regime after the size heuristics (chained 3x3 + shortcut only from 192 tiles; split-K to >= 128 workgroups)
    timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -x -k "chain_res or c3_chain or splitk" > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/summary.txt; tail -3 $O/tests.log
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    run() { tag=$1; shift; timeout 400 python bench.py $X "$@" > $O/b.json 2>> $O/bench.log; python -c "import json;d=json.load(open('$O/b.json'));r=d['roofline'];print('$tag', d['value'], d['ms_per_step'], (d.get('single_in_flight') or {}).get('value'), r['whole_step']['frac'], r['by_block']['cft_block_whole']['ms'], r['by_block']['cft_block_whole']['frac'], r['by_block']['backbone_head_convs']['ms'])" | tee -a $O/summary.txt; }
    run "bs8" --batch 8
    run "bs8 no splitk" --batch 8 --no-splitk
    run "bs8 no chain" --batch 8 --no-conv-chain
    run "bs8" --batch 8
    run "bs16" --batch 16
    run "bs16 no splitk" --batch 16 --no-splitk
    run "cfg5 16 pairs 1280" --config cfg5 --batch 16 --size 1280
    run "cfg5 16 pairs 1280 no splitk" --config cfg5 --batch 16 --size 1280 --no-splitk ;;
  knobs)       # runtime-environment knobs on the default step (each leg in its own process; a crash only loses its leg)
    X="--no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity"
    run() { tag=$1; shift; timeout 300 env "$@" python bench.py $X > $O/b.json 2>> $O/bench.log; rc=$?; python -c "import json;d=json.load(open('$O/b.json'));print('$tag', d['value'], d['ms_per_step'], (d.get('single_in_flight') or {}).get('value'), d['config'].get('stream_group_probe_ms_per_step'))" 2>/dev/null | tee -a $O/summary.txt || echo "$tag FAILED rc=$rc" | tee -a $O/summary.txt; rm -f $O/b.json; }
    run "default" A=1
    run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
    run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
    run "GPU_MAX_HW_QUEUES=2" GPU_MAX_HW_QUEUES=2
    run "GPU_MAX_HW_QUEUES=6" GPU_MAX_HW_QUEUES=6
    run "HSA_ENABLE_SDMA=0" HSA_ENABLE_SDMA=0
    run "default" A=1 ;;
  pmc5)        # round 5: per-pair timing + PMC counters of the chained 3x3 + shortcut + 1x1 kernel at the bench shape
    timeout 300 python tools/chain_bench.py --res > $O/chain_res_pairs.txt 2>&1; cat $O/chain_res_pairs.txt
    bash tools/pmc.sh $O/pmc -- python tools/chain_bench.py --res --iters 3 > $O/pmc.log 2>&1
    python tools/pmc_summary.py $O/pmc conv_gemm > $O/pmc_chainres_3x3_256ch_40x40.txt; rm -rf $O/pmc
    cat $O/pmc_chainres_3x3_256ch_40x40.txt | head -80 ;;
  bench)       # headline bench line (+ extra args)
    timeout 900 python bench.py "$@" > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee $O/summary.txt
    tail -4 $O/bench.log; head -c 400 $O/bench.json ;;
  evidence)    # PMC traffic passes, bench line, rocprofv3 kernel stats (single- and two-stream)
    bash tools/pmc_traffic.sh $O/traffic > $O/traffic.log 2>&1
    python tools/traffic_summary.py $O/traffic $O/traffic.json config=cfg3 batch=64 size=640 dtype=bf16 | tee $O/summary.txt
    rm -rf $O/traffic; cp $O/traffic.json profiles/r06_traffic.json
    timeout 900 python bench.py > $O/bench_bs64.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt
    cp gpurun_out/bench_families.json $O/gemm_families_cfg3.json
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof1 --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg --no-overlap --in-flight 1 --sustained-steps 0 --no-parity > $O/prof1.log 2>&1; echo "prof single-stream rc=$?" | tee -a $O/summary.txt
    cp $(ls $O/prof1/*/*kernel_stats.csv | head -1) $O/bench_bs64_kernel_stats.csv; rm -rf $O/prof1
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof2 --output-format csv -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-f16-leg --sustained-steps 0 --no-parity > $O/prof2.log 2>&1; echo "prof two-stream rc=$?" | tee -a $O/summary.txt
    cp $(ls $O/prof2/*/*kernel_stats.csv | head -1) $O/bench_bs64_kernel_stats_two_streams_40steps.csv; rm -rf $O/prof2
    # the parity-green 16-bit mode with the same evidence as the bf16 line: full line + rocprofv3 kernel stats
    timeout 900 python bench.py --dtype f16 --no-cpu-baseline > $O/bench_bs64_f16.json 2> $O/bench_f16.log; echo "bench f16 rc=$?" | tee -a $O/summary.txt
    cp gpurun_out/bench_families.json $O/gemm_families_cfg3_f16.json
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof3 --output-format csv -- python bench.py --dtype f16 --steps 20 --warmup 3 --no-cpu-baseline --no-overlap --in-flight 1 --sustained-steps 0 --no-parity > $O/prof3.log 2>&1; echo "prof f16 single-stream rc=$?" | tee -a $O/summary.txt
    cp $(ls $O/prof3/*/*kernel_stats.csv | head -1) $O/bench_bs64_f16_kernel_stats.csv; rm -rf $O/prof3
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
