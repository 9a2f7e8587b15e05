#!/bin/bash
# Evidence for profiles/: HBM traffic PMC passes first (the bench line cites them), headline bench line, rocprofv3 kernel
# stats (single-stream and two-stream).  Output directory: gpurun_out/$1 (default r2m).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/${1:-r2m}; mkdir -p $O
bash tools/pmc_traffic.sh $O/traffic > $O/traffic.log 2>&1
python tools/traffic_summary.py $O/traffic $O/traffic.json config=cfg3 batch=64 size=640 dtype=bf16 | tee -a $O/summary.txt
rm -rf $O/traffic
cp $O/traffic.json profiles/r02_traffic.json
timeout 600 python bench.py > $O/bench_bs64.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/summary.txt
cp gpurun_out/bench_families.json $O/gemm_families_cfg3.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof1 --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f16-leg --no-overlap > $O/prof1.log 2>&1; echo "prof single-stream rc=$?" | tee -a $O/summary.txt
cp $(ls $O/prof1/*/*kernel_stats.csv | head -1) $O/bench_bs64_kernel_stats.csv; rm -rf $O/prof1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof2 --output-format csv -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-f16-leg > $O/prof2.log 2>&1; echo "prof two-stream (40 steps) rc=$?" | tee -a $O/summary.txt
cp $(ls $O/prof2/*/*kernel_stats.csv | head -1) $O/bench_bs64_kernel_stats_two_streams_40steps.csv; rm -rf $O/prof2
grep -h "copyBuffer" $O/bench_bs64_kernel_stats.csv $O/bench_bs64_kernel_stats_two_streams_40steps.csv | cut -d, -f1-3
head -c 1500 $O/bench_bs64.json
