#!/usr/bin/env python
"""Headline benchmark: image-pairs/sec, forward only, yolov5l + CFTx3 (FLIR cfg), 640x640, batch 64
per GPU, bf16 compute (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one forward of the whole two-stream network over one batch of synthetic pairs already
resident in HBM (HIP-graph replay of every kernel, pre-NMS detections materialised), followed, for
N > 1, by the RCCL all-gather of the detections.  Weak scaling: 64 pairs per GPU.  Rank 0 prints
ONE JSON line; it also carries
  roofline      the dominant kernel family (implicit-GEMM conv/linear, bf16 MFMA): algorithmic
                FLOPs / summed launch time, each launch bracketed by HIP events on its stream
  cpu_baseline  the CPU oracle (oracle/cft_oracle.py, a port of the reference forward to plain
                torch fp32) timed on this box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd import distributed as D  # noqa: E402
from msod_amd import ops  # noqa: E402
from msod_amd.models.configs import named_config  # noqa: E402
from msod_amd.models.yolo_test import Model  # noqa: E402
from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict  # noqa: E402

WORKLOADS = {"cfg1": "yolov5s add-fusion, no CFT", "cfg2": "yolov5s + 1 CFT block",
             "cfg3": "yolov5l_fusion_transformerx3_FLIR_aligned", "cfg4": "yolov5l_fusion_transformerx3_llvip",
             "cfg5": "yolov5x x3 CFT (derived)"}
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def event_bracket_overhead(n=200):
    """Elapsed time reported by an EMPTY event bracket on the current stream: what recording two events
    back to back costs; subtracted from every bracketed launch (the rocprofv3 trace of the instrumented
    forward shows exactly this gap around each kernel)."""
    pairs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in pairs)
    return ts[len(ts) // 2] * 1e-3


def gemm_family_time(model, rgb, ir):
    """One eager forward with every GEMM launch bracketed by HIP events -> per-family totals."""
    ovh = event_bracket_overhead()
    log = []
    ops.set_launch_log(log)
    overlap = model.overlap_streams
    model.overlap_streams = False       # per-launch brackets are only meaningful on a single stream
    try:
        with torch.no_grad():
            model.forward_once(rgb, ir)
        torch.cuda.synchronize()
    finally:
        ops.set_launch_log(None)
        model.overlap_streams = overlap
    fam = {}
    abytes = 0.0
    for name, flops, e0, e1, ab in log:
        abytes += ab
        f = fam.setdefault(name, [0, 0.0, 0.0])
        f[0] += 1
        f[1] += flops
        f[2] += max(e0.elapsed_time(e1) * 1e-3 - ovh, 1e-7)
    n = sum(v[0] for v in fam.values())
    flops = sum(v[1] for v in fam.values())
    secs = sum(v[2] for v in fam.values())
    return n, flops, secs, fam, ovh, abytes


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def traffic_from_profile(args, n_launch, abytes):
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh ->
    profiles/*_traffic.json: FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, summed
    over the conv_gemm family of one forward).  Only reported for the configuration it was collected on."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    if (t.get("config"), t.get("batch"), t.get("size"), t.get("dtype")) != (args.config, args.batch, args.size, args.dtype):
        return None
    return {"bytes_per_launch": t["gemm_bytes_per_forward"] / n_launch, "algorithmic_bytes_per_launch": abytes / n_launch,
            "source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"}


def cpu_baseline(cfg, sd, height, width, budget_s=20.0):
    """Reported baseline only: the oracle (port of the reference forward) on the host cores, torch's
    default intra-op thread count (NOT os.cpu_count(): the box may expose more CPUs than the
    container may use).  Bounded: one warm-up pair, then batches of 2 until ~budget_s is spent."""
    from oracle.cft_oracle import OracleModel
    rgb, ir = seeded_inputs(2, height, width, seed=0)
    om = OracleModel(cfg)
    t0 = time.perf_counter()
    om(sd, rgb[:1], ir[:1])
    warm = time.perf_counter() - t0
    times = []
    spent = 0.0
    while not times or (spent + statistics.median(times) < budget_s and len(times) < 9):
        t0 = time.perf_counter()
        om(sd, rgb, ir)
        times.append(time.perf_counter() - t0)
        spent += times[-1]
    med = statistics.median(times)
    return {"value": round(2 / med, 3), "unit": "image-pairs/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/cft_oracle.py fp32, same network and {height}x{width} inputs, batch 2, median of "
                      f"{len(times)} forwards ({med:.2f} s each; first call {warm:.2f} s for 1 pair); "
                      f"os.cpu_count()={os.cpu_count()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="image pairs per GPU")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--no-concat-plan", action="store_true", help="A/B: let Concat copy all its sources")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run both backbones on one HIP stream")
    args = ap.parse_args()

    rank, world, local = D.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    log(f"rank {rank}/{world} building {args.config}")
    cfg = named_config(args.config)
    model = Model(cfg)
    sd = seeded_state_dict(model.state_dict(), seed=0)      # random-init weights, BN/pos_emb non-trivial
    model.load_state_dict(sd)
    model = model.to(dev).fuse().set_compute_dtype(dtype)   # deployed form: BN folded (attempt_load does .fuse())
    model.overlap_streams = not args.no_overlap
    model.plan_concats = not args.no_concat_plan
    rgb, ir = seeded_inputs(args.batch, args.size, args.size, seed=rank)
    rgb, ir = rgb.to(dev), ir.to(dev)

    log("weights loaded, packing + capturing")
    with torch.no_grad():
        if args.no_graph:
            step_fn = lambda: model.forward_once(rgb, ir)   # noqa: E731
            pred, _ = step_fn()
        else:
            cap = model.capture(args.batch, args.size, args.size)
            cap.rgb.copy_(rgb)
            cap.ir.copy_(ir)
            step_fn = cap.replay_static
            pred, _ = step_fn()
        # N > 1: the all-gather of step i runs on RCCL's stream while the forward of step i+1 runs on the compute
        # stream (51.6 MB per rank per step would otherwise add ~10 % serial time); see distributed.OverlappedGather.
        gather = D.OverlappedGather(pred, world) if world > 1 else None

        def step():
            p, _ = step_fn()
            if gather is not None:
                gather.submit(p)

        def drain():
            if gather is not None:
                gather.drain()

        for _ in range(args.warmup):
            step()
        drain()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()                                             # every gather of the K timed steps has completed
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(pred).all(), "non-finite detections"
    log(f"timed region: {elapsed:.3f} s for {args.steps} steps")

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        n_launch, flops, secs, fam, ovh, abytes = gemm_family_time(model, rgb, ir)
        log(f"gemm family: {n_launch} launches, {secs * 1e3:.2f} ms, {flops / secs / 1e12:.1f} TFLOP/s")
        top = sorted(fam.items(), key=lambda kv: -kv[1][2])[:6]
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            with open(os.path.join(ROOT, "gpurun_out", "bench_families.json"), "w") as fh:
                json.dump({k: {"launches": v[0], "gflop": v[1] / 1e9, "ms": v[2] * 1e3, "tflops": v[1] / v[2] / 1e12}
                           for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2])}, fh, indent=1)
        peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
        achieved = flops / secs / 1e12
        split = {}
        for kind in ("conv", "linear"):     # convolutions vs the nn.Linear GEMMs of the CFT (GPT) blocks
            fl = sum(v[1] for k, v in fam.items() if k.startswith(kind))
            tt = sum(v[2] for k, v in fam.items() if k.startswith(kind))
            if tt > 0:
                split[kind] = {"tflops": round(fl / tt / 1e12, 1), "frac": round(fl / tt / 1e12 / peak, 4), "ms": round(tt * 1e3, 3)}
        line = {
            "metric": "image-pairs/sec fwd, yolov5l+CFTx3 640x640 bs64, 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "image-pairs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.config} ({WORKLOADS.get(args.config, args.config)}) two-stream forward, "
                                   f"{args.size}x{args.size}, {args.batch} pairs/GPU, BN folded, pre-NMS detections",
                       "pairs_per_gpu": args.batch, "image_size": args.size, "parallelism": f"batch-shard x{world}",
                       "hip_graph": not args.no_graph, "two_hip_streams": not args.no_overlap},
            "roofline": {"bound": "mfma", "kernel": "conv_gemm_kernel (implicit-GEMM conv/linear family; incl. the dedicated Focus and 64-channel Bottleneck kernels, 8 of the launches)",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic_from_profile(args, n_launch, abytes), "launches_per_step": n_launch,
                         "event_bracket_overhead_us": round(ovh * 1e6, 2),
                         "avg_launch_us": round(secs / n_launch * 1e6, 2),
                         "flops_per_step": flops, "gemm_time_share_of_step": round(secs * 1e3 / ms, 3),
                         "by_block": {"backbone_head_convs": split.get("conv"), "cft_linears": split.get("linear")},
                         "top_shapes": [{"shape": k, "launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1),
                                         "ms": round(v[2] * 1e3, 3)} for k, v in top]},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(cfg, {k: v for k, v in model.cpu().state_dict().items()},
                                                args.size, args.size)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
