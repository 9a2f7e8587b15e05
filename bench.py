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

# HIP runtime knob, read when the runtime initialises (so: before torch is imported): kernel arguments are written straight into device
# memory instead of being fetched from host memory at dispatch.  A forward is ~480 launches per HIP-graph replay, many of them 10-20 us
# long: +1.7 % pairs/s with two forwards in flight, +1.9 % with one (profiles/r03_forwards_in_flight.txt).  A deployment sets the same
# variable in its environment; an explicit setting of the caller wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

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
    fam, aux = {}, {}
    abytes = 0.0
    for name, flops, e0, e1, ab in log:
        gemm = not name.startswith("cft_")       # cft_*: the non-GEMM kernels of the CFT block (attention, LN, (de)tokeniser)
        if gemm:
            abytes += ab
        f = (fam if gemm else aux).setdefault(name, [0, 0.0, 0.0, 0.0])
        f[0] += 1
        f[1] += flops
        f[2] += max(e0.elapsed_time(e1) * 1e-3 - ovh, 1e-7)
        f[3] += ab
    n = sum(v[0] for v in fam.values())
    flops = sum(v[1] for v in fam.values())
    secs = sum(v[2] for v in fam.values())
    return n, flops, secs, fam, ovh, abytes, aux


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def traffic_from_profile(args, n_launch, abytes):
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh ->
    profiles/*_traffic.json: FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, summed
    over the conv_gemm family of one forward).  Only reported for the configuration it was collected on."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_traffic.json") for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(q)), None)
    if path is None:
        return None
    t = json.load(open(path))
    if (t.get("config"), t.get("batch"), t.get("size"), t.get("dtype")) != (args.config, args.batch, args.size, args.dtype):
        return None
    # forward_kernels_*: the kernels of a forward; all_kernels_* (what rounds 1-3 quoted) also holds the process's one-time set-up traffic
    return {"bytes_per_launch": t["gemm_bytes_per_forward"] / n_launch, "algorithmic_bytes_per_launch": abytes / n_launch,
            "forward_kernels_bytes_per_forward": t.get("forward_kernels_bytes_per_forward", t.get("all_kernels_bytes_per_forward")),
            "all_kernels_bytes_per_forward": t.get("all_kernels_bytes_per_forward"),
            "source": f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"}


def cpu_baseline(cfg, sd, rgb, ir, budget_s=30.0):
    """Reported baseline only: the oracle (a port of the reference forward to plain torch fp32; the reference tree
    itself is not on the GPU box) on the host cores, as BASELINE.md section 2 prescribes: batch min(B, 8), fused
    weights, and the BEST of a sweep over intra-op thread counts {8, 16, 32, 64, nproc} - more threads is not
    faster on a many-core host (round 1 ran 128 threads and lost 4x).  Bounded to ~budget_s of CPU work."""
    from oracle.cft_oracle import OracleModel
    b, height, width = rgb.shape[0], rgb.shape[2], rgb.shape[3]     # the first min(B, 8) pairs of the timed batch itself
    om = OracleModel(cfg)
    ncpu = os.cpu_count() or 8
    default_threads = torch.get_num_threads()
    sweep = sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu})
    t0 = time.perf_counter()
    om(sd, rgb[:1], ir[:1])
    warm = time.perf_counter() - t0
    spent, results = warm, {}
    try:
        for t in sweep:
            if results and spent + min(results.values()) > budget_s * 0.7:
                break
            if len(results) >= 2 and list(results.values())[-1] > 1.5 * min(results.values()):
                break                                     # past the optimum: wider only gets slower (256 threads: 200 s per forward)
            torch.set_num_threads(t)
            t0 = time.perf_counter()
            om(sd, rgb[:1], ir[:1])                       # thread-pool warm-up at this width
            spent += time.perf_counter() - t0
            t0 = time.perf_counter()
            want = om(sd, rgb, ir)
            results[t] = time.perf_counter() - t0
            spent += results[t]
        best = min(results, key=results.get)
        torch.set_num_threads(best)
        times = [results[best]]
        while spent + statistics.median(times) < budget_s and len(times) < 5:
            t0 = time.perf_counter()
            om(sd, rgb, ir)
            times.append(time.perf_counter() - t0)
            spent += times[-1]
    finally:
        torch.set_num_threads(default_threads)
    med = statistics.median(times)
    cpu_baseline.best_threads = best
    return want, {"value": round(b / med, 3), "unit": "image-pairs/sec", "cores": best, "kind": "port",
            "sample": f"oracle/cft_oracle.py fp32, same network and {height}x{width} inputs, batch {b}, median of "
                      f"{len(times)} forwards at the best thread count ({med:.2f} s each); sweep pairs/s by threads: "
                            + ", ".join(f"{t}: {b / v:.2f}" for t, v in sorted(results.items()))
                      + f"; os.cpu_count()={ncpu}; {spent:.0f} s of CPU work in total"}


BOUNDS = {"f32": ("raw logits max-abs", 1e-3), "f16": ("sigmoid-space max-abs", 1e-2), "bf16": ("sigmoid-space max-abs", 2.5e-2)}
# What a 16-bit leg is gated on, in words: fp16 meets north_star's literal 1e-2 and is the declared parity-green 16-bit mode;
# bf16 cannot on these weights (two thirds of its error is the single rounding of the WEIGHTS, profiles/r04_bf16_sites.md: the reference's own
# bf16-autocast forward is at the same 1.4e-2) and is gated on that reference level instead.
GATES = {"f32": "north_star 1e-3 on raw logits vs the fp32 oracle",
         "f16": "north_star 1e-2 (sigmoid space) vs the fp32 oracle - the parity-green 16-bit mode, the reference's own GPU precision (test.py:66-68)",
         "bf16": "NOT the literal 1e-2: <= 2.5e-2 AND at least as close to fp32 as the reference's own bf16-autocast forward on the same weights "
                 "and inputs (max <= 1.30x + 1e-3, rms <= 1.10x); see parity_green_dtype for the 16-bit mode that meets 1e-2"}
HBM_ACHIEVABLE_TBPS, HBM_SPEC_TBPS = 6.29, 8.0      # MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured (float4 copy)
HBM_MIXED_TBPS = 5.0      # tools/micro/hbm_rw.hip on the same chip: reads alone reach 6.2 - 7.2 TB/s, writes 4.4 - 5.4, read + write streams 4.8 - 5.7
REF_BF16_MAX_RATIO, REF_BF16_RMS_RATIO = 1.30, 1.10     # as tests/test_gpu_model.py: HIP bf16 error level vs the reference-style bf16 forward


def _flat(raws):
    return torch.cat([r.reshape(-1).float() for r in raws])


def parity_at_bench_shape(dtype_name, got_raw, want_raw, ref16_raw=None, n_ref=0):
    """The timed configuration checked against the oracle: ``got_raw`` are the head logits of the first
    pairs of the replayed batch, ``want_raw`` the CPU oracle's for the same pairs (fp32).  16-bit runs are compared in sigmoid
    space; bf16 additionally against the error level of the reference-style bf16 forward (oracle under CPU
    autocast, pinned to the reference's own autocast outputs by tests/test_oracle_golden.py) on the first ``n_ref`` pairs."""
    what, bound = BOUNDS[dtype_name]
    g, w = _flat(got_raw), _flat(want_raw)
    err = (g - w).abs().max().item() if dtype_name == "f32" else (g.sigmoid() - w.sigmoid()).abs().max().item()
    out = {"pairs": int(got_raw[0].shape[0]), "check": what, "gate": GATES[dtype_name], "max_err": round(err, 6), "bound": bound,
           "meets_north_star_bound": bool(err <= (1e-3 if dtype_name == "f32" else 1e-2)),
           "rms_logit_err_over_std": round(((g - w).pow(2).mean().sqrt() / w.std()).item(), 6), "ok": bool(err <= bound)}
    if ref16_raw is not None and n_ref:
        g2, w2, r2 = _flat([r[:n_ref] for r in got_raw]), _flat([r[:n_ref] for r in want_raw]), _flat(ref16_raw)
        e_hip = (g2.sigmoid() - w2.sigmoid()).abs().max().item()
        e_ref = (r2.sigmoid() - w2.sigmoid()).abs().max().item()
        rms_hip = ((g2 - w2).pow(2).mean().sqrt() / w2.std()).item()
        rms_ref = ((r2 - w2).pow(2).mean().sqrt() / w2.std()).item()
        level_ok = e_hip <= REF_BF16_MAX_RATIO * e_ref + 1e-3 and rms_hip <= REF_BF16_RMS_RATIO * rms_ref + 2e-4
        out["vs_reference_style_bf16"] = {"pairs": n_ref, "hip_max_err": round(e_hip, 6), "reference_bf16_max_err": round(e_ref, 6),
                                          "hip_rms": round(rms_hip, 6), "reference_bf16_rms": round(rms_ref, 6), "ok": bool(level_ok)}
        out["ok"] = bool(out["ok"] and level_ok)
    return out


def _input_sensitivity(want):
    """rms change of the fp32 oracle's logits from one image pair of the batch to the next: what a parity bound on these weights can see"""
    if want[0].shape[0] < 2:
        return None
    a, b = _flat([r[:-1] for r in want]), _flat([r[1:] for r in want])
    return round((a - b).pow(2).mean().sqrt().item(), 6)


def ladder_weights_parity(cfg, args, dev, dtype, rgb, ir, n_pairs=8):
    """The benchmarked configuration on the HIGHEST rung of the gain ladder (utils/seeded.ladder_state_dict, BENCH_LADDER_GAIN) on which the
    reference's own bf16-autocast forward still meets 1e-2 (tests/golden/ladder_ref.pt, recorded from the reference: 8.9e-3 at 256 x 256,
    the logits moving 4.2e-2 rms between image pairs): north_star's literal bound asserted outright on weights where it CAN fail -
    VERDICT r5 item 2.  Same batch, shape, compute dtype and HIP-graph replay as the timed run; the first ``n_pairs`` pairs vs the fp32
    oracle (pinned to the reference on this rung by tests/test_oracle_golden.py)."""
    from msod_amd.utils.seeded import BENCH_LADDER_GAIN, ladder_state_dict
    from oracle.cft_oracle import OracleModel
    m2 = Model(cfg)
    m2.load_state_dict(ladder_state_dict(m2.state_dict(), BENCH_LADDER_GAIN, 5))
    m2.fuse()
    sd2 = {k: v.clone() for k, v in m2.state_dict().items()}
    m2 = m2.to(dev).set_compute_dtype(dtype)
    n_pairs = min(n_pairs, args.batch)
    with torch.no_grad():
        if args.no_graph:
            _, raw = m2.forward_once(rgb, ir)
        else:
            m2.capture(args.batch, args.size, args.size)
            _, raw = m2(rgb, ir)
        torch.cuda.synchronize()
        got = [r[:n_pairs].float().cpu() for r in raw]
    m2.release_graphs()
    _, want = OracleModel(cfg)(sd2, rgb[:n_pairs].cpu(), ir[:n_pairs].cpu())
    g, w = _flat(got), _flat(want)
    f32 = dtype == torch.float32
    err = (g - w).abs().max().item() if f32 else (g.sigmoid() - w.sigmoid()).abs().max().item()
    bound = 1e-3 if f32 else 1e-2
    return {"weights": f"gain-ladder rung {BENCH_LADDER_GAIN} (utils/seeded.ladder_state_dict; the seeded stress weights are rung 1.45)", "pairs": n_pairs,
            "check": BOUNDS[args.dtype][0], "gate": "north_star's literal bound, asserted outright", "max_err": round(err, 6), "bound": bound,
            "meets_north_star_bound": bool(err <= bound), "logit_std": round(w.std().item(), 4), "input_sensitivity_rms": _input_sensitivity(want),
            "rms_logit_err_over_std": round(((g - w).pow(2).mean().sqrt() / w.std()).item(), 6), "ok": bool(err <= bound)}


def survey_weights_parity(cfg, args, dev, dtype, rgb, ir, n_pairs=8):
    """The benchmarked configuration once more, on the weights SURVEY.md section 8c / BASELINE.md section 2 literally prescribe
    (utils/seeded.survey_state_dict: torch.manual_seed(0) constructor weights, BatchNorm statistics / affine and pos_emb
    randomised; pinned to the reference's constructor and forward in tests/test_oracle_golden.py): same batch, shape, compute
    dtype and HIP-graph replay; the first ``n_pairs`` pairs vs the fp32 oracle, north_star's literal bound (1e-2 in sigmoid space for
    the 16-bit types, 1e-3 on raw logits for fp32)."""
    from msod_amd.utils.seeded import survey_state_dict
    from oracle.cft_oracle import OracleModel
    n_pairs = min(n_pairs, args.batch)
    m2 = Model(cfg)
    m2.load_state_dict(survey_state_dict(lambda: Model(cfg), seed=0))
    m2.fuse()
    sd2 = {k: v.clone() for k, v in m2.state_dict().items()}
    m2 = m2.to(dev).set_compute_dtype(dtype)
    with torch.no_grad():
        if args.no_graph:
            _, raw = m2.forward_once(rgb, ir)
        else:
            m2.capture(args.batch, args.size, args.size)
            _, raw = m2(rgb, ir)
        torch.cuda.synchronize()
        got = [r[:n_pairs].float().cpu() for r in raw]
    m2.release_graphs()
    _, want = OracleModel(cfg)(sd2, rgb[:n_pairs].cpu(), ir[:n_pairs].cpu())
    g, w = _flat(got), _flat(want)
    f32 = dtype == torch.float32
    err = (g - w).abs().max().item() if f32 else (g.sigmoid() - w.sigmoid()).abs().max().item()
    bound = 1e-3 if f32 else 1e-2
    return {"weights": "SURVEY.md 8c recipe (utils/seeded.survey_state_dict)", "pairs": n_pairs, "check": BOUNDS[args.dtype][0],
            "gate": "north_star's literal bound, asserted outright", "max_err": round(err, 6), "bound": bound,
            "meets_north_star_bound": bool(err <= bound), "logit_std": round(w.std().item(), 4),
            "input_sensitivity_rms": _input_sensitivity(want),      # ~1e-5: the logits hardly depend on the images on these weights - this entry cannot
            "note": "the logits move by input_sensitivity_rms between image pairs on these weights: a bound of 1e-2 cannot fail here; "   # fail
                    "parity_at_bench_shape_ladder is the entry that can",
            "rms_logit_err_over_std": round(((g - w).pow(2).mean().sqrt() / w.std()).item(), 6), "ok": bool(err <= bound)}


class ShaderClockProbe:
    """Shader clock the SIMDs run at while a leg is in flight, read from inside the GPU: every ``period`` seconds a thread
    launches ``cft_clock_probe`` (one wave, ``spin_us`` of the constant-rate wall clock) on its own stream next to the forward;
    the kernel returns its s_memtime ticks, its wall-clock ticks and the dependent FMAs it executed.  ``summary()`` gives MHz
    from ticks / wall time (when s_memtime runs at the shader clock) and, independent of that, the FMA rate relative to a probe
    taken on the idle GPU (a dependent v_fma_f32 chain costs a fixed number of shader cycles)."""

    def __init__(self, dev, period=0.25, spin_us=200, max_samples=256):
        import ctypes
        import threading
        from msod_amd import _lib
        self.lib, self.ct = _lib.load(), ctypes
        self.dev, self.period, self.spin_us, self.max = dev, period, spin_us, max_samples
        self.buf = torch.zeros((max_samples + 1, 4), dtype=torch.int64, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.khz = ctypes.c_int(0)
        self.n, self.stop = 0, threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.idle = None

    def _launch(self, slot):
        st = self.lib.cft_clock_probe(self.buf[slot].data_ptr(), self.spin_us, self.ct.byref(self.khz), self.stream.cuda_stream)
        return st == 0

    def measure_idle(self):
        torch.cuda.synchronize(self.dev)
        time.sleep(0.2)
        if self._launch(self.max):
            self.stream.synchronize()
            self.idle = self.buf[self.max].tolist()

    def _run(self):
        torch.cuda.set_device(self.dev)
        while not self.stop.is_set() and self.n < self.max:
            if self._launch(self.n):
                self.n += 1
            self.stop.wait(self.period)

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        self.thread.join(5)
        self.stream.synchronize()

    def summary(self):
        rows = [r for r in self.buf[:self.n].tolist() if r[1] > 0]
        if not rows or self.khz.value <= 0:
            return None
        wall_mhz = self.khz.value / 1e3
        mhz = sorted(r[0] / r[1] * wall_mhz for r in rows)
        fma = sorted(r[2] / r[1] * wall_mhz for r in rows)          # dependent FMAs per microsecond
        out = {"samples": len(rows), "s_memtime_mhz": {"median": round(mhz[len(mhz) // 2], 1), "min": round(mhz[0], 1), "max": round(mhz[-1], 1)},
               "dependent_fma_per_us": {"median": round(fma[len(fma) // 2], 1), "min": round(fma[0], 1), "max": round(fma[-1], 1)},
               "method": "cft_clock_probe: a one-wave kernel on a side stream every 0.25 s (s_memtime / s_memrealtime ticks and a dependent v_fma_f32 chain over 200 us)"}
        if self.idle and self.idle[1] > 0:
            out["idle_gpu"] = {"s_memtime_mhz": round(self.idle[0] / self.idle[1] * wall_mhz, 1), "dependent_fma_per_us": round(self.idle[2] / self.idle[1] * wall_mhz, 1)}
            out["fma_rate_vs_idle"] = round(out["dependent_fma_per_us"]["median"] / out["idle_gpu"]["dependent_fma_per_us"], 4)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="image pairs per GPU")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--no-concat-plan", action="store_true", help="A/B: let Concat copy all its sources")
    ap.add_argument("--no-conv-chain", action="store_true", help="A/B: the stride-2 Conv in front of a C3 and the C3's cv1|cv2 as two launches (round 3) instead of one")
    ap.add_argument("--no-pair-chain", action="store_true", help="A/B: the 256-channel Bottleneck pairs (3x3 [+ shortcut] + next 1x1) as two launches each "
                    "(the 3x3 then runs on the hand-scheduled kernel) instead of the chained 16-wave kernel")
    ap.add_argument("--no-cft-fusion", action="store_true", help="A/B: de-tokenise + Add2 per stream and Add as three launches (round 3) instead of one")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-f16-leg", action="store_true", help="skip the extra fp16 measurement (N = 1, 16-bit runs only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run both backbones on one HIP stream")
    ap.add_argument("--sustained-steps", type=int, default=500, help="extra >= 10 s leg at N = 1 (0 = skip)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed configuration")
    ap.add_argument("--in-flight", type=int, default=0, help="forwards in flight (own graphs, buffers and streams each); 1 = one at a time; 0 (default) = 3 from 32 pairs "
                    "per GPU on, 4 below (8 pairs: 2 719 pairs/s with 4, 2 553 with 3, 2 360 with the rounds-3-5 mode: profiles/r06_forwards_in_flight.txt)")
    ap.add_argument("--fly-two-streams", action="store_true", help="A/B: the forwards in flight keep their two backbone streams (rounds 3-5 ran 2 such forwards); "
                    "by default, from 3 in flight on, each forward in flight is captured on ONE stream (no stream-group lottery: profiles/r06_forwards_in_flight.txt) "
                    "and the two-stream graph serves the one-at-a-time leg")
    ap.add_argument("--force-gather", action="store_true", help="N = 1: run the detection all-gather anyway (RCCL with world size 1, OverlappedGather "
                    "inside the timed steps) and report multi_gpu_selfcheck - exercises the N > 1 step mode on the one GPU of a box")
    ap.add_argument("--no-splitk", action="store_true", help="A/B: the CFT blocks' out_proj / fc2 as one launch each (round 4) instead of split-K + LayerNorm-reduce")
    ap.add_argument("--depth-first", default="", help="CHUNKS[,ROWS]: Model.depth_first - the image-only prefix of each backbone sub-batch by sub-batch "
                    "(Infinity-Cache residency); empty = layer by layer over the whole batch")
    ap.add_argument("--stream-priorities", default="", help="A/B: HIP stream priority per forward in flight, e.g. -1,0 (ForwardPipeline.pick_streams(priorities=...); "
                    "round 4 measured (-1, 0) and (-1, -1) slower than equal priorities, profiles/r04_forwards_in_flight.txt)")
    ap.add_argument("--conv-variant", type=int, default=0, help="A/B runs: cft_set_conv_variant() for the whole process (0 = automatic)")
    args = ap.parse_args()

    relaunch = D.launch_command(os.path.abspath(__file__), args.gpus, sys.argv[1:])
    if relaunch is not None:       # --gpus N > 1 without a launcher: N ranks under torch.distributed.run (never one rank posing as N)
        log("re-launching: " + " ".join(relaunch))
        os.execvp(relaunch[0], relaunch)
    exit_code = 0
    rank, world, local = D.init_from_env(force=args.force_gather)
    gathering = world > 1 or args.force_gather
    if world != args.gpus and (world > 1 or args.gpus > 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]

    if args.conv_variant:
        from msod_amd import _lib
        _lib.load().cft_set_conv_variant(args.conv_variant)
    log(f"rank {rank}/{world} building {args.config}")
    cfg = named_config(args.config)
    model = Model(cfg)
    sd = seeded_state_dict(model.state_dict(), seed=0)      # random-init weights, BN/pos_emb non-trivial
    model.load_state_dict(sd)
    model = model.to(dev).fuse().set_compute_dtype(dtype)   # deployed form: BN folded (attempt_load does .fuse())
    model.overlap_streams = not args.no_overlap
    model.plan_concats = not args.no_concat_plan
    model.fuse_cft_outputs = not args.no_cft_fusion
    model.chain_convs = not args.no_conv_chain
    if args.no_pair_chain:
        from msod_amd.models.common import C3
        C3.chain_pairs = False
    model.splitk = not args.no_splitk
    if args.depth_first:
        df = [int(v) for v in args.depth_first.split(",")]
        model.depth_first = (df[0], df[1] if len(df) > 1 else None)
    rgb, ir = seeded_inputs(args.batch, args.size, args.size, seed=rank)
    rgb, ir = rgb.to(dev), ir.to(dev)

    log("weights loaded, packing + capturing")
    if args.in_flight <= 0:
        args.in_flight = 3 if args.batch >= 32 else 4
    k_fly = 1 if args.no_graph else max(1, args.in_flight)
    fly_single = False
    with torch.no_grad():
        if args.no_graph:
            step_seq = lambda: model.forward_once(rgb, ir)   # noqa: E731
            caps = []
        else:
            from msod_amd.graph import CapturedForward
            # From 3 forwards in flight on, each forward in flight runs its two backbones on ONE stream: K single-stream graphs fill the chip as
            # well as 2 two-stream ones (+2 % measured) and the throughput no longer depends on WHICH HIP streams replay them (the two attractors of
            # rounds 3-5, 7 % apart, collapse to one).  caps[0] stays the two-stream graph: best latency for one forward at a time.
            fly_single = k_fly >= 3 and not args.fly_two_streams and not args.no_overlap
            caps = [model.capture(args.batch, args.size, args.size)]
            if fly_single:
                model.overlap_streams = False
                fly_caps = [CapturedForward(model, args.batch, args.size, args.size) for _ in range(k_fly)]
                model.overlap_streams = True
                caps += fly_caps
            else:
                caps += [CapturedForward(model, args.batch, args.size, args.size) for _ in range(k_fly - 1)]
                fly_caps = caps
            for c in caps:
                c.rgb.copy_(rgb)
                c.ir.copy_(ir)
            step_seq = caps[0].replay_static
        pred, _ = step_seq()
        # N > 1: the all-gather of step i runs on RCCL's stream while the forward of step i+1 runs on the compute
        # stream (51.6 MB per rank per step would otherwise add ~10 % serial time); see distributed.OverlappedGather.
        gather = D.OverlappedGather(pred, world) if gathering else None
        # A step = one forward over one batch of `--batch` pairs.  With --in-flight K (default 3 at >= 32 pairs per GPU, 4 below) step t replays graph t % K on HIP
        # stream t % K: consecutive steps are independent batches and overlap on the GPU (distributed.ForwardPipeline) - the
        # low-occupancy stretches of one forward (CFT blocks with M = 8192, the single-stream head) run under the other's
        # backbone convolutions.  --in-flight 1 = one forward at a time (also reported as "single_in_flight").
        stream_probe_ms = None
        if k_fly > 1:
            runners = [(lambda c=c: c.replay_static()[0]) for c in fly_caps]
            prios = [int(v) for v in args.stream_priorities.split(",")] if args.stream_priorities else None
            fly_streams, stream_probe_ms = D.ForwardPipeline.pick_streams(runners, dev, priorities=prios)      # untimed set-up: the stream group that overlaps best
            pipe = D.ForwardPipeline(runners, fly_streams, gather)
            step_fn = pipe.step
        else:
            step_fn = lambda: step_seq()[0]   # noqa: E731
        torch.cuda.synchronize()
        # the measurement loop (warm-up, barrier + synchronize on both sides, MAX over ranks) is distributed.timed_steps:
        # the same code runs under tests/test_distributed_gloo.py with two CPU ranks
        elapsed = D.timed_steps(step_fn, args.steps, max(args.warmup, k_fly), world=world, gather=gather, sync=torch.cuda.synchronize)
    assert torch.isfinite(pred).all(), "non-finite detections"
    assert all(torch.equal(c.pred, pred) for c in caps[1:]), "the graphs in flight disagree"
    log(f"timed region: {elapsed:.3f} s for {args.steps} steps ({k_fly} in flight)")
    local_elapsed = getattr(D.timed_steps, "last_local_elapsed", elapsed)
    n_par = 0 if args.no_parity else min(args.batch, 8)     # the first 8 pairs of the timed batch, at every N (N > 1 lines checked 2)
    with torch.no_grad():
        raw_now = step_seq()[1]
        torch.cuda.synchronize()
    got_raw = [r[:n_par].float().cpu() for r in raw_now] if n_par else None
    single = None
    if world == 1 and k_fly > 1:      # the same K steps, one forward at a time
        with torch.no_grad():
            el1 = D.timed_steps(lambda: step_seq()[0], args.steps, args.warmup, sync=torch.cuda.synchronize)
        single = {"value": round(args.batch * args.steps / el1, 2), "unit": "image-pairs/sec", "ms_per_step": round(el1 / args.steps * 1e3, 3),
                  "steps": args.steps, "note": "one forward in flight (each step starts when the previous one has finished)"}
        log(f"single in flight: {single}")
    selfcheck = None
    if gathering:       # evidence that every rank took part and that the gathered rows are the ranks' own rows
        selfcheck = D.gather_selfcheck(pred, gather.drain(), rank, world, elapsed_local=local_elapsed, force=args.force_gather)
        if args.force_gather and world == 1:
            selfcheck["note"] = "forced at N = 1: the collectives ran through RCCL with world size 1"
    sustained = None
    if world == 1 and args.sustained_steps > 0 and not args.no_graph:
        probe = None
        try:
            probe = ShaderClockProbe(dev)
            probe.measure_idle()
        except Exception as e:  # noqa: BLE001 - the probe is an extra; the leg is measured without it
            log(f"shader clock probe unavailable: {e!r}")
            probe = None
        import contextlib
        with (probe or contextlib.nullcontext()), torch.no_grad():
            el_s = D.timed_steps(step_fn, args.sustained_steps, 2, sync=torch.cuda.synchronize)
        sustained = {"steps": args.sustained_steps, "seconds": round(el_s, 3), "value": round(args.batch * args.sustained_steps / el_s, 2),
                     "unit": "image-pairs/sec", "ms_per_step": round(el_s / args.sustained_steps * 1e3, 3),
                     "shader_clock_under_load": probe.summary() if probe is not None else None}
        sc = sustained["shader_clock_under_load"]
        if sc and sc["s_memtime_mhz"]["median"] > 500 and dtype != torch.float32:
            # the dense MFMA peak of MI355X_MICROARCH.md is quoted at the 2.4 GHz boost clock; this is the same machine at the clock it held
            sc["bf16_mfma_peak_at_median_clock_tflops"] = round(PEAK_BF16_TFLOPS * sc["s_memtime_mhz"]["median"] / 2400.0, 1)
        log(f"sustained leg: {sustained}")

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        n_launch, flops, secs, fam, ovh, abytes, aux = gemm_family_time(model, rgb, ir)
        log(f"gemm family: {n_launch} launches, {secs * 1e3:.2f} ms, {flops / secs / 1e12:.1f} TFLOP/s")
        top = sorted(fam.items(), key=lambda kv: -kv[1][2])[:6]
        peak = PEAK_F32_TFLOPS if dtype == torch.float32 else PEAK_BF16_TFLOPS     # bf16 and fp16 MFMA have the same dense peak

        def family_row(v):
            """Per-shape floors (DESIGN.md section 9's additive law as data): per launch, the MFMA time of the shape's
            algorithmic FLOPs at the dense peak, the HBM time of its algorithmic bytes (inputs once + outputs once + weights) at the
            achievable HBM rate, and measured / (mfma + hbm)."""
            n, us = v[0], v[2] / v[0] * 1e6
            mfma_us, hbm_us = v[1] / n / (peak * 1e12) * 1e6, v[3] / n / (HBM_ACHIEVABLE_TBPS * 1e12) * 1e6
            return {"launches": n, "gflop": v[1] / 1e9, "ms": v[2] * 1e3, "tflops": v[1] / v[2] / 1e12, "mbytes_per_launch": round(v[3] / n / 1e6, 2),
                    "us_per_launch": round(us, 2), "mfma_us": round(mfma_us, 2), "hbm_us": round(hbm_us, 2),
                    "measured_over_mfma_plus_hbm": round(us / (mfma_us + hbm_us), 3), "measured_over_max_floor": round(us / max(mfma_us, hbm_us), 3),
                    "excess_ms": round((us - (mfma_us + hbm_us)) * n * 1e-3, 3)}
        fam_rows = {k: family_row(v) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][2])}
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            with open(os.path.join(ROOT, "gpurun_out", "bench_families.json"), "w") as fh:
                json.dump({**fam_rows, **{k: family_row(v) for k, v in aux.items()}}, fh, indent=1)
        # the five shapes that lose the most time against their own floors (time above mfma + hbm, summed over their launches)
        worst = sorted(fam_rows.items(), key=lambda kv: -kv[1]["excess_ms"])[:5]
        achieved = flops / secs / 1e12
        split = {}
        for kind in ("conv", "linear"):     # convolutions vs the nn.Linear GEMMs of the CFT (GPT) blocks
            fl = sum(v[1] for k, v in fam.items() if k.startswith(kind))
            tt = sum(v[2] for k, v in fam.items() if k.startswith(kind))
            if tt > 0:
                split[kind] = {"tflops": round(fl / tt / 1e12, 1), "frac": round(fl / tt / 1e12 / peak, 4), "ms": round(tt * 1e3, 3)}
        # the whole CFT (GPT) block: its linears (GEMM family) + attention + LayerNorm + tokeniser + de-tokeniser
        lin = split.get("linear") or {"ms": 0.0}
        cft_flops = sum(v[1] for k, v in fam.items() if k.startswith("linear")) + sum(v[1] for v in aux.values())
        cft_ms = lin["ms"] + sum(v[2] for v in aux.values()) * 1e3
        cft_block = None
        if cft_ms > 0:
            cft_bytes = sum(v[3] for k, v in fam.items() if k.startswith("linear")) + sum(v[3] for v in aux.values())
            cft_mfma_ms, cft_hbm_ms = cft_flops / (peak * 1e12) * 1e3, cft_bytes / (HBM_ACHIEVABLE_TBPS * 1e12) * 1e3
            # The block's TRUE minimum traffic: its feature maps in (tokeniser), the bases re-read and the three maps
            # written by the output stage, and every linear's weights once - what an implementation that keeps tokens / QKV / hidden layers
            # on chip would move.  The per-kernel figure (every kernel's inputs + outputs once) is kept as hbm_ms_per_kernel_tensors.
            import re as _re
            es_ = 4 if dtype == torch.float32 else 2
            w_bytes = sum(v[0] * int(m_.group(1)) * int(m_.group(2)) * es_ for k, v in fam.items() if k.startswith("linear")
                          for m_ in [_re.match(r"linear_k1s1_n(\d+)_K(\d+)", k)] if m_)
            io_bytes = sum(v[3] for k, v in aux.items() if k in ("cft_tokenize", "cft_upsample_add"))
            blk_min_ms = (w_bytes + io_bytes) / (HBM_ACHIEVABLE_TBPS * 1e12) * 1e3
            cft_block = {"tflops": round(cft_flops / cft_ms / 1e9, 1), "frac": round(cft_flops / cft_ms / 1e9 / peak, 4), "ms": round(cft_ms, 3),
                         "floors": {"mfma_ms": round(cft_mfma_ms, 3),
                                    "hbm_ms_block_minimum": round(blk_min_ms, 3), "block_minimum_gbytes": round((w_bytes + io_bytes) / 1e9, 3),
                                    "hbm_ms_per_kernel_tensors": round(cft_hbm_ms, 3),
                                    "bound_by_roofline": "mfma" if cft_mfma_ms >= blk_min_ms else "hbm",
                                    "note": "block minimum = feature maps in + bases re-read + three maps out + weights once (tokens, QKV, hidden layers on chip); "
                                            "per_kernel_tensors = every launched kernel's inputs once + outputs once (what this decomposition moves at best)",
                                    "max_frac_of_mfma_peak_with_this_decomposition_if_mfma_and_hbm_overlap": round(cft_mfma_ms / max(cft_mfma_ms, cft_hbm_ms), 4),
                                    "max_frac_with_this_decomposition_if_they_add": round(cft_mfma_ms / (cft_mfma_ms + cft_hbm_ms), 4)},
                         "launches": sum(v[0] for k, v in fam.items() if k.startswith("linear")) + sum(v[0] for v in aux.values()),
                         "parts_ms": {"linears": lin["ms"], **{k[4:]: round(v[2] * 1e3, 3) for k, v in aux.items()}}}
        line = {
            "metric": "image-pairs/sec fwd, yolov5l+CFTx3 640\u00d7640 bs64, 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "image-pairs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.config} ({WORKLOADS.get(args.config, args.config)}) two-stream forward, "
                                   f"{args.size}x{args.size}, {args.batch} pairs/GPU, BN folded, pre-NMS detections",
                       "pairs_per_gpu": args.batch, "image_size": args.size, "parallelism": f"batch-shard x{world}",
                       "hip_graph": not args.no_graph, "two_hip_streams": not args.no_overlap and not fly_single, "forwards_in_flight": k_fly,
                       **({"streams_per_forward_in_flight": 1, "single_in_flight_streams": 2} if fly_single else {}), "env": {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")},
                       **({"stream_group_probe_ms_per_step": stream_probe_ms} if stream_probe_ms else {}),
                       **({"depth_first": args.depth_first} if args.depth_first else {}), **({"stream_priorities": args.stream_priorities} if args.stream_priorities else {}),
                       **({"conv_variant": args.conv_variant} if args.conv_variant else {})},
            "sustained": sustained, "single_in_flight": single, "multi_gpu_selfcheck": selfcheck,
            "roofline": {"bound": "mfma", "kernel": "conv_gemm_kernel (implicit-GEMM conv/linear family; incl. the dedicated Focus kernel and the fused 64- / 128-channel Bottleneck kernels: 2 + 27 launches of the cfg3 forward)",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic_from_profile(args, n_launch, abytes), "launches_per_step": n_launch,
                         "event_bracket_overhead_us": round(ovh * 1e6, 2),
                         "avg_launch_us": round(secs / n_launch * 1e6, 2),
                         "flops_per_step": flops, "gemm_time_share_of_step": round(secs * 1e3 / ms, 3),
                         # per-launch times come from ONE forward on ONE stream; in the timed steps launches of two forwards (and of the
                         # two backbones) overlap, so their sum may exceed ms_per_step.  whole_step = all algorithmic FLOPs / ms_per_step.
                         "whole_step": {"tflops": round(flops / (ms * 1e-3) / 1e12 / max(world, 1) * world, 1), "frac": round(flops / (ms * 1e-3) / 1e12 / peak, 4)},
                         "by_block": {"backbone_head_convs": split.get("conv"), "cft_linears": split.get("linear"),
                                      "cft_block_whole": cft_block},
                         "top_shapes": [{"shape": k, "launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1),
                                         "ms": round(v[2] * 1e3, 3)} for k, v in top],
                         "worst_shapes_vs_own_floors": [{"shape": k, "launches": r["launches"], "us_per_launch": r["us_per_launch"], "mfma_us": r["mfma_us"],
                                                         "hbm_us": r["hbm_us"], "measured_over_mfma_plus_hbm": r["measured_over_mfma_plus_hbm"],
                                                         "excess_ms": r["excess_ms"]} for k, r in worst]},
        }
        # Both floors of the whole step: the matrix floor at the dense peak and the HBM floor on the step's own bytes
        # (algorithmic = every logged kernel's inputs once + outputs once; counters = the PMC profile of the same configuration).  The
        # larger one binds: for cfg3 at 64 pairs that is HBM, although 99.6 % of the FLOPs are MFMA work.
        rl = line["roofline"]
        alg_bytes = abytes + sum(v[3] for v in aux.values())
        tr = rl["traffic"] or {}
        floors = {"mfma_ms": round((flops + sum(v[1] for v in aux.values())) / (peak * 1e12) * 1e3, 3),
                  "hbm_ms_algorithmic": round(alg_bytes / (HBM_ACHIEVABLE_TBPS * 1e12) * 1e3, 3),
                  "hbm_ms_counters": round(tr["forward_kernels_bytes_per_forward"] / (HBM_ACHIEVABLE_TBPS * 1e12) * 1e3, 3) if tr.get("forward_kernels_bytes_per_forward") else None,
                  "hbm_ms_counters_at_measured_mixed_rate": round(tr["forward_kernels_bytes_per_forward"] / (HBM_MIXED_TBPS * 1e12) * 1e3, 3) if tr.get("forward_kernels_bytes_per_forward") else None,
                  "algorithmic_gbytes_per_step": round(alg_bytes / 1e9, 2),
                  "hbm_rate": f"{HBM_ACHIEVABLE_TBPS} TB/s achievable (spec {HBM_SPEC_TBPS}; read + write streams measure {HBM_MIXED_TBPS} on this chip, profiles/r04_hbm_rw_microbench.txt: informational field only); algorithmic bytes cover the GEMM family and the CFT pointwise kernels "
                              "(SPP / concat copies / Add / Detect decode are not logged: < 2 % of the bytes)"}
        binding = max(v for v in (floors["mfma_ms"], floors["hbm_ms_algorithmic"], floors["hbm_ms_counters"]) if v is not None)
        rl["floors"] = floors
        rl["frac_of_binding_floor"] = round(binding / ms, 4)
        rl["whole_step_bound"] = "hbm" if binding > floors["mfma_ms"] else "mfma"
        rl["bound_note"] = ("'bound' names what bounds the dominant KERNEL FAMILY the achieved / peak pair is about (MFMA: 99.6 % of the FLOPs); the whole "
                            "step is bound by 'whole_step_bound' - see 'floors'")
        if args.dtype == "bf16" and world == 1 and not args.no_f16_leg and not args.no_graph:
            # the same step in fp16 - the 16-bit type that meets the 1e-2 parity bound: same MFMA
            # rate and bytes as bf16, measured with the same loop
            model.set_compute_dtype(torch.float16)
            with torch.no_grad():
                from msod_amd.graph import CapturedForward
                model.overlap_streams = not fly_single and not args.no_overlap
                caps16 = [CapturedForward(model, args.batch, args.size, args.size) for _ in range(k_fly)]
                model.overlap_streams = not args.no_overlap
                for c in caps16:
                    c.rgb.copy_(rgb)
                    c.ir.copy_(ir)
                cap16 = caps16[0]
                if k_fly > 1:
                    run16 = [(lambda c=c: c.replay_static()[0]) for c in caps16]
                    fn16 = D.ForwardPipeline(run16, D.ForwardPipeline.pick_streams(run16, dev)[0]).step    # (new graphs: probe the stream groups again)
                else:
                    fn16 = lambda: cap16.replay_static()[0]   # noqa: E731
                torch.cuda.synchronize()
                el16 = D.timed_steps(fn16, args.steps, max(args.warmup, k_fly), sync=torch.cuda.synchronize)
            assert torch.isfinite(cap16.pred).all()
            got_raw16 = [r[:n_par].float().cpu() for r in cap16.raw] if n_par else None
            line["f16"] = {"value": round(args.batch * args.steps / el16, 2), "unit": "image-pairs/sec",
                           "ms_per_step": round(el16 / args.steps * 1e3, 3), "steps": args.steps,
                           "note": "same workload and step mode with compute dtype fp16 (the reference's own GPU precision, test.py:66-68)"}
            model.release_graphs()
            model.set_compute_dtype(dtype)
        want_raw = None
        sd_cpu = {k: v for k, v in model.cpu().state_dict().items()}      # fused (deployed) weights, fp32
        rgb_c, ir_c = rgb[:max(n_par, 1)].cpu(), ir[:max(n_par, 1)].cpu()
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is reported at N = 1 only
            nb = max(1, min(args.batch, 8))
            (_, want_raw), line["cpu_baseline"] = cpu_baseline(cfg, sd_cpu, rgb[:nb].cpu(), ir[:nb].cpu())
        ok = True
        if n_par:
            from oracle.cft_oracle import OracleModel
            from oracle.lowp_oracle import AutocastOracle
            threads = torch.get_num_threads()
            torch.set_num_threads(getattr(cpu_baseline, "best_threads", min(16, os.cpu_count() or 8)))
            try:
                if want_raw is None or want_raw[0].shape[0] < n_par:
                    _, want_raw = OracleModel(cfg)(sd_cpu, rgb_c, ir_c)
                want_raw = [r[:n_par] for r in want_raw]
                ref16, n_ref = None, 0
                if args.dtype == "bf16":
                    n_ref = min(2, n_par)
                    t0 = time.perf_counter()
                    _, ref16 = AutocastOracle(cfg)(sd_cpu, rgb_c[:n_ref], ir_c[:n_ref])
                    log(f"reference-style bf16 forward of {n_ref} pairs on the host: {time.perf_counter() - t0:.1f} s")
            finally:
                torch.set_num_threads(threads)
            line["parity_at_bench_shape"] = parity_at_bench_shape(args.dtype, got_raw, want_raw, ref16, n_ref)
            ok = line["parity_at_bench_shape"]["ok"]
            if "f16" in line and got_raw16 is not None:
                line["f16"]["parity_at_bench_shape"] = parity_at_bench_shape("f16", got_raw16, want_raw)
                ok = ok and line["f16"]["parity_at_bench_shape"]["ok"]
            # the 16-bit mode of THIS line that meets north_star's literal bound at the benchmarked shape (fp32 runs: the run itself)
            legs = {args.dtype: line["parity_at_bench_shape"]}
            if "f16" in line and "parity_at_bench_shape" in line["f16"]:
                legs["f16"] = line["f16"]["parity_at_bench_shape"]
            green = [d for d in ("f16", "bf16", "f32") if d in legs and legs[d]["ok"] and legs[d]["meets_north_star_bound"]]
            line["parity_green_dtype"] = green[0] if green else None
            if line["parity_green_dtype"] == "f16" and "f16" in line:
                line["parity_green_value"] = line["f16"]["value"]
            line["parity_weights"] = "utils/seeded.seeded_state_dict (deliberately lively stress weights: activations O(1) through the depth)"
            if world == 1:
                # (the oracle legs below at the thread count the cpu_baseline sweep found best: torch's default on a 256-thread host is 3-5 x slower)
                torch.set_num_threads(getattr(cpu_baseline, "best_threads", min(16, os.cpu_count() or 8)))
                # the contract's own weights: the timed dtype, literal bound; a miss fails the run like any parity miss
                line["parity_at_bench_shape_survey_weights"] = sp = survey_weights_parity(cfg, args, dev, dtype, rgb, ir)
                ok = ok and sp["ok"]
                log(f"survey-weights parity: {sp}")
                # the green claim of the timed dtype: the highest ladder rung on which the reference's own bf16 meets the literal bound and the
                # output demonstrably depends on the images - a miss fails the run like any parity miss
                line["parity_at_bench_shape_ladder"] = lp = ladder_weights_parity(cfg, args, dev, dtype, rgb, ir)
                ok = ok and lp["ok"]
                if lp["meets_north_star_bound"]:
                    line["parity_green_dtype_ladder"] = args.dtype
                log(f"ladder-rung parity: {lp}")
        print(json.dumps(line), flush=True)
        if not ok:
            log("PARITY FAILURE at the benchmarked shape - see parity_at_bench_shape in the line above")
            exit_code = 3
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    sys.exit(exit_code)


if __name__ == "__main__":
    main()
