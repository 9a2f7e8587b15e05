"""Micro-benchmark of cft_conv2d tile variants on the layer shapes that dominate
yolov5l+CFTx3 @ 640x640, batch 64 (run on the GPU box; writes gpurun_out/gemm_bench.json).

    python tools/gemm_bench.py [--variants 0,1,2,5] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd import _lib, ops  # noqa: E402

# name, B, H, W, Cin, N, k, s, residual, count per forward (both streams + head)
SHAPES = [
    ("focus 3x3 16->64 @320", 64, 320, 320, 16, 64, 3, 1, False, 2),
    ("3x3s2 64->128 @320->160", 64, 320, 320, 64, 128, 3, 2, False, 2),
    ("C3 1x1 128->128 @160 (cv1|cv2, cv3)", 64, 160, 160, 128, 128, 1, 1, False, 4),
    ("bneck 1x1 64->64 @160", 64, 160, 160, 64, 64, 1, 1, False, 6),
    ("bneck 3x3 64->64 @160 +res", 64, 160, 160, 64, 64, 3, 1, True, 6),
    ("3x3s2 128->256 @160->80", 64, 160, 160, 128, 256, 3, 2, False, 2),
    ("C3 1x1 256->256 @80", 64, 80, 80, 256, 256, 1, 1, False, 6),
    ("bneck 1x1 128->128 @80", 64, 80, 80, 128, 128, 1, 1, False, 21),
    ("bneck 3x3 128->128 @80 +res", 64, 80, 80, 128, 128, 3, 1, True, 21),
    ("3x3s2 256->512 @80->40", 64, 80, 80, 256, 512, 3, 2, False, 3),
    ("C3 1x1 512->512 @40", 64, 40, 40, 512, 512, 1, 1, False, 8),
    ("bneck 1x1 256->256 @40", 64, 40, 40, 256, 256, 1, 1, False, 24),
    ("bneck 3x3 256->256 @40 +res", 64, 40, 40, 256, 256, 3, 1, True, 24),
    ("3x3s2 512->1024 @40->20", 64, 40, 40, 512, 1024, 3, 2, False, 3),
    ("bneck 3x3 512->512 @20", 64, 20, 20, 512, 512, 3, 1, False, 9),
    ("1x1 1024->1024 @20", 64, 20, 20, 1024, 1024, 1, 1, False, 6),
    ("SPP cv2 1x1 2048->1024 @20", 64, 20, 20, 2048, 1024, 1, 1, False, 2),
    # tile-count quantisation probe: the same layer at M = 65 536 rows (256 tiles of 256 = exactly one round of 256 CUs)
    ("quant 3x3 256->256 @32 (M=65536) +res", 64, 32, 32, 256, 256, 3, 1, True, 0),
    ("quant 1x1 256->256 @32 (M=65536)", 64, 32, 32, 256, 256, 1, 1, False, 0),
    ("x5 3x3 160->160 @160 +res", 16, 160, 160, 160, 160, 3, 1, True, 0),
    ("x5 3x3 320->320 @80 +res", 16, 80, 80, 320, 320, 3, 1, True, 0),
    ("x5 3x3 80->80 @320 +res", 16, 320, 320, 80, 80, 3, 1, True, 0),
    ("x5 3x3 640->640 @40", 16, 40, 40, 640, 640, 3, 1, False, 0),
    ("x5 1x1 160->160 @160", 16, 160, 160, 160, 160, 1, 1, False, 0),
    ("x5 1x1 320->320 @80", 16, 80, 80, 320, 320, 1, 1, False, 0),
    ("GPT qkv 1024->3072 (M=8192)", 1, 1, 8192, 1024, 3072, 1, 1, False, 8),
    ("GPT fc1 1024->4096", 1, 1, 8192, 1024, 4096, 1, 1, False, 8),
    ("GPT fc2 4096->1024", 1, 1, 8192, 4096, 1024, 1, 1, False, 8),
    ("GPT out 512->512", 1, 1, 8192, 512, 512, 1, 1, False, 8),
    ("GPT fc1 256->1024", 1, 1, 8192, 256, 1024, 1, 1, False, 8),
    ("GPT out 1024->1024", 1, 1, 8192, 1024, 1024, 1, 1, False, 8),
    ("GPT qkv 512->1536", 1, 1, 8192, 512, 1536, 1, 1, False, 8),
    ("GPT fc1 512->2048", 1, 1, 8192, 512, 2048, 1, 1, False, 8),
    ("GPT fc2 2048->512", 1, 1, 8192, 2048, 512, 1, 1, False, 8),
    ("GPT qkv 256->768", 1, 1, 8192, 256, 768, 1, 1, False, 8),
    ("GPT out 256->256", 1, 1, 8192, 256, 256, 1, 1, False, 8),
    ("GPT fc2 1024->256", 1, 1, 8192, 1024, 256, 1, 1, False, 8),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="1,2,5,9")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3, help="timed passes per variant, interleaved over the variants (A B C A B C ...): the median is reported")
    ap.add_argument("--warm", type=int, default=10, help="untimed launches before every timed pass (a pass that follows idle time runs at boost clocks: "
                    "the first variant of a back-to-back pair measured up to 10 %% faster on MFMA-bound shapes, profiles/r03_gemm_experiments.md 5d)")
    ap.add_argument("--only", default="")
    ap.add_argument("--used", action="store_true", help="only the shapes that occur in the cfg3 forward (count > 0)")
    ap.add_argument("--out", default="gemm_bench.json")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--lib", default="", help="load this library instead of the product one (tools/build_probes.sh: libcft_hip_probes.so with the timing probes)")
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    dev = torch.device("cuda:0")
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    lib = _lib.load()
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    results = []
    g = torch.Generator().manual_seed(0)
    for name, B, H, W, Cin, N, k, s, use_res, count in SHAPES:
        if args.only and not any(o in name for o in args.only.split('|')):
            continue
        if args.used and count == 0:
            continue
        x = ops.new_nhwc(B, H, W, Cin, dtype, dev)
        x.copy_(torch.randn(x.shape, device=dev) )
        w = torch.randn((N, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
        pk = ops.pack_conv(w, torch.randn(N, generator=g) * 0.1, dtype, s=s, device=dev)
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        res = None
        if use_res:
            res = ops.new_nhwc(B, Ho, Wo, N, dtype, dev)
            res.copy_(torch.randn(res.shape, device=dev))
        flops = 2.0 * B * Ho * Wo * N * k * k * Cin
        byts = 2.0 * (B * H * W * Cin + B * Ho * Wo * N * (2 if use_res else 1) + N * k * k * Cin)
        rec = {"shape": name, "gflop": flops / 1e9, "mbytes": byts / 1e6, "count": count, "variants": {}}
        ref = None
        outs, times = {}, {v: [] for v in variants}
        for v in variants:                       # correctness pass (and first-touch) per variant
            lib.cft_set_conv_variant(v)
            try:
                out = ops.conv2d(x, pk, 1, residual=res)
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.float()
                    diff = 0.0
                else:
                    diff = (out.float() - ref).abs().max().item()
                outs[v] = (out, diff)
            except Exception as e:  # noqa: BLE001
                rec["variants"][v] = {"error": repr(e)[:200]}
        for _ in range(max(1, args.rounds)):     # timed passes, interleaved over the variants, each behind its own warm-up
            for v in variants:
                if v not in outs:
                    continue
                lib.cft_set_conv_variant(v)
                out = outs[v][0]
                for _ in range(args.warm):
                    ops.conv2d(x, pk, 1, residual=res, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    ops.conv2d(x, pk, 1, residual=res, out=out)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3 / args.iters)
        for v in variants:
            if v in outs:
                ts = sorted(times[v])
                us = ts[len(ts) // 2]
                rec["variants"][v] = {"us": round(us, 1), "tflops": round(flops / us / 1e6, 1), "tbps": round(byts / us / 1e6, 2),
                                      "maxdiff_vs_first": outs[v][1], "passes_us": [round(t, 1) for t in times[v]]}
        del outs
        lib.cft_set_conv_variant(0)
        print(json.dumps(rec), flush=True)
        results.append(rec)
        del x, res
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", args.out), "w") as fh:
        json.dump(results, fh, indent=1)
    # summary: time per forward per variant
    for v in variants:
        tot = sum(r["variants"][v]["us"] * r["count"] for r in results if "us" in r["variants"].get(v, {}))
        print(f"variant {v}: sum(us*count) = {tot / 1e3:.2f} ms over listed shapes")
    best = sum(min(x["us"] for x in r["variants"].values() if "us" in x) * r["count"] for r in results)
    print(f"best-of per shape: {best / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
