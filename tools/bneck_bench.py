"""Time cft_bottleneck against the two cft_conv2d launches it replaces, and (probe build only: tools/build_probes.sh) its ablation probes.
    python tools/bneck_bench.py [C] [variants]      C = 64 (160x160 stage) or 128 (80x80 stage), batch 64
variant 0 = the shipped kernel.  Probe build, 64 channels: 9601 / 9602 / 9604 / 9608 = without W1-stage MFMAs / 3x3 MFMAs / epilogue /
weight DMA; 128 channels: 9201 / 9202 / 9204 / 9208 = the same, 9456 = no x requests, 9712 = no SiLU."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=20, repeats=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    dev = torch.device("cuda:0")
    if os.environ.get("CFT_BENCH_LIB"):          # e.g. the probe build (tools/build_probes.sh): libcft_hip_probes.so
        _lib.LIB_PATH = os.path.abspath(os.environ["CFT_BENCH_LIB"])
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    B, H, W = (64, 160, 160) if C == 64 else (64, 80, 80)
    cat = ops.new_nhwc(B, H, W, 2 * C, torch.bfloat16, dev)
    cat.copy_(torch.randn(cat.shape, device=dev))
    x = cat[:, :C]
    pk1 = ops.pack_conv(torch.randn(C, C, 1, 1, generator=g) / C ** 0.5, torch.randn(C, generator=g) * 0.1, torch.bfloat16, device=dev)
    pk2 = ops.pack_conv(torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5), torch.randn(C, generator=g) * 0.1, torch.bfloat16, device=dev)
    out = ops.new_nhwc(B, H, W, C, torch.bfloat16, dev)
    t = ops.new_nhwc(B, H, W, C, torch.bfloat16, dev)
    only = len(sys.argv) > 2 and [int(v) for v in sys.argv[2].split(",")]
    flops = 2.0 * B * H * W * C * C * 10
    res = {"C": C, "gflop": flops / 1e9}
    if not only:
        us = timeit(lambda: ops.conv2d(ops.conv2d(x, pk1, 1, out=t), pk2, 1, residual=x, out=out))
        res["two_launches_us"] = round(us, 1)
        print(f"two launches: {us:.1f} us  ({flops / us / 1e6:.0f} TFLOP/s)")
    probes = "probes" in os.path.basename(getattr(_lib, "LIB_PATH", ""))
    for v in (only or ((0, 9601, 9602, 9604, 9608) if C == 64 else ((0, 98, 0, 98, 9301, 9302, 9304, 9308, 9303, 9306, 9305) if probes else (0,)))):
        lib.cft_set_conv_variant(v)
        us = timeit(lambda: ops.bottleneck(x, pk1, pk2, True, out=out))
        res[f"fused_v{v}_us"] = round(us, 1)
        print(f"fused variant {v}: {us:.1f} us  ({flops / us / 1e6:.0f} TFLOP/s)")
    lib.cft_set_conv_variant(0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"bneck_bench_c{C}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
