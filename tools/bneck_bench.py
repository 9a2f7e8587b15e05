"""Time cft_bottleneck (64-channel stage, 160x160, batch 64) against the two cft_conv2d launches it replaces, and
its ablation probes (variants 901 = no phase 1, 902 = no phase-2 MFMAs, 904 = no epilogue, 907 = none of them)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    B, H, W = 64, 160, 160
    cat = ops.new_nhwc(B, H, W, 128, torch.bfloat16, dev)
    cat.copy_(torch.randn(cat.shape, device=dev))
    x = cat[:, :64]
    pk1 = ops.pack_conv(torch.randn(64, 64, 1, 1, generator=g) / 8, torch.randn(64, generator=g) * 0.1, torch.bfloat16, device=dev)
    pk2 = ops.pack_conv(torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g) * 0.1, torch.bfloat16, device=dev)
    out = ops.new_nhwc(B, H, W, 64, torch.bfloat16, dev)
    t = ops.new_nhwc(B, H, W, 64, torch.bfloat16, dev)
    only = sys.argv[1:] and [int(v) for v in sys.argv[1].split(",")]
    if not only:
        us = timeit(lambda: ops.conv2d(ops.conv2d(x, pk1, 1, out=t), pk2, 1, residual=x, out=out))
        print(f"two launches: {us:.1f} us")
    for v in (only or (0, 901, 902, 904, 907)):
        lib.cft_set_conv_variant(v)
        us = timeit(lambda: ops.bottleneck(x, pk1, pk2, True, out=out))
        print(f"fused variant {v}: {us:.1f} us")
    lib.cft_set_conv_variant(0)


if __name__ == "__main__":
    main()
