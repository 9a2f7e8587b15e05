"""Per-pair timing of cft_conv2d_chain against the two cft_conv2d launches it replaces (stride-2 Conv 64 -> 128 / 128 -> 256 + the C3's
packed cv1|cv2 at the bench shape), interleaved and warmed; HIP events on the launch stream.
  python tools/chain_bench.py [--batch 64] [--size 640]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import msod_amd  # noqa: E402,F401
from msod_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--res", action="store_true", help="only the Bottleneck pair WITH a shortcut (cft_conv2d_chain_res): 3x3 256 -> 256 + shortcut, then 1x1 256 -> 256 @ size / 16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    if os.environ.get("CFT_CHAIN_BENCH_VARIANT"):          # e.g. 97 = the round-5 kernels (16-wave chained kernel, 16-wave plain GEMMs)
        from msod_amd import _lib
        _lib.load().cft_set_conv_variant(int(os.environ["CFT_CHAIN_BENCH_VARIANT"]))
    if args.res:
        return res_pair(args, dev, g)
    # (input channels, first layer's width, input size = image size / div, stride): the two backbone pairs, and cv2 of Bottleneck j +
    # cv1 of Bottleneck j + 1 in the head's 256-channel C3s
    for dtype, (cin, n1, div, st) in ((d, c) for d in (torch.bfloat16, torch.float16) for c in ((64, 128, 2, 2), (128, 256, 4, 2), (256, 256, 16, 1))):
        H = args.size // div
        x = ops.new_nhwc(args.batch, H, H, cin, dtype, dev)
        x.copy_(torch.randn(args.batch, cin, H, H, generator=g).to(dtype))
        pk1 = ops.pack_conv(torch.randn(n1, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5, torch.randn(n1, generator=g) * 0.1, dtype, s=st, device=dev)
        pk2 = ops.pack_conv(torch.randn(n1, n1, 1, 1, generator=g) * (2.0 / n1) ** 0.5, torch.randn(n1, generator=g) * 0.1, dtype, device=dev)
        mid = ops.conv2d(x, pk1, ops.ACT_SILU)
        out2 = ops.conv2d(mid, pk2, ops.ACT_SILU)
        out1 = ops.conv2d_chain(x, pk1, pk2, ops.ACT_SILU)
        torch.cuda.synchronize()
        assert torch.equal(out1, out2)

        def two():
            ops.conv2d(x, pk1, ops.ACT_SILU, out=mid)
            ops.conv2d(mid, pk2, ops.ACT_SILU, out=out2)

        def first():
            ops.conv2d(x, pk1, ops.ACT_SILU, out=mid)

        def one():
            ops.conv2d_chain(x, pk1, pk2, ops.ACT_SILU, out=out1)

        best = {}
        for rnd in range(4):
            for name, fn in (("two launches", two), ("first layer alone", first), ("chained", one)):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.iters
                if rnd > 0:
                    best[name] = min(best.get(name, 1e9), us)
        M = args.batch * (H // st) ** 2
        fl = 2.0 * M * (n1 * 9 * cin + n1 * n1)
        print(f"{str(dtype):16s} {args.batch} x {cin}ch {H}x{H} -> {n1} -> {n1} @ {H // st}:  " +
              "   ".join(f"{k} {v:7.1f} us" for k, v in best.items()) +
              f"   chained = {best['chained'] / best['two launches']:.3f} x two launches, {fl / best['chained'] / 1e6:.0f} TFLOP/s")


def res_pair(args, dev, g):
    """cv2[j] (3x3, 256 channels) + shortcut and cv1[j + 1] (1x1) of a C3 with shortcuts: one cft_conv2d_chain_res launch against the two
    cft_conv2d launches (bit-identical), at the bench shape (64 x 40 x 40)."""
    for dtype in (torch.bfloat16, torch.float16):
        H = args.size // 16
        mk = lambda c: ops.new_nhwc(args.batch, H, H, c, dtype, dev)      # noqa: E731
        x, res = mk(256), mk(256)
        x.copy_(torch.randn(args.batch, 256, H, H, generator=g).to(dtype))
        res.copy_(torch.randn(args.batch, 256, H, H, generator=g).to(dtype))
        pk1 = ops.pack_conv(torch.randn(256, 256, 3, 3, generator=g) * (2.0 / 2304) ** 0.5, torch.randn(256, generator=g) * 0.1, dtype, device=dev)
        pk2 = ops.pack_conv(torch.randn(256, 256, 1, 1, generator=g) * (2.0 / 256) ** 0.5, torch.randn(256, generator=g) * 0.1, dtype, device=dev)
        y1, y2, z1, z2 = mk(256), mk(256), mk(256), mk(256)
        ops.conv2d(x, pk1, ops.ACT_SILU, residual=res, out=y1)
        ops.conv2d(y1, pk2, ops.ACT_SILU, out=y2)
        ops.conv2d_chain_res(x, pk1, res, pk2, ops.ACT_SILU, out1=z1, out2=z2)
        torch.cuda.synchronize()
        assert torch.equal(y1, z1) and torch.equal(y2, z2)

        def two():
            ops.conv2d(x, pk1, ops.ACT_SILU, residual=res, out=y1)
            ops.conv2d(y1, pk2, ops.ACT_SILU, out=y2)

        def first():
            ops.conv2d(x, pk1, ops.ACT_SILU, residual=res, out=y1)

        def one():
            ops.conv2d_chain_res(x, pk1, res, pk2, ops.ACT_SILU, out1=z1, out2=z2)

        best = {}
        for rnd in range(4):
            for name, fn in (("two launches", two), ("3x3 + shortcut alone", first), ("chained", one)):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.iters
                if rnd > 0:
                    best[name] = min(best.get(name, 1e9), us)
        M = args.batch * H * H
        fl = 2.0 * M * (256 * 2304 + 256 * 256)
        print(f"{str(dtype):16s} {args.batch} x 256ch {H}x{H}, 3x3 + shortcut -> 1x1:  " + "   ".join(f"{k} {v:7.1f} us" for k, v in best.items()) +
              f"   chained = {best['chained'] / best['two launches']:.3f} x two launches, {fl / best['chained'] / 1e6:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
