"""Per-tile cycle anatomy of the persistent 128-channel Bottleneck kernel (variant 932 writes s_memtime-style stamps):
0 tile start | 1 after the W1 phases | 2 t patch written | 3 after K tile 2 of the 3x3 loop | 4 3x3 loop done |
5 hand-over drain done | 6 epilogue done | 7 past the closing barrier.  Prints median cycles per segment."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    C, B, H, W = 128, 64, 80, 80
    cat = ops.new_nhwc(B, H, W, 2 * C, torch.bfloat16, dev)
    cat.copy_(torch.randn(cat.shape, device=dev))
    x = cat[:, :C]
    pk1 = ops.pack_conv(torch.randn(C, C, 1, 1, generator=g) / C ** 0.5, torch.randn(C, generator=g) * 0.1, torch.bfloat16, device=dev)
    pk2 = ops.pack_conv(torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5), torch.randn(C, generator=g) * 0.1, torch.bfloat16, device=dev)
    out = ops.new_nhwc(B, H, W, C, torch.bfloat16, dev)
    dbg = torch.zeros((256, 2, 8, 8), dtype=torch.int64, device=dev)
    lib.cft_set_debug_buffer(dbg.data_ptr())
    lib.cft_set_conv_variant(932)
    for _ in range(3):
        ops.bottleneck(x, pk1, pk2, True, out=out)
    torch.cuda.synchronize()
    lib.cft_set_conv_variant(0)
    lib.cft_set_debug_buffer(None)
    d = dbg.cpu()
    names = ["W1 phases", "t write", "3x3 K tiles 0-2", "3x3 K tiles 3-17", "hand-over drain", "epilogue", "closing barrier"]
    res = {}
    for grp in (0, 1):
        for ti in (0, 1, 3, 5):
            t = d[:, grp, ti]                       # [256, 8]
            ok = t[:, 7] > 0
            seg = (t[ok, 1:] - t[ok, :-1]).float()
            med = seg.median(0)[0].tolist()
            tot = float((t[ok, 7] - t[ok, 0]).float().median())
            res[f"group{grp}_tile{ti}"] = {"total": tot, **{n: v for n, v in zip(names, med)}}
            print(f"group {grp} tile {ti}: total {tot:.0f} cycles | " + " | ".join(f"{n} {v:.0f}" for n, v in zip(names, med)))
    span = (d[:, 0, :, 7].max(1)[0] - d[:, 0, 0, 0]).float()
    print("kernel span per workgroup (cycles): median %.0f max %.0f" % (span.median(), span.max()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "bneck_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
