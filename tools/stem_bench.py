"""Per-launch time of the one-kernel stem (cft_stem) against cft_focus_conv + cft_conv2d_chain at the bench shape, and of its timing probes
(cft_set_conv_variant(8800 + bits): results wrong, time only).  Since round 6 the stem lives in the PROBE build only
(csrc/probes/stem.hip; not in the product ABI), so this script binds cft_stem itself:
    bash tools/build_probes.sh && CFT_HIP_LIB=$PWD/multispectral-object-detection_amd/libcft_hip_probes.so python tools/stem_bench.py"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd import _lib, ops  # noqa: E402


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def stem(lib, img, pkf, pk1, pk2, act2, dtype):
    """act2(conv1x1(SiLU(conv3x3s2(SiLU(focus_conv(img)))))) as the one cft_stem kernel of the probe build (fp32 image, yolov5l widths)."""
    B, _, H, W = img.shape
    Ho, Wo = (H // 2 - 1) // 2 + 1, (W // 2 - 1) // 2 + 1
    out = ops.new_nhwc(B, Ho, Wo, pk2.n, dtype, img.device)
    code = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}[dtype]
    st = lib.cft_stem(img.data_ptr(), 0, img.stride(0), img.stride(1), img.stride(2), 1.0,
                      pkf.w.data_ptr(), pkf.kpad, pkf.bias.data_ptr(), pk1.w.data_ptr(), pk1.kpad, pk1.bias.data_ptr(),
                      pk2.w.data_ptr(), pk2.bias.data_ptr(), out.data_ptr(), out.stride(3), 0, B, H, W, pkf.n, pk1.n, pk2.n, act2, code,
                      torch.cuda.current_stream().cuda_stream)
    _lib.check(st, "cft_stem")
    return out


def main():
    if "probes" not in os.path.basename(_lib.LIB_PATH):
        sys.exit("stem_bench.py needs the probe build: CFT_HIP_LIB=.../libcft_hip_probes.so (tools/build_probes.sh)")
    dev, dt = torch.device("cuda:0"), torch.bfloat16
    B, S = 64, 640
    g = torch.Generator().manual_seed(0)
    img = torch.rand((B, 3, S, S), generator=g).to(dev)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    pkf = ops.pack_conv(rnd(64, 12, 3, 3) * (2.0 / 108) ** 0.5, rnd(64) * 0.1, dt, cin_pad=16, device=dev)
    pk1 = ops.pack_conv(rnd(128, 64, 3, 3) * (2.0 / 576) ** 0.5, rnd(128) * 0.1, dt, s=2, device=dev)
    pk2 = ops.pack_conv(rnd(128, 128, 1, 1) * (2.0 / 128) ** 0.5, rnd(128) * 0.1, dt, device=dev)
    lib = _lib.load()
    vp, i, l, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
    lib.cft_stem.restype = i
    lib.cft_stem.argtypes = [vp, i, l, l, l, f, vp, i, vp, vp, i, vp, vp, vp, vp] + [i] * 10 + [vp]
    out = {}
    out["focus_conv"] = timed(lambda: ops.focus_conv(img, pkf, ops.ACT_SILU, dt))
    f = ops.focus_conv(img, pkf, ops.ACT_SILU, dt)
    out["conv2d_chain"] = timed(lambda: ops.conv2d_chain(f, pk1, pk2, ops.ACT_SILU))
    names = {0: "stem 8x8 (shipped form)", 8801: "8x8: no weight ring", 8802: "8x8: no Focus phase",
             8804: "8x8: no conv-tap MFMAs", 8808: "8x8: no epilogue", 8816: "8x8: no image samples", 8803: "8x8: no ring, no Focus",
             8807: "8x8: no ring, no Focus, no taps", 8831: "8x8: everything off (barriers, patch build, images, pointwise GEMM)"}
    for v, name in names.items():
        old = lib.cft_set_conv_variant(v)
        try:
            out[name] = timed(lambda: stem(lib, img, pkf, pk1, pk2, ops.ACT_SILU, dt))
        finally:
            lib.cft_set_conv_variant(old)
    for k, v in out.items():
        print(f"{k:70s} {v:9.1f} us")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stem_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
