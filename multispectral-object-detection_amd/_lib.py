"""ctypes binding of ``libcft_hip.so`` (C ABI declared in ``include/cft_hip.h``).

There is deliberately NO fallback: if the library is missing or the device is not gfx950 every
op raises.  ``build()`` compiles the library in-tree with hipcc (cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess

# torch must be imported BEFORE the library is dlopen'ed: torch ships its own libamdhip64 and a
# process that first loads the system copy (through libcft_hip.so) and then torch's ends up with
# two HIP runtimes ("No HIP GPUs are available").  Loaded in this order both bind to torch's copy.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CFT_HIP_LIB") or os.path.join(_HERE, "libcft_hip.so")      # (CFT_HIP_LIB: experiments with an alternative build, e.g. the probe library)
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ("runtime.hip", "conv_gemm.hip", "conv_gemm_asm.hip", "focus_conv.hip", "bottleneck.hip", "pointwise.hip", "attention.hip", "nms.hip", "train.hip")

HEADERS = ("cft_common.h", "conv_common.h", "focus_common.h", "bneck_common.h", "conv_gemm_asm.inc")

CFT_BF16, CFT_F32, CFT_F16 = 0, 1, 2
ABI_VERSION = 11
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2

_c = ctypes
_vp, _i, _l, _f = _c.c_void_p, _c.c_int, _c.c_long, _c.c_float

# name -> argtypes; every function returns int except cft_last_error.  Mirrors include/cft_hip.h.
SIGNATURES = {
    "cft_abi_version": [],
    "cft_device_check": [],
    "cft_clock_probe": [_vp, _i, _c.POINTER(_c.c_int), _vp],
    "cft_conv2d_chain": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_conv2d_chain_ok": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "cft_conv2d_chain_res": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_linear_splitk": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_layernorm_reduce": [_vp, _vp, _i, _vp, _vp, _vp, _l, _i, _f, _i, _vp],
    "cft_conv2d": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_set_conv_variant": [_i],
    "cft_bottleneck": [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_bottleneck_pack_w2": [_vp, _i, _i, _vp, _i, _vp],
    "cft_focus_s2d": [_vp, _vp, _i, _i, _i, _i, _vp],
    "cft_focus_s2d_u8": [_vp, _l, _l, _l, _vp, _i, _i, _i, _f, _i, _vp],
    "cft_focus_conv": [_vp, _i, _l, _l, _l, _f, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_letterbox_u8": [_vp, _i, _i, _l, _vp, _i, _i, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_to_nhwc": [_vp, _i, _l, _l, _l, _l, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_spp_maxpool": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_copy_channels": [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_add": [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _l, _i, _i, _vp],
    "cft_gpt_tokenize": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "cft_layernorm": [_vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp],
    "cft_attention": [_vp, _vp, _i, _i, _i, _i, _i, _f, _c.c_ulonglong, _vp],
    "cft_batchnorm_train": [_vp, _i, _i, _l, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _l, _vp],
    "cft_dropout": [_vp, _l, _f, _c.c_ulonglong, _i, _vp],
    "cft_batchnorm_train_workspace": [_l, _i],      # returns long (bytes)
    "cft_gpt_upsample_add": [_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_gpt_upsample_add2": [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "cft_nms": [_vp, _i, _i, _i, _f, _f, _i, _i, _vp, _i, _i, _vp, _l, _vp, _vp, _vp],
    "cft_detect_decode": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _l, _l, _vp],
}

_lib = None


def build(verbose=False):
    """Compile every HIP source for gfx950 into ``libcft_hip.so`` next to this file."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(_HERE, "..", "include", "cft_hip.h")]
    newest = max(os.path.getmtime(p) for p in srcs + hdrs)
    if os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    # one object per source, compiled concurrently (only the stale ones), then one link
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    obj_dir = os.path.join(_HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time):
            continue
        cmd = ["hipcc"] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    global _lib
    _lib = None
    return LIB_PATH


def load():
    """Return the loaded library (cached).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP kernels are the only execution path of this package. "
            "Build them with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch, let it propagate
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.cft_last_error.argtypes = []
    lib.cft_last_error.restype = ctypes.c_char_p
    lib.cft_batchnorm_train_workspace.restype = ctypes.c_long
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().cft_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (status {status}): {msg}")
