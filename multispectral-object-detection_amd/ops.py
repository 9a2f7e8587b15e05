"""Thin tensor-level wrappers over the C ABI of libcft_hip.so.

PyTorch is used here only for device memory (``torch.empty``), the current HIP stream and
one-time weight packing; every arithmetic op of the forward is a HIP kernel.  Activations are
torch tensors of logical shape [B,C,H,W] whose memory is NHWC ("channels-last"), possibly a
channel slice of a wider buffer (stride(3) = channels per pixel of the parent).

No CPU path exists: CPU tensors raise.
"""
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_SILU, CFT_BF16, CFT_F16, CFT_F32  # noqa: F401


COMPUTE_DTYPES = (torch.bfloat16, torch.float16, torch.float32)


def _dt(dtype):
    if dtype == torch.bfloat16:
        return CFT_BF16
    if dtype == torch.float16:
        return CFT_F16
    if dtype == torch.float32:
        return CFT_F32
    raise TypeError(f"compute dtype must be torch.bfloat16, torch.float16 or torch.float32, got {dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; this package runs on MI355X only "
                           "(no CPU fallback - use oracle/ for a CPU reference)")


def granule(dtype):
    return 4 if dtype == torch.float32 else 8


def new_nhwc(B, H, W, C, dtype, device):
    """Logical [B,C,H,W] tensor backed by a fresh NHWC buffer."""
    return torch.empty((B, H, W, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


def to_nhwc(x, dtype=None, cpad=None):
    """Any strided [B,C,H,W] fp32/bf16/half tensor -> fresh NHWC buffer in ``dtype`` with the channel count
    padded to ``cpad`` (default: next granule multiple; padding channels zero) - the cft_to_nhwc kernel."""
    _require_cuda(x, "to_nhwc")
    B, C, H, W = x.shape
    dtype = dtype or x.dtype
    ge = granule(dtype)
    cp = cpad or ((C + ge - 1) // ge) * ge
    y = new_nhwc(B, H, W, cp, dtype, x.device)
    st = _lib.load().cft_to_nhwc(x.data_ptr(), _dt(x.dtype), x.stride(0), x.stride(1), x.stride(2), x.stride(3),
                                 y.data_ptr(), cp, 0, B, C, cp, H, W, _dt(dtype), _stream())
    _lib.check(st, "cft_to_nhwc")
    return y


def as_nhwc(x):
    """Return (x', ld): x' has NHWC memory, ld = channels/pixel.  A tensor in another layout (e.g. a plain NCHW
    tensor handed to a module from outside the network) is converted by the cft_to_nhwc kernel; its channel
    count must already be a granule multiple (the consumer's packed weights fix the channel count)."""
    B, C, H, W = x.shape
    ld = x.stride(3)
    ok = x.stride(1) == 1 and ld >= C and x.stride(2) == W * ld and x.stride(0) == H * W * ld
    ge = granule(x.dtype)
    ok = ok and ld % ge == 0 and (x.storage_offset() % ge == 0)
    if not ok:
        if C % ge:
            raise ValueError(f"as_nhwc: {C} channels is not a multiple of the {ge}-element granule")
        return to_nhwc(x), C
    return x, ld


def _view_ld(x, what):
    """ld of an NHWC view that must NOT be copied (outputs / residuals)."""
    B, C, H, W = x.shape
    ld = x.stride(3)
    if not (x.stride(1) == 1 and ld >= C and x.stride(2) == W * ld and x.stride(0) == H * W * ld):
        raise ValueError(f"{what}: tensor is not an NHWC buffer or a channel slice of one")
    return ld


# ------------------------------------------------------------------------------ weight packing
@dataclass
class PackedConv:
    """Weights in the layout cft_conv2d consumes: w[n][kpad] with (kh,kw,ci) flattened, ci
    fastest, zero padded; bias fp32."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    n: int        # rows of w (output channels incl. zero padding to a multiple of 8)
    n_valid: int  # real output channels
    cin: int      # input channels the kernel expects (incl. zero padding)
    kpad: int
    k: int
    s: int
    flops_per_row: float = 0.0   # algorithmic 2*N*K of the unpadded layer (roofline accounting)
    w_stages: Optional[torch.Tensor] = None   # stage-major image of w for cft_bottleneck (128-channel 3x3), built on first use


def fold_bn(weight, bn_weight, bn_bias, running_mean, running_var, eps):
    """Conv+BN(eval) folding, W' = diag(g/sqrt(v+eps)) W, b' = beta - g*mu/sqrt(v+eps)
    (reference utils/torch_utils.py:181-201)."""
    scale = bn_weight.float() / torch.sqrt(running_var.float() + eps)
    return weight.float() * scale.view(-1, 1, 1, 1), bn_bias.float() - running_mean.float() * scale


def pack_conv(weight, bias, dtype, k=None, s=1, cin_pad=None, device=None, n_true=None):
    """weight [N,Cin,kh,kw] (or [N,K] for a Linear) fp32 -> PackedConv in ``dtype``.
    ``n_true``: number of non-padding rows when the caller already padded N (padded heads)."""
    if weight.dim() == 2:
        weight = weight.view(weight.shape[0], weight.shape[1], 1, 1)
    N, Cin, kh, kw = weight.shape
    assert kh == kw, "square kernels only"
    device = device or weight.device
    ge = granule(dtype)
    cin_p = cin_pad or ((Cin + ge - 1) // ge) * ge
    n_p = ((N + 7) // 8) * 8
    bk = 8 * ge
    K = kh * kw * cin_p
    kpad = ((K + bk - 1) // bk) * bk
    w = torch.zeros((n_p, kh, kw, cin_p), dtype=torch.float32, device=device)
    w[:N, :, :, :Cin] = weight.detach().to(device=device, dtype=torch.float32).permute(0, 2, 3, 1)
    wp = torch.zeros((n_p, kpad), dtype=dtype, device=device)
    wp[:, :K] = w.view(n_p, K).to(dtype)
    b = None
    if bias is not None:
        b = torch.zeros((n_p,), dtype=torch.float32, device=device)
        b[:N] = bias.detach().to(device=device, dtype=torch.float32)
    return PackedConv(wp.contiguous(), b, n_p, N, cin_p, kpad, kh, s, 2.0 * (n_true or N) * kh * kw * Cin)


# ------------------------------------------------------------------------------ launch timing
# When a list is installed here every GEMM launch is bracketed by HIP events recorded on the launch
# stream (bench.py uses it to time the dominant kernel family live): entries are
# (tile_family, algorithmic_flops, start_event, end_event).
_launch_log = None


def set_launch_log(log):
    global _launch_log
    _launch_log = log


def _timed(name, flops, abytes, fn):
    """Run ``fn()`` (one kernel launch); when a launch log is installed bracket it with HIP events."""
    if _launch_log is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st = fn()
    e1.record()
    _launch_log.append((name, flops, e0, e1, abytes))
    return st


def _timed_gemm(lib, rows, pk, args, abytes=0.0, kind="conv"):
    if _launch_log is None:
        return lib.cft_conv2d(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st = lib.cft_conv2d(*args)
    e1.record()
    _launch_log.append((f"{kind}_k{pk.k}s{pk.s}_n{pk.n}_K{pk.kpad}", rows * pk.flops_per_row, e0, e1, abytes))
    return st


# ------------------------------------------------------------------------------ ops
def conv2d(x, pk, act, residual=None, out=None, out_dtype=None):
    """act(conv(x) + bias) (+ residual) as one implicit-GEMM kernel; see cft_conv2d."""
    _require_cuda(x, "conv2d")
    x, ldx = as_nhwc(x)
    B, C, H, W = x.shape
    if C != pk.cin:
        raise ValueError(f"conv2d: input has {C} channels, packed weight expects {pk.cin}")
    p = pk.k // 2
    Ho, Wo = (H + 2 * p - pk.k) // pk.s + 1, (W + 2 * p - pk.k) // pk.s + 1
    if out is None:
        out = new_nhwc(B, Ho, Wo, pk.n, out_dtype or x.dtype, x.device)
    if tuple(out.shape) != (B, pk.n, Ho, Wo):
        raise ValueError(f"conv2d: out has shape {tuple(out.shape)}, expected {(B, pk.n, Ho, Wo)}")
    ldy = _view_ld(out, "conv2d out")
    rp, ldr, rdt = None, 0, CFT_BF16
    if residual is not None:
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError("conv2d: residual shape mismatch")
        ldr = _view_ld(residual, "conv2d residual")
        rp, rdt = residual.data_ptr(), _dt(residual.dtype)
    lib = _lib.load()
    es_in, es_out = x.element_size(), out.element_size()
    abytes = (B * H * W * pk.cin * es_in + B * Ho * Wo * pk.n_valid * es_out + pk.w.numel() * es_in
              + (0 if residual is None else residual.numel() * residual.element_size()))
    st = _timed_gemm(lib, B * Ho * Wo, pk,
                     (x.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr() if pk.bias is not None else None, rp,
                      out.data_ptr(), B, H, W, pk.cin, ldx, 0, pk.n, pk.kpad, pk.k, pk.s,
                      ldy, 0, ldr, 0, act, _dt(x.dtype), _dt(out.dtype), rdt, _stream()), abytes)
    _lib.check(st, "cft_conv2d")
    return out


def conv2d_chain_ok_geometry(B, H, W, dtype, pk1, pk2, ldx=None, ldy=None):
    """``conv2d_chain_ok`` from the input geometry alone ([B, pk1.cin, H, W] of ``dtype`` on the GPU).  ``ldx`` / ``ldy``: channels per
    pixel of the buffers the input / output are slices of (default: dense) - the launcher's 2^31-element limits are on THOSE extents, so
    'ok' here means the launcher accepts."""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    if pk2.k != 1 or pk2.s != 1 or pk2.cin != pk1.n or pk2.kpad != pk1.n or pk1.n_valid != pk1.n:
        return False
    return bool(_lib.load().cft_conv2d_chain_ok(B, H, W, pk1.cin, ldx or pk1.cin, pk1.n, pk1.kpad, pk1.k, pk1.s, pk2.n, ldy or pk2.n, _dt(dtype)))


def conv2d_chain_ok(x, pk1, pk2):
    """True when ``conv2d_chain`` takes this pair: conv ``pk1`` (SiLU) then the pointwise conv ``pk2`` on its output (cft_conv2d_chain_ok)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.shape[1] == pk1.cin):
        return False
    ldx = x.stride(3) if (x.stride(1) == 1 and x.stride(3) >= x.shape[1]) else None      # (another layout is converted to dense NHWC first)
    return conv2d_chain_ok_geometry(x.shape[0], x.shape[2], x.shape[3], x.dtype, pk1, pk2, ldx=ldx)


def conv2d_chain(x, pk1, pk2, act2, out=None):
    """act2(conv1x1(SiLU(conv(x)))) as ONE kernel (cft_conv2d_chain): the first conv's output tile becomes the second GEMM's A operand
    in LDS.  Bit-identical to ``conv2d(conv2d(x, pk1, ACT_SILU), pk2, act2)``."""
    _require_cuda(x, "conv2d_chain")
    if not conv2d_chain_ok(x, pk1, pk2):
        raise ValueError("conv2d_chain: layer pair not eligible (conv2d_chain_ok)")
    x, ldx = as_nhwc(x)
    B, C, H, W = x.shape
    p = pk1.k // 2
    Ho, Wo = (H + 2 * p - pk1.k) // pk1.s + 1, (W + 2 * p - pk1.k) // pk1.s + 1
    if out is None:
        out = new_nhwc(B, Ho, Wo, pk2.n, x.dtype, x.device)
    if tuple(out.shape) != (B, pk2.n, Ho, Wo) or out.dtype != x.dtype:
        raise ValueError(f"conv2d_chain: out has shape {tuple(out.shape)}, expected {(B, pk2.n, Ho, Wo)}")
    ldy = _view_ld(out, "conv2d_chain out")
    lib = _lib.load()
    es = x.element_size()
    M = B * Ho * Wo
    abytes = B * H * W * pk1.cin * es + M * pk2.n_valid * es + (pk1.w.numel() + pk2.w.numel()) * es
    args = (x.data_ptr(), pk1.w.data_ptr(), pk1.bias.data_ptr() if pk1.bias is not None else None,
            pk2.w.data_ptr(), pk2.bias.data_ptr() if pk2.bias is not None else None, out.data_ptr(),
            B, H, W, pk1.cin, ldx, 0, pk1.n, pk1.kpad, pk1.k, pk1.s, pk2.n, ldy, 0, act2, _dt(x.dtype), _stream())
    # counted with the GEMM family: both layers' FLOPs, one launch
    st = _timed(f"conv_chain_k{pk1.k}s{pk1.s}_n{pk1.n}_K{pk1.kpad}+{pk2.kpad}", M * (pk1.flops_per_row + pk2.flops_per_row), abytes,
                lambda: lib.cft_conv2d_chain(*args))
    _lib.check(st, "cft_conv2d_chain")
    return out


CHAIN_RES_MIN_ROWS = 192 * 256      # output pixels from which the chained kernel's 256-row tiles fill >= 3/4 of the 256 CUs


def conv2d_chain_res_ok(x, pk1, pk2, any_size=False):
    """True when ``conv2d_chain_res`` takes the pair: ``conv2d_chain_ok``, a 256-channel first layer (the four-image form of the kernel) and -
    unless ``any_size`` - enough output pixels: the chained kernel exists in the 256 x 256 tile only, and below ~49 000 pixels (cfg3: < 32 pairs
    at 640 x 640) the separate launches run on smaller tiles with 4 - 8 x the workgroups (measured at 8 pairs: 71 against ~33 us per pair of
    launches, profiles/r05_splitk_ab.md)."""
    if not (pk1.n == 256 and conv2d_chain_ok(x, pk1, pk2)):
        return False
    p_ = pk1.k // 2
    rows = x.shape[0] * ((x.shape[2] + 2 * p_ - pk1.k) // pk1.s + 1) * ((x.shape[3] + 2 * p_ - pk1.k) // pk1.s + 1)
    return any_size or rows >= CHAIN_RES_MIN_ROWS


def conv2d_chain_res_ok_geometry(B, H, W, dtype, pk1, pk2, any_size=False):
    """``conv2d_chain_res_ok`` for DENSE input / output tensors of the given pixel grid (what ``C3.forward`` passes: the hidden tensor and y1 are
    dense; the shortcut's own row stride is checked at launch)."""
    if not (pk1.n == 256 and conv2d_chain_ok_geometry(B, H, W, dtype, pk1, pk2)):
        return False
    p_ = pk1.k // 2
    rows = B * ((H + 2 * p_ - pk1.k) // pk1.s + 1) * ((W + 2 * p_ - pk1.k) // pk1.s + 1)
    return any_size or rows >= CHAIN_RES_MIN_ROWS


def conv2d_chain_res(x, pk1, res, pk2, act2, out1=None, out2=None):
    """(y1, y2) = (SiLU(conv(x)) + res, act2(conv1x1(y1))) as ONE kernel (cft_conv2d_chain_res): Bottleneck j's 3x3 conv with its shortcut
    and Bottleneck j+1's 1x1 conv inside a C3 with shortcuts.  Bit-identical to ``conv2d(x, pk1, SILU, residual=res)`` followed by
    ``conv2d(y1, pk2, act2)``; y1 is stored but never re-read, and the 1x1 launch disappears."""
    _require_cuda(x, "conv2d_chain_res")
    if not conv2d_chain_res_ok(x, pk1, pk2, any_size=True):
        raise ValueError("conv2d_chain_res: layer pair not eligible (conv2d_chain_res_ok)")
    x, ldx = as_nhwc(x)
    B, C, H, W = x.shape
    p = pk1.k // 2
    Ho, Wo = (H + 2 * p - pk1.k) // pk1.s + 1, (W + 2 * p - pk1.k) // pk1.s + 1
    if tuple(res.shape) != (B, pk1.n, Ho, Wo) or res.dtype != x.dtype:
        raise ValueError(f"conv2d_chain_res: residual has shape {tuple(res.shape)}, expected {(B, pk1.n, Ho, Wo)}")
    ldr = _view_ld(res, "conv2d_chain_res residual")
    if out1 is None:
        out1 = new_nhwc(B, Ho, Wo, pk1.n, x.dtype, x.device)
    if out2 is None:
        out2 = new_nhwc(B, Ho, Wo, pk2.n, x.dtype, x.device)
    if tuple(out1.shape) != (B, pk1.n, Ho, Wo) or tuple(out2.shape) != (B, pk2.n, Ho, Wo) or out1.dtype != x.dtype or out2.dtype != x.dtype:
        raise ValueError("conv2d_chain_res: output shape / dtype mismatch")
    ldy1, ldy2 = _view_ld(out1, "conv2d_chain_res out1"), _view_ld(out2, "conv2d_chain_res out2")
    lib = _lib.load()
    es = x.element_size()
    M = B * Ho * Wo
    abytes = (B * H * W * pk1.cin + 2 * M * pk1.n_valid + M * pk2.n_valid + pk1.w.numel() + pk2.w.numel()) * es
    args = (x.data_ptr(), pk1.w.data_ptr(), pk1.bias.data_ptr() if pk1.bias is not None else None, res.data_ptr(), out1.data_ptr(),
            pk2.w.data_ptr(), pk2.bias.data_ptr() if pk2.bias is not None else None, out2.data_ptr(),
            B, H, W, pk1.cin, ldx, 0, pk1.n, pk1.kpad, pk1.k, pk1.s, ldr, 0, ldy1, 0, pk2.n, ldy2, 0, act2, _dt(x.dtype), _stream())
    st = _timed(f"conv_chainres_k{pk1.k}s{pk1.s}_n{pk1.n}_K{pk1.kpad}+{pk2.kpad}", M * (pk1.flops_per_row + pk2.flops_per_row), abytes,
                lambda: lib.cft_conv2d_chain_res(*args))
    _lib.check(st, "cft_conv2d_chain_res")
    return out1, out2


# cft_bottleneck covers 64 and 128 channels (activation patch resident in LDS, weights streamed through a 4-slot LDS ring, two
# workgroups per CU: 147-173 vs 220-234 us for the two launches it replaces at 128 channels - profiles/r02_bottleneck128.md).
FUSED_BOTTLENECK_WIDTHS = (64, 128)


def bottleneck_fusable(x, pk1, pk2, act1, act2):
    """True when ``bottleneck`` below can run as the single cft_bottleneck kernel."""
    return bottleneck_kernel_covers(x, pk1, pk2, act1, act2) and x.shape[1] in FUSED_BOTTLENECK_WIDTHS


def bottleneck_kernel_covers(x, pk1, pk2, act1, act2):
    """True when cft_bottleneck implements this Bottleneck at all (64 or 128 channels, 16-bit, SiLU)."""
    return (x.dtype in (torch.bfloat16, torch.float16) and act1 == ACT_SILU and act2 == ACT_SILU
            and pk1.k == 1 and pk1.s == 1 and pk2.k == 3 and pk2.s == 1
            and pk1.cin == pk1.n == pk2.cin == pk2.n == x.shape[1] and x.shape[1] in (64, 128))


def bottleneck(x, pk1, pk2, shortcut, out=None):
    """x (+) SiLU(conv3x3(SiLU(conv1x1(x)))) for 64 / 128 channels in one kernel (cft_bottleneck); ``out`` must not
    overlap ``x`` (a disjoint channel slice of the same buffer is fine)."""
    _require_cuda(x, "bottleneck")
    x, ldx = as_nhwc(x)
    B, C, H, W = x.shape
    if out is None:
        out = new_nhwc(B, H, W, C, x.dtype, x.device)
    if tuple(out.shape) != (B, C, H, W):
        raise ValueError(f"bottleneck: out has shape {tuple(out.shape)}, expected {(B, C, H, W)}")
    ldy = _view_ld(out, "bottleneck out")
    lib = _lib.load()
    if C == 128 and pk2.w_stages is None:   # one-time: the 3x3 weights as 36 contiguous 8-KiB stage images (cft_bottleneck_pack_w2)
        pk2.w_stages = torch.empty_like(pk2.w)
        _lib.check(lib.cft_bottleneck_pack_w2(pk2.w.data_ptr(), pk2.kpad, C, pk2.w_stages.data_ptr(), _dt(x.dtype), _stream()),
                   "cft_bottleneck_pack_w2")
    args = (x.data_ptr(), ldx, 0, pk1.w.data_ptr(), pk1.kpad, pk1.bias.data_ptr() if pk1.bias is not None else None,
            pk2.w.data_ptr(), pk2.kpad, pk2.w_stages.data_ptr() if pk2.w_stages is not None else None,
            pk2.bias.data_ptr() if pk2.bias is not None else None,
            out.data_ptr(), ldy, 0, B, H, W, C, 1 if shortcut else 0, _dt(x.dtype), _stream())
    if _launch_log is None:
        st = lib.cft_bottleneck(*args)
    else:   # counted with the GEMM family: both convolutions' FLOPs, one launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st = lib.cft_bottleneck(*args)
        e1.record()
        rows = B * H * W
        abytes = rows * C * 2 * (3 if shortcut else 2) + (pk1.w.numel() + pk2.w.numel()) * 2
        _launch_log.append((f"conv_bneck_c{C}_K{pk1.kpad}+{pk2.kpad}", rows * (pk1.flops_per_row + pk2.flops_per_row), e0, e1, abytes))
    _lib.check(st, "cft_bottleneck")
    return out


def linear(x, pk, act=ACT_NONE, residual=None, out=None, out_dtype=None):
    """x [rows, K] (row stride >= K) -> [rows, pk.n]; nn.Linear(+bias)(+act)(+residual)."""
    _require_cuda(x, "linear")
    rows, K = x.shape
    if K != pk.cin or x.stride(1) != 1:
        raise ValueError(f"linear: input [{rows},{K}] does not match packed weight (cin {pk.cin})")
    if out is None:
        out = torch.empty((rows, pk.n), dtype=out_dtype or x.dtype, device=x.device)
    rp, ldr, rdt = None, 0, CFT_BF16
    if residual is not None:
        rp, ldr, rdt = residual.data_ptr(), residual.stride(0), _dt(residual.dtype)
    lib = _lib.load()
    abytes = (rows * K * x.element_size() + rows * pk.n_valid * out.element_size() + pk.w.numel() * x.element_size()
              + (0 if residual is None else rows * pk.n_valid * residual.element_size()))
    st = _timed_gemm(lib, rows, pk,
                     (x.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr() if pk.bias is not None else None, rp,
                      out.data_ptr(), 1, 1, rows, pk.cin, x.stride(0), 0, pk.n, pk.kpad, 1, 1,
                      out.stride(0), 0, ldr, 0, act, _dt(x.dtype), _dt(out.dtype), rdt, _stream()), abytes, kind="linear")
    _lib.check(st, "cft_conv2d(linear)")
    return out


SPLITK_MAX_ROWS = 1024      # token rows (B * 128) up to which split-K pays with two forwards in flight: <= 8 pairs per GPU.  At 2 048 rows it still buys
                            # +4.8 % with ONE forward in flight (latency) but costs 0.6 - 2.6 % of the throughput with two (profiles/r05_splitk_ab.md): a
                            # latency-bound deployment raises this constant


def splitk_choice(rows, pk, dtype):
    """Number of K splits for ``linear_splitk`` (1 = run ``linear``): GEMMs whose 256 x 256 tiles would leave most of the 256 CUs idle and
    whose K loop is long enough to cut (the CFT block's out_proj / fc2 at B * 128 rows) - the smallest of 2 / 4 / 8 that yields >= 128
    workgroups, each split keeping >= 4 K steps."""
    bk = 32 if dtype == torch.float32 else 64
    if pk.k != 1 or pk.kpad != pk.cin or pk.cin % bk or 2 * pk.kpad * (4 if dtype == torch.float32 else 2) + 128 > 65536:
        return 1
    steps = pk.kpad // bk
    t256 = -(-rows // 256) * -(-pk.n // 256)
    # Measured (profiles/r05_splitk_ab.md): at 8192 rows (64 pairs) the split GEMMs gain 0.31 ms per forward and the LayerNorms that fold
    # 2-4 x [rows, d] fp32 partial sums lose 0.50 ms (HBM bytes); at 1024 rows (8 pairs) everything is latency-bound and the CFT block drops
    # from 2.37 to 1.84 ms.  So: only where the un-split GEMM leaves >= 3/4 of the chip idle, and only for small row counts (SPLITK_MAX_ROWS).
    if t256 >= 64 or rows > SPLITK_MAX_ROWS:
        return 1
    best = 1
    for s in (2, 4, 8):
        if steps % s or steps // s < 4 or s * rows * pk.n >= 2 ** 31:
            break
        best = s
        if t256 * s >= 128:      # half the CUs busy is enough: every further split doubles the partial sums the LayerNorm has to fold
            break
    return best


def linear_splitk(x, pk, splits):
    """fp32 partial sums [splits, rows, pk.n] of ``x @ w.T (+ bias)`` over ``splits`` slices of K (cft_linear_splitk); summed in order they equal
    ``linear(x, pk, out_dtype=float32)`` up to fp32 rounding.  ``layernorm_reduce`` folds them into the residual stream."""
    _require_cuda(x, "linear_splitk")
    rows, K = x.shape
    if K != pk.cin or x.stride(1) != 1:
        raise ValueError(f"linear_splitk: input [{rows},{K}] does not match packed weight (cin {pk.cin})")
    parts = torch.empty((splits, rows, pk.n), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    abytes = rows * K * x.element_size() + splits * rows * pk.n_valid * 4 + pk.w.numel() * x.element_size()
    st = _timed(f"linear_k{pk.k}s{pk.s}_n{pk.n}_K{pk.kpad}", rows * pk.flops_per_row, abytes,
                lambda: lib.cft_linear_splitk(x.data_ptr(), pk.w.data_ptr(), pk.bias.data_ptr() if pk.bias is not None else None, parts.data_ptr(),
                                              rows, pk.cin, x.stride(0), pk.n, pk.kpad, splits, _dt(x.dtype), _stream()))
    _lib.check(st, "cft_linear_splitk")
    return parts


def layernorm_reduce(x, parts, gamma, beta, out_dtype, eps=1e-5):
    """x (float32 [rows, C], IN PLACE) += parts[0] + parts[1] + ... (float32 [n, rows, C]); returns LayerNorm(x) in ``out_dtype``."""
    _require_cuda(x, "layernorm_reduce")
    rows, C = x.shape
    if parts.dtype != torch.float32 or tuple(parts.shape[1:]) != (rows, C) or not parts.is_contiguous() or not x.is_contiguous() or x.dtype != torch.float32:
        raise ValueError(f"layernorm_reduce: x {tuple(x.shape)} {x.dtype} / parts {tuple(parts.shape)} {parts.dtype} must be contiguous fp32 with matching rows")
    out = torch.empty((rows, C), dtype=out_dtype, device=x.device)
    lib = _lib.load()
    st = _timed("cft_layernorm", 0.0, rows * C * (8.0 + 4.0 * parts.shape[0] + out.element_size()),
                lambda: lib.cft_layernorm_reduce(x.data_ptr(), parts.data_ptr(), parts.shape[0], gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                                 rows, C, eps, _dt(out_dtype), _stream()))
    _lib.check(st, "cft_layernorm_reduce")
    return out


def focus_s2d(img, dtype):
    """[B,3,H,W] float image batch (NCHW) -> [B,16,H/2,W/2] NHWC in ``dtype`` (12 real channels)."""
    _require_cuda(img, "focus_s2d")
    if img.dtype == torch.uint8:
        # raw camera bytes, e.g. ``img6[:, :3]`` / ``img6[:, 3:]`` of the reference's [B,6,H,W] uint8 batch:
        # normalisation (/255), channel split (strides) and cast are fused into the gather
        B, C, H, W = img.shape
        if C != 3 or img.stride(3) != 1:
            raise ValueError("focus_s2d: uint8 input must be [B,3,H,W] with contiguous rows")
        out = new_nhwc(B, H // 2, W // 2, 16, dtype, img.device)
        st = _lib.load().cft_focus_s2d_u8(img.data_ptr(), img.stride(0), img.stride(1), img.stride(2), out.data_ptr(),
                                          B, H, W, 1.0 / 255.0, _dt(dtype), _stream())
        _lib.check(st, "cft_focus_s2d_u8")
        return out
    B, C, H, W = img.shape
    if C != 3:
        raise ValueError(f"focus_s2d: expected 3 input channels, got {C}")
    if img.dtype != torch.float32 or not img.is_contiguous():
        # half / bf16 / strided images -> contiguous fp32 NCHW with the conversion kernel: seen as an "NHWC" problem
        # with (batch*channel, row, 1 pixel, W "channels") the output is exactly the contiguous NCHW image
        if img.dtype not in COMPUTE_DTYPES:
            raise TypeError(f"focus_s2d: unsupported image dtype {img.dtype}")
        if W % 4:
            raise ValueError("focus_s2d: image width must be a multiple of 4")
        flat = torch.empty((B, 3, H, W), dtype=torch.float32, device=img.device)
        lib, es = _lib.load(), img.element_size()
        sb, sc, sh, sw = img.stride()
        if sb == 3 * sc:          # (batch, channel) collapse into one uniform axis of 3B planes
            calls = [(img.data_ptr(), flat.data_ptr(), (sc, sw, sh, 0), B * 3, H)]
        elif sc == H * sh:        # the three planes of an image are stacked rows, e.g. f[:, :3] / f[:, 3:] of the callers'
            calls = [(img.data_ptr(), flat.data_ptr(), (sb, sw, sh, 0), B, 3 * H)]      # [B,6,H,W] batch (test.py:112-113)
        else:                     # arbitrary strides: one launch per image
            calls = [(img.data_ptr() + b * sb * es, flat.data_ptr() + b * 3 * H * W * 4, (sc, sw, sh, 0), 3, H) for b in range(B)]
        for src, dst, (s0, s1, s2, s3), nb, nh in calls:
            _lib.check(lib.cft_to_nhwc(src, _dt(img.dtype), s0, s1, s2, s3, dst, W, 0, nb, W, W, nh, 1, CFT_F32, _stream()),
                       "cft_to_nhwc(image)")
        img = flat
    out = new_nhwc(B, H // 2, W // 2, 16, dtype, img.device)
    st = _lib.load().cft_focus_s2d(img.data_ptr(), out.data_ptr(), B, H, W, _dt(dtype), _stream())
    _lib.check(st, "cft_focus_s2d")
    return out


def focus_conv(img, pk, act, dtype):
    """Focus = space-to-depth + 3x3 Conv of the image batch ``img`` ([B,3,H,W] float or uint8, NCHW).
    bf16 with 32/48/64/80 output channels runs as ONE kernel (cft_focus_conv: no intermediate tensor, bit-identical);
    anything else as cft_focus_s2d + cft_conv2d."""
    _require_cuda(img, "focus_conv")
    if img.dim() != 4 or img.shape[1] != 3:
        raise ValueError(f"focus_conv: expected a [B,3,H,W] image batch, got {tuple(img.shape)}")
    fusable = (dtype in (torch.bfloat16, torch.float16) and pk.k == 3 and pk.s == 1 and pk.cin == 16 and pk.kpad == 192
               and pk.n in (32, 48, 64, 80) and act in (ACT_NONE, ACT_SILU)
               and (img.dtype in (torch.uint8, torch.float32) or (img.dtype == torch.float16 and dtype == torch.float16)))
    if not fusable:
        return conv2d(focus_s2d(img, dtype), pk, act)
    es = img.element_size()
    if img.stride(3) != 1 or any(img.stride(i) % 2 for i in range(3)) or img.data_ptr() % (2 * es):
        img = img.contiguous()
    B, _, H, W = img.shape
    if H % 2 or W % 2:
        raise ValueError("focus_conv: H and W must be even")
    u8 = img.dtype == torch.uint8
    kind = 1 if u8 else (2 if img.dtype == torch.float16 else 0)
    out = new_nhwc(B, H // 2, W // 2, pk.n, dtype, img.device)
    lib = _lib.load()
    args = (img.data_ptr(), kind, img.stride(0), img.stride(1), img.stride(2), 1.0 / 255.0 if u8 else 1.0,
            pk.w.data_ptr(), pk.kpad, pk.bias.data_ptr() if pk.bias is not None else None, out.data_ptr(),
            _view_ld(out, "focus_conv out"), 0, B, H, W, pk.n, act, _dt(dtype), _stream())
    if _launch_log is None:
        st = lib.cft_focus_conv(*args)
    else:   # counted with the GEMM family (it is the same implicit-GEMM MFMA work, on a dedicated kernel)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st = lib.cft_focus_conv(*args)
        e1.record()
        rows = B * (H // 2) * (W // 2)
        abytes = img.numel() * es + rows * pk.n_valid * 2 + pk.w.numel() * 2
        _launch_log.append((f"conv_focus_k3s1_n{pk.n}_K{pk.kpad}", rows * pk.flops_per_row, e0, e1, abytes))
    _lib.check(st, "cft_focus_conv")
    return out


def spp_maxpool(buf, C, ks):
    """In-place: buf [B,4C,H,W] NHWC, channels [0,C) already hold x; fills the three pooled slices."""
    _require_cuda(buf, "spp_maxpool")
    B, C4, H, W = buf.shape
    ld = _view_ld(buf, "spp_maxpool")
    st = _lib.load().cft_spp_maxpool(buf.data_ptr(), B, H, W, C, ld, ks[0], ks[1], ks[2], _dt(buf.dtype), _stream())
    _lib.check(st, "cft_spp_maxpool")
    return buf


def copy_channels(src, dst, up=0):
    """dst[b, :, y, x] = src[b, :, y >> up, x >> up] (dst is usually a channel slice of a concat buffer)."""
    _require_cuda(src, "copy_channels")
    src, ldi = as_nhwc(src)
    ldo = _view_ld(dst, "copy_channels dst")
    B, C, Ho, Wo = dst.shape
    if src.shape[0] != B or src.shape[1] != C or (src.shape[2] << up) != Ho or (src.shape[3] << up) != Wo:
        raise ValueError(f"copy_channels: {tuple(src.shape)} -> {tuple(dst.shape)} with up={up}")
    st = _lib.load().cft_copy_channels(src.data_ptr(), ldi, 0, dst.data_ptr(), ldo, 0, B, Ho, Wo, C, up, _dt(dst.dtype), _stream())
    _lib.check(st, "cft_copy_channels")
    return dst


def add(a, b, out=None):
    _require_cuda(a, "add")
    a, lda = as_nhwc(a)
    b, ldb = as_nhwc(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError(f"add: {tuple(a.shape)}/{a.dtype} vs {tuple(b.shape)}/{b.dtype}")
    B, C, H, W = a.shape
    if out is None:
        out = new_nhwc(B, H, W, C, a.dtype, a.device)
    ldo = _view_ld(out, "add out")
    st = _lib.load().cft_add(a.data_ptr(), lda, 0, b.data_ptr(), ldb, 0, out.data_ptr(), ldo, 0, B * H * W, C, _dt(a.dtype), _stream())
    _lib.check(st, "cft_add")
    return out


def gpt_tokenize(rgb, ir, pos_emb):
    """-> float32 tokens [B,128,C]."""
    _require_cuda(rgb, "gpt_tokenize")
    rgb, ld_r = as_nhwc(rgb)
    ir, ld_i = as_nhwc(ir)
    B, C, H, W = rgb.shape
    tokens = torch.empty((B, 128, C), dtype=torch.float32, device=rgb.device)
    lib = _lib.load()
    st = _timed("cft_tokenize", 0.0, 2.0 * B * H * W * C * rgb.element_size() + B * 128 * C * 4,
                lambda: lib.cft_gpt_tokenize(rgb.data_ptr(), ld_r, 0, ir.data_ptr(), ld_i, 0, pos_emb.data_ptr(), tokens.data_ptr(),
                                             B, H, W, C, _dt(rgb.dtype), _stream()))
    _lib.check(st, "cft_gpt_tokenize")
    return tokens


def layernorm(x, gamma, beta, out_dtype, eps=1e-5):
    """x float32 [rows, C] -> out_dtype [rows, C]."""
    _require_cuda(x, "layernorm")
    rows, C = x.shape
    out = torch.empty((rows, C), dtype=out_dtype, device=x.device)
    lib = _lib.load()
    st = _timed("cft_layernorm", 0.0, rows * C * (4.0 + out.element_size()),
                lambda: lib.cft_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), rows, C, eps, _dt(out_dtype), _stream()))
    _lib.check(st, "cft_layernorm")
    return out


def attention(qkv, B, heads, dk, dkp, pdrop=0.0):
    """``pdrop`` > 0: training-mode dropout of the attention probabilities inside the kernel."""
    _require_cuda(qkv, "attention")
    out = torch.empty((B * 128, heads * dkp), dtype=qkv.dtype, device=qkv.device)
    lib = _lib.load()
    seed = next_dropout_seed() if pdrop > 0 else 0
    st = _timed("cft_attention", 4.0 * 128 * 128 * dk * heads * B, 4.0 * B * 128 * heads * dkp * qkv.element_size(),
                lambda: lib.cft_attention(qkv.data_ptr(), out.data_ptr(), B, heads, dk, dkp, _dt(qkv.dtype), float(pdrop), seed, _stream()))
    _lib.check(st, "cft_attention")
    return out


def gpt_upsample_add(tokens, s, base, H, W, dtype):
    """bilinear(8x8 -> HxW) of stream ``s`` of ``tokens`` [B,128,C] (+ base) -> NHWC [B,C,H,W]."""
    _require_cuda(tokens, "gpt_upsample_add")
    B, T, C = tokens.shape
    out = new_nhwc(B, H, W, C, dtype, tokens.device)
    bp, ldb = None, 0
    if base is not None:
        base, ldb = as_nhwc(base)
        if tuple(base.shape) != (B, C, H, W) or base.dtype != dtype:
            raise ValueError("gpt_upsample_add: base shape/dtype mismatch")
        bp = base.data_ptr()
    lib = _lib.load()
    st = _timed("cft_upsample_add", 0.0, (2.0 if bp else 1.0) * B * H * W * C * out.element_size(),
                lambda: lib.cft_gpt_upsample_add(tokens.data_ptr(), s, bp, ldb, 0, out.data_ptr(), C, 0, B, H, W, C, _dt(dtype), _stream()))
    _lib.check(st, "cft_gpt_upsample_add")
    return out


def gpt_dual_tokens_ok(tokens):
    """What ``cft_gpt_upsample_add2`` hard-codes about its token tensor: [B, 128, C] (two streams x 8 x 8 anchors), fp32, contiguous,
    C a multiple of 4 and a 16-byte aligned base (it reads rows as float4 at ``b * 128 + s * 64``)."""
    return (isinstance(tokens, torch.Tensor) and tokens.is_cuda and tokens.dim() == 3 and tokens.shape[1] == 128
            and tokens.dtype == torch.float32 and tokens.is_contiguous() and tokens.shape[2] % 4 == 0 and tokens.data_ptr() % 16 == 0)


def gpt_upsample_add_dual(tokens, base0, base1, H, W, dtype, sum_out=None, want_sum=True):
    """Both streams of a CFT block and the Add behind them in one kernel (cft_gpt_upsample_add2):
    returns (base0 + up(tokens[:, :64]), base1 + up(tokens[:, 64:]), their sum or None); ``sum_out``: write the sum there
    (e.g. a channel slice of a planned concat buffer)."""
    _require_cuda(tokens, "gpt_upsample_add_dual")
    if not gpt_dual_tokens_ok(tokens):
        raise ValueError(f"gpt_upsample_add_dual: tokens must be a contiguous, 16-byte aligned fp32 [B, 128, C] tensor (C % 4 == 0), got "
                         f"{tuple(tokens.shape)} {tokens.dtype} (other anchor grids: gpt_upsample_add per stream)")
    B, T, C = tokens.shape
    base0, ldb0 = as_nhwc(base0)
    base1, ldb1 = as_nhwc(base1)
    for bse in (base0, base1):
        if tuple(bse.shape) != (B, C, H, W) or bse.dtype != dtype:
            raise ValueError("gpt_upsample_add_dual: base shape/dtype mismatch")
    out0 = new_nhwc(B, H, W, C, dtype, tokens.device)
    out1 = new_nhwc(B, H, W, C, dtype, tokens.device)
    sp, lds = None, 0
    if want_sum:
        if sum_out is None:
            sum_out = new_nhwc(B, H, W, C, dtype, tokens.device)
        if tuple(sum_out.shape) != (B, C, H, W) or sum_out.dtype != dtype:
            raise ValueError("gpt_upsample_add_dual: sum_out shape/dtype mismatch")
        lds = _view_ld(sum_out, "gpt_upsample_add_dual sum")
        sp = sum_out.data_ptr()
    else:
        sum_out = None
    lib = _lib.load()
    st = _timed("cft_upsample_add", 0.0, (5.0 if want_sum else 4.0) * B * H * W * C * out0.element_size(),
                lambda: lib.cft_gpt_upsample_add2(tokens.data_ptr(), base0.data_ptr(), ldb0, 0, base1.data_ptr(), ldb1, 0,
                                                  out0.data_ptr(), C, 0, out1.data_ptr(), C, 0, sp, lds, 0, B, H, W, C, _dt(dtype), _stream()))
    _lib.check(st, "cft_gpt_upsample_add2")
    return out0, out1, sum_out


def detect_decode(logits, raw, pred, anchors_px, na, no, stride, row0):
    """logits [B, ldl, ny, nx] NHWC float32; raw [B,na,ny,nx,no]; pred [B,rows,no] (filled at row0)."""
    B, ldl, ny, nx = logits.shape
    st = _lib.load().cft_detect_decode(logits.data_ptr(), ldl, raw.data_ptr(), pred.data_ptr(), anchors_px.data_ptr(),
                                       B, ny, nx, na, no, float(stride), row0, pred.shape[1], _stream())
    _lib.check(st, "cft_detect_decode")



# ------------------------------------------------------------------------------ training-mode forward
_dropout_state = {"seed": 0x5EED, "calls": 0}


def manual_dropout_seed(seed):
    """Seed of the counter-based dropout masks (every dropout call of a forward draws seed + call index)."""
    _dropout_state["seed"], _dropout_state["calls"] = int(seed), 0


def next_dropout_seed():
    _dropout_state["calls"] += 1
    return (_dropout_state["seed"] * 0x9E3779B1 + _dropout_state["calls"]) & 0xFFFFFFFFFFFFFFFF


def dropout_(x, p):
    """In-place nn.Dropout(p) in training mode on a contiguous tensor (cft_dropout)."""
    _require_cuda(x, "dropout_")
    if p <= 0.0:
        return x
    if not x.is_contiguous():
        raise ValueError("dropout_: tensor must be contiguous")
    st = _lib.load().cft_dropout(x.data_ptr(), x.numel(), float(p), next_dropout_seed(), _dt(x.dtype), _stream())
    _lib.check(st, "cft_dropout")
    return x


def add_rows_(x, y):
    """x += y for two [rows, C] tensors of one dtype (the CFT residual stream in training mode)."""
    rows, C = x.shape
    st = _lib.load().cft_add(x.data_ptr(), x.stride(0), 0, y.data_ptr(), y.stride(0), 0, x.data_ptr(), x.stride(0), 0, rows, C, _dt(x.dtype), _stream())
    _lib.check(st, "cft_add(rows)")
    return x


def batchnorm_train(y32, C, bn, act, residual=None, out=None, out_dtype=None):
    """act(BatchNorm2d(y32)) with BATCH statistics (+ residual) and the running-statistics update of ``bn``
    (an nn.BatchNorm2d in training mode): y32 is the fp32 conv output [B, pad8(C), H, W] (NHWC)."""
    _require_cuda(y32, "batchnorm_train")
    B, Cp, H, W = y32.shape
    ldx = _view_ld(y32, "batchnorm_train input")
    out_dtype = out_dtype or torch.float32
    if out is None:
        out = new_nhwc(B, H, W, Cp, out_dtype, y32.device)[:, :C] if Cp != C else new_nhwc(B, H, W, C, out_dtype, y32.device)
    if tuple(out.shape) != (B, C, H, W):
        raise ValueError(f"batchnorm_train: out has shape {tuple(out.shape)}, expected {(B, C, H, W)}")
    ldy = _view_ld(out, "batchnorm_train out")
    for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != y32.device:
            raise TypeError("batchnorm_train: BatchNorm parameters / running statistics must be contiguous fp32 tensors on the GPU")
    rp, ldr, rdt = None, 0, CFT_F32
    if residual is not None:
        if tuple(residual.shape) != (B, C, H, W):
            raise ValueError("batchnorm_train: residual shape mismatch")
        ldr, rp, rdt = _view_ld(residual, "batchnorm_train residual"), residual.data_ptr(), _dt(residual.dtype)
    lib = _lib.load()
    M = B * H * W
    ws = torch.empty((lib.cft_batchnorm_train_workspace(M, C),), dtype=torch.uint8, device=y32.device)
    track = bn.track_running_stats and bn.running_mean is not None
    if bn.momentum is not None:
        mom = float(bn.momentum)
    else:   # torch: momentum=None is the cumulative moving average, factor 1 / num_batches_tracked (after the increment)
        mom = 1.0 / (int(bn.num_batches_tracked) + 1) if (track and bn.num_batches_tracked is not None) else 0.0
    st = lib.cft_batchnorm_train(y32.data_ptr(), ldx, 0, M, C, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                 bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                 mom, float(bn.eps), rp, ldr, 0, rdt, out.data_ptr(), ldy, 0, act, _dt(out.dtype),
                                 ws.data_ptr(), ws.numel(), _stream())
    _lib.check(st, "cft_batchnorm_train")
    if track and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return out
