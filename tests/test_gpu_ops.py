"""Per-kernel parity: every C-ABI entry point against the CPU oracle on seeded inputs.

bf16 cases round the oracle's INPUTS (activations, weights) to bf16 first, so the remaining
difference is fp32 accumulation order plus the single output rounding - the tolerance is
relative to the output scale and is stated in ``helpers.tol``.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err, to_cpu_f32, to_dev_nhwc, tol

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
DTYPE_IDS = ["f32", "bf16", "f16"]
LOWP = [torch.bfloat16, torch.float16]


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _q(x, dtype):
    return x if dtype == torch.float32 else x.to(dtype).float()


CONV_CASES = [
    # B, H, W, Cin, Cout, k, s, act, residual
    (2, 16, 16, 64, 64, 1, 1, 1, False),
    (2, 16, 16, 64, 128, 3, 1, 1, True),
    (1, 20, 20, 128, 256, 3, 2, 1, False),
    (3, 13, 17, 32, 64, 3, 1, 1, True),       # ragged M, Cin < K step (taps share a K step)
    (2, 10, 10, 16, 32, 3, 1, 1, False),      # Focus-like Cin=16, K=144 (K tail)
    (1, 9, 11, 80, 160, 3, 2, 0, False),      # yolov5x-like widths: K step straddles taps, N tail
    (1, 8, 8, 512, 24, 1, 1, 0, False),       # Detect-like narrow N
    (2, 40, 40, 256, 256, 1, 1, 1, False),    # 128x128 tiles
    (4, 40, 40, 128, 128, 3, 1, 1, True),     # 128x128 tiles, 3x3
    (1, 6, 6, 1024, 512, 1, 1, 2, False),     # long K, GELU
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("case", CONV_CASES, ids=[f"c{i}" for i in range(len(CONV_CASES))])
def test_conv2d(dev, dtype, case):
    from msod_amd import ops
    B, H, W, Cin, Cout, k, s, act, use_res = case
    x = _q(_rnd(B, Cin, H, W, seed=1), dtype)
    w = _q(_rnd(Cout, Cin, k, k, seed=2, scale=1.0 / math.sqrt(Cin * k * k)), dtype)
    b = _rnd(Cout, seed=3, scale=0.5)
    ref = F.conv2d(x, w, b, s, k // 2)
    ref = F.silu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    res = None
    if use_res:
        res = _q(_rnd(*ref.shape, seed=4), dtype)
        ref = ref + res
    pk = ops.pack_conv(w, b, dtype, s=s, device=dev)
    y = ops.conv2d(to_dev_nhwc(x, dev, dtype), pk, act, residual=None if res is None else to_dev_nhwc(res, dev, dtype))
    torch.cuda.synchronize()
    got = to_cpu_f32(y)[:, :Cout]
    assert got.shape == ref.shape
    assert rel_err(got, ref) < tol(dtype), f"rel err {rel_err(got, ref):.3e}"


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
def test_conv2d_channel_slices_and_fp32_out(dev, dtype):
    """Input read from a channel slice, output written into a slice of a wider buffer, residual
    aliasing the output (the C3 / Bottleneck pattern), and bf16-in / fp32-out (Detect, GPT)."""
    from msod_amd import ops
    B, H, W, C = 2, 12, 12, 64
    wide = _q(_rnd(B, 2 * C, H, W, seed=5), dtype)
    w = _q(_rnd(C, C, 3, 3, seed=6, scale=1.0 / math.sqrt(C * 9)), dtype)
    b = _rnd(C, seed=7, scale=0.1)
    xin = wide[:, C:]
    ref = F.silu(F.conv2d(xin, w, b, 1, 1)) + wide[:, :C]
    buf = to_dev_nhwc(wide, dev, dtype)
    pk = ops.pack_conv(w, b, dtype, device=dev)
    ops.conv2d(buf[:, C:], pk, 1, residual=buf[:, :C], out=buf[:, :C])
    torch.cuda.synchronize()
    got = to_cpu_f32(buf)
    assert rel_err(got[:, :C], ref) < tol(dtype)
    assert torch.equal(got[:, C:], wide[:, C:]), "the untouched slice must be bit-identical"
    y32 = ops.conv2d(buf[:, C:], pk, 0, out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert y32.dtype == torch.float32
    assert rel_err(to_cpu_f32(y32), F.conv2d(xin, w, b, 1, 1)) < 2e-5 * (1 if dtype == torch.float32 else 50)


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
def test_linear_residual_fp32_stream(dev, dtype):
    """nn.Linear + bias + fp32 residual updated in place (CFT out-proj / fc2 epilogue)."""
    from msod_amd import ops
    rows, K, N = 256, 512, 128
    x = _q(_rnd(rows, K, seed=8), dtype)
    w = _q(_rnd(N, K, seed=9, scale=1 / math.sqrt(K)), dtype)
    b = _rnd(N, seed=10, scale=0.1)
    r = _rnd(rows, N, seed=11)
    ref = F.linear(x, w, b) + r
    pk = ops.pack_conv(w, b, dtype, device=dev)
    rd = r.to(dev)
    out = ops.linear(x.to(dev).to(dtype), pk, residual=rd, out=rd, out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert out.data_ptr() == rd.data_ptr()
    assert rel_err(out.cpu(), ref) < 2e-5 * (1 if dtype == torch.float32 else 20)


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
def test_focus(dev, dtype):
    from msod_amd import ops
    from oracle import cft_oracle as O
    img = torch.rand(2, 3, 32, 48, generator=torch.Generator().manual_seed(12))
    z = ops.focus_s2d(img.to(dev), dtype)
    torch.cuda.synchronize()
    ref = torch.cat([img[..., ::2, ::2], img[..., 1::2, ::2], img[..., ::2, 1::2], img[..., 1::2, 1::2]], 1)
    got = to_cpu_f32(z)
    assert torch.equal(got[:, :12], _q(ref, dtype)), "space-to-depth is a pure gather: must be exact"
    assert got[:, 12:].abs().max() == 0
    # and through the conv, against the oracle's Focus
    sd = {"f.conv.conv.weight": _q(_rnd(32, 12, 3, 3, seed=13, scale=0.1), dtype), "f.conv.conv.bias": _rnd(32, seed=14, scale=0.1)}
    pk = ops.pack_conv(sd["f.conv.conv.weight"], sd["f.conv.conv.bias"], dtype, cin_pad=16, device=dev)
    y = ops.conv2d(z, pk, 1)
    torch.cuda.synchronize()
    assert rel_err(to_cpu_f32(y), O.focus(sd, "f.", _q(img, dtype), 3, 1)) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("hw", [(20, 20), (7, 12), (40, 40)])
def test_spp_maxpool(dev, dtype, hw):
    from msod_amd import ops
    H, W = hw
    C = 32
    x = _q(_rnd(2, C, H, W, seed=15), dtype)
    buf = torch.zeros(2, 4 * C, H, W)
    buf[:, :C] = x
    d = to_dev_nhwc(buf, dev, dtype)
    ops.spp_maxpool(d, C, (5, 9, 13))
    torch.cuda.synchronize()
    got = to_cpu_f32(d)
    for i, k in enumerate((5, 9, 13)):
        assert torch.equal(got[:, (i + 1) * C:(i + 2) * C], F.max_pool2d(x, k, 1, k // 2)), f"k={k}: max is exact"
    assert torch.equal(got[:, :C], x)


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
def test_copy_upsample_add(dev, dtype):
    from msod_amd import ops
    a = _q(_rnd(2, 32, 6, 10, seed=16), dtype)
    b = _q(_rnd(2, 16, 12, 20, seed=17), dtype)
    out = torch.zeros(2, 48, 12, 20)
    d = to_dev_nhwc(out, dev, dtype)
    ops.copy_channels(to_dev_nhwc(a, dev, dtype), d[:, :32], up=1)
    ops.copy_channels(to_dev_nhwc(b, dev, dtype), d[:, 32:], up=0)
    torch.cuda.synchronize()
    ref = torch.cat([F.interpolate(a, scale_factor=2.0, mode="nearest"), b], 1)
    assert torch.equal(to_cpu_f32(d), ref)
    c = _q(_rnd(2, 48, 12, 20, seed=18), dtype)
    s = ops.add(d, to_dev_nhwc(c, dev, dtype))
    torch.cuda.synchronize()
    assert rel_err(to_cpu_f32(s), ref + c) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("hw", [(80, 80), (20, 20), (12, 20), (8, 8), (5, 7)])
def test_gpt_tokenize_and_upsample(dev, dtype, hw):
    """adaptive avg-pool (overlapping windows at 20->8, rectangular maps, maps smaller than 8x8)
    + pos_emb, and the bilinear de-tokeniser fused with the residual add."""
    from msod_amd import ops
    H, W = hw
    B, C = 2, 64
    rgb, ir = _q(_rnd(B, C, H, W, seed=19), dtype), _q(_rnd(B, C, H, W, seed=20), dtype)
    pe = _rnd(1, 128, C, seed=21, scale=0.3)
    tok = ops.gpt_tokenize(to_dev_nhwc(rgb, dev, dtype), to_dev_nhwc(ir, dev, dtype), pe.to(dev))
    torch.cuda.synchronize()
    r = F.adaptive_avg_pool2d(rgb, (8, 8)).reshape(B, C, -1)
    t = F.adaptive_avg_pool2d(ir, (8, 8)).reshape(B, C, -1)
    ref = torch.cat([r, t], 2).permute(0, 2, 1) + pe
    assert rel_err(tok.cpu(), ref) < 1e-5
    for s in (0, 1):
        up = ops.gpt_upsample_add(tok, s, to_dev_nhwc(rgb, dev, dtype), H, W, dtype)
        torch.cuda.synchronize()
        m = ref[:, s * 64:(s + 1) * 64].view(B, 8, 8, C).permute(0, 3, 1, 2).contiguous()
        want = rgb + F.interpolate(m, size=(H, W), mode="bilinear")
        assert rel_err(to_cpu_f32(up), want) < tol(dtype), f"stream {s}"
    up0 = ops.gpt_upsample_add(tok, 0, None, H, W, dtype)
    torch.cuda.synchronize()
    m = ref[:, :64].view(B, 8, 8, C).permute(0, 3, 1, 2).contiguous()
    assert rel_err(to_cpu_f32(up0), F.interpolate(m, size=(H, W), mode="bilinear")) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("hw,C", [((80, 80), 256), ((20, 20), 1024), ((12, 20), 64), ((5, 7), 64), ((40, 40), 1280)])
def test_gpt_upsample_add_dual(dev, dtype, hw, C):
    """cft_gpt_upsample_add2 = both streams' de-tokenise + Add2 and the Add behind them in one launch: out0 / out1 are bit-identical
    to two cft_gpt_upsample_add launches; the sum is formed from the unrounded fp32 sums (one rounding): within half an output ulp of
    the exact sum, so at least as close to it as add(out0, out1); fp32 is bit-identical to the three-launch path; a channel-slice
    destination for the sum (the planned concat buffer of the head) works."""
    from msod_amd import ops
    H, W = hw
    B = 2
    rgb, ir = _q(_rnd(B, C, H, W, seed=19), dtype), _q(_rnd(B, C, H, W, seed=20), dtype)
    tok = _rnd(B, 128, C, seed=23).to(dev)
    rd, idv = to_dev_nhwc(rgb, dev, dtype), to_dev_nhwc(ir, dev, dtype)
    a0 = ops.gpt_upsample_add(tok, 0, rd, H, W, dtype)
    a1 = ops.gpt_upsample_add(tok, 1, idv, H, W, dtype)
    three = ops.add(a0, a1)
    buf = ops.new_nhwc(B, H, W, 2 * C, dtype, dev)
    buf.zero_()
    o0, o1, osum = ops.gpt_upsample_add_dual(tok, rd, idv, H, W, dtype, sum_out=buf[:, C:])
    p0, p1, none = ops.gpt_upsample_add_dual(tok, rd, idv, H, W, dtype, want_sum=False)
    torch.cuda.synchronize()
    assert none is None and torch.equal(o0, a0) and torch.equal(o1, a1) and torch.equal(p0, a0) and torch.equal(p1, a1)
    assert osum.data_ptr() == buf[:, C:].data_ptr() and float(buf[:, :C].abs().max()) == 0.0
    m = tok.cpu().view(B, 2, 8, 8, C).permute(0, 1, 4, 2, 3)
    exact = (rgb + F.interpolate(m[:, 0].contiguous(), size=(H, W), mode="bilinear")) + (ir + F.interpolate(m[:, 1].contiguous(), size=(H, W), mode="bilinear"))
    if dtype == torch.float32:
        assert torch.equal(osum, three)
    else:
        e_dual, e_three = (to_cpu_f32(osum) - exact).abs(), (to_cpu_f32(three) - exact).abs()
        assert e_dual.max() <= e_three.max() + 1e-6 and e_dual.mean() <= e_three.mean()
    assert rel_err(to_cpu_f32(osum), exact) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("B,H,W,cin,k,s,n1,n2", [(2, 160, 160, 64, 3, 2, 128, 128), (1, 51, 37, 64, 3, 2, 128, 128), (3, 40, 24, 128, 3, 1, 128, 64),
                                                 (2, 33, 20, 64, 1, 1, 128, 72), (1, 9, 9, 192, 3, 2, 128, 128),
                                                 (2, 160, 160, 128, 3, 2, 256, 256), (1, 45, 37, 128, 3, 2, 256, 256), (2, 24, 40, 64, 3, 1, 256, 136),
                                                 (1, 20, 20, 256, 3, 2, 256, 256)])
def test_conv2d_chain_is_bit_identical_to_two_launches(dev, dtype, B, H, W, cin, k, s, n1, n2):
    """cft_conv2d_chain (stride-2 Conv + the C3's packed cv1|cv2 in one kernel; the 128- / 256-channel tensor between them stays in
    LDS): same values, roundings and k order as the two cft_conv2d launches -> bit-identical, for full tiles, pixel tails, n2 < n1,
    other first-layer geometries (1x1, stride 1, the chunk-major K walk of 256 input channels), both tile forms (128 wide: weights of
    the second layer resident; 256 wide: streamed); a channel-slice destination works; ineligible pairs are refused."""
    from msod_amd import ops
    x = to_dev_nhwc(_q(_rnd(B, cin, H, W, seed=31), dtype), dev, dtype)
    pk1 = ops.pack_conv(_rnd(n1, cin, k, k, seed=32) * (2.0 / (cin * k * k)) ** 0.5, _rnd(n1, seed=33) * 0.1, dtype, s=s, device=dev)
    pk2 = ops.pack_conv(_rnd(n2, n1, 1, 1, seed=34) * (2.0 / n1) ** 0.5, _rnd(n2, seed=35) * 0.1, dtype, device=dev)
    assert ops.conv2d_chain_ok(x, pk1, pk2)
    two = ops.conv2d(ops.conv2d(x, pk1, ops.ACT_SILU), pk2, ops.ACT_SILU)
    one = ops.conv2d_chain(x, pk1, pk2, ops.ACT_SILU)
    Ho, Wo = two.shape[2], two.shape[3]
    buf = ops.new_nhwc(B, Ho, Wo, pk2.n + 16, dtype, dev)
    buf.zero_()
    sl = ops.conv2d_chain(x, pk1, pk2, ops.ACT_NONE, out=buf[:, 8:8 + pk2.n])
    lin = ops.conv2d(ops.conv2d(x, pk1, ops.ACT_SILU), pk2, ops.ACT_NONE)
    torch.cuda.synchronize()
    assert torch.equal(one, two) and float(two.float().abs().max()) > 0.1
    assert torch.equal(sl, lin) and float(buf[:, :8].float().abs().max()) == 0.0 and float(buf[:, 8 + pk2.n:].float().abs().max()) == 0.0
    pk_bad = ops.pack_conv(_rnd(64, cin, k, k, seed=36), None, dtype, s=s, device=dev)          # first layer neither 128 nor 256 wide
    pk2_bad = ops.pack_conv(_rnd(n2, 64, 1, 1, seed=37), None, dtype, device=dev)
    assert not ops.conv2d_chain_ok(x, pk_bad, pk2_bad)
    with pytest.raises(ValueError):
        ops.conv2d_chain(x, pk_bad, pk2_bad, ops.ACT_SILU)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("B,H,W,k,n2", [(2, 40, 40, 3, 256), (1, 19, 23, 3, 256), (3, 16, 16, 3, 64), (1, 40, 40, 1, 256), (64, 40, 40, 3, 256)])
def test_conv2d_chain_res_is_bit_identical_to_two_launches(dev, dtype, B, H, W, k, n2):
    """cft_conv2d_chain_res - Bottleneck j's 3x3 conv WITH its shortcut and Bottleneck j + 1's 1x1 conv in one kernel (256 channels): the
    shortcut tile is added in fp32 before the one rounding, exactly like the plain epilogue does -> y1 and y2 are bit-identical to
    conv2d(x, pk1, residual=res) followed by conv2d(y1, pk2); full tiles (the bench's 64 x 40 x 40), pixel tails, a narrow second layer, a 1x1
    first layer, y1 written in place over the shortcut, channel-slice outputs."""
    from msod_amd import ops
    cin = 256
    x = to_dev_nhwc(_q(_rnd(B, cin, H, W, seed=41), dtype), dev, dtype)
    res = to_dev_nhwc(_q(_rnd(B, 256, H, W, seed=42), dtype), dev, dtype)
    pk1 = ops.pack_conv(_rnd(256, cin, k, k, seed=43) * (2.0 / (cin * k * k)) ** 0.5, _rnd(256, seed=44) * 0.1, dtype, device=dev)
    pk2 = ops.pack_conv(_rnd(n2, 256, 1, 1, seed=45) * (2.0 / 256) ** 0.5, _rnd(n2, seed=46) * 0.1, dtype, device=dev)
    assert ops.conv2d_chain_res_ok(x, pk1, pk2, any_size=True)
    assert ops.conv2d_chain_res_ok(x, pk1, pk2) == (B * H * W >= ops.CHAIN_RES_MIN_ROWS)      # small problems: the separate launches' smaller tiles
    y1_two = ops.conv2d(x, pk1, ops.ACT_SILU, residual=res)
    y2_two = ops.conv2d(y1_two, pk2, ops.ACT_SILU)
    y1, y2 = ops.conv2d_chain_res(x, pk1, res, pk2, ops.ACT_SILU)
    torch.cuda.synchronize()
    assert torch.equal(y1, y1_two) and torch.equal(y2, y2_two) and float(y2.float().abs().max()) > 0.1
    # channel-slice outputs, y1 in place over the shortcut
    buf1 = ops.new_nhwc(B, H, W, 256 + 16, dtype, dev); buf1.zero_()
    buf2 = ops.new_nhwc(B, H, W, pk2.n + 8, dtype, dev); buf2.zero_()
    ops.conv2d_chain_res(x, pk1, res, pk2, ops.ACT_SILU, out1=buf1[:, 8:264], out2=buf2[:, :pk2.n])
    res2 = res.clone()
    y1_ip, y2_ip = ops.conv2d_chain_res(x, pk1, res2, pk2, ops.ACT_SILU, out1=res2)
    torch.cuda.synchronize()
    assert torch.equal(buf1[:, 8:264], y1_two) and torch.equal(buf2[:, :pk2.n], y2_two)
    assert float(buf1[:, :8].float().abs().max()) == 0.0 and float(buf1[:, 264:].float().abs().max()) == 0.0 and float(buf2[:, pk2.n:].float().abs().max()) == 0.0
    assert torch.equal(res2, y1_two) and torch.equal(y2_ip, y2_two)
    pk128 = ops.pack_conv(_rnd(128, cin, k, k, seed=47), None, dtype, device=dev)            # 128-channel first layers: the Bottleneck kernel's domain
    pk2b = ops.pack_conv(_rnd(128, 128, 1, 1, seed=48), None, dtype, device=dev)
    assert not ops.conv2d_chain_res_ok(x, pk128, pk2b, any_size=True)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "fp16", "f32"])
@pytest.mark.parametrize("rows,n,K,splits", [(1024, 1024, 4096, 2), (1024, 512, 2048, 4), (520, 256, 1024, 4), (1024, 1024, 1024, 2), (256, 320, 1280, 4), (8192, 1024, 4096, 2)])
def test_linear_splitk_and_layernorm_reduce(dev, dtype, rows, n, K, splits):
    """cft_linear_splitk: fp32 partial sums over slices of K (bias in slice 0) - summed they equal the one-launch fp32-output linear up to
    fp32 summation order; identical from run to run.  cft_layernorm_reduce: x += sum of the parts (in order), y = LayerNorm(x), vs torch."""
    from msod_amd import ops
    x = _q(_rnd(rows, K, seed=51), dtype).to(dev).to(dtype)
    pk = ops.pack_conv(_rnd(n, K, seed=52) * (1.0 / K) ** 0.5, _rnd(n, seed=53) * 0.1, dtype, device=dev)
    one = ops.linear(x, pk, out_dtype=torch.float32)
    parts = ops.linear_splitk(x, pk, splits)
    parts2 = ops.linear_splitk(x, pk, splits)
    torch.cuda.synchronize()
    assert parts.shape == (splits, rows, pk.n) and torch.equal(parts, parts2)
    tot = parts.sum(0)
    assert (tot - one).abs().max().item() <= 2e-5 * max(1.0, one.abs().max().item()) * (8 if dtype == torch.float32 else 1)
    assert float(parts[1].abs().max()) > 0.01                                # every slice carries its share
    # the slices are what they claim to be: slice s = x[:, s*K/S:(s+1)*K/S] @ w[:, same].T (+ bias for s = 0), against torch in fp32
    kc = K // splits
    wf = pk.w[:n, :K].float()
    for s_ in (0, splits - 1):
        want = x[:, s_ * kc:(s_ + 1) * kc].float() @ wf[:, s_ * kc:(s_ + 1) * kc].T + (pk.bias[:n] if s_ == 0 else 0.0)
        assert (parts[s_][:, :n] - want).abs().max().item() <= 1e-3 * max(1.0, want.abs().max().item())
    res = _rnd(rows, pk.n, seed=54).to(dev)
    gamma, beta = (1.0 + 0.1 * _rnd(pk.n, seed=55)).to(dev), (0.1 * _rnd(pk.n, seed=56)).to(dev)
    xr = res.clone()
    y = ops.layernorm_reduce(xr, parts, gamma, beta, dtype)
    want_x = res.clone()
    for s_ in range(splits):
        want_x += parts[s_]
    want_y = torch.nn.functional.layer_norm(want_x, (pk.n,), gamma, beta, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(xr, want_x)                                           # fixed order: bit-exact running sum
    assert (y.float() - want_y).abs().max().item() <= (1e-4 if dtype == torch.float32 else 4e-2)
    fc2 = ops.pack_conv(torch.zeros(1024, 4096), None, torch.bfloat16, device=dev)
    assert ops.splitk_choice(8192, fc2, torch.bfloat16) == 1            # 64 pairs: the partial-sum traffic costs more than the split gains
    assert ops.splitk_choice(1024, fc2, torch.bfloat16) == 8 and ops.splitk_choice(2048, fc2, torch.bfloat16) == 1      # SPLITK_MAX_ROWS
    assert ops.splitk_choice(1024, ops.pack_conv(torch.zeros(4096, 1024), None, torch.bfloat16, device=dev), torch.bfloat16) == 1   # wide N: enough tiles


@pytest.mark.parametrize("seed", range(10))
def test_conv2d_chain_random_geometries(dev, seed):
    """Seeded sweep over what cft_conv2d_chain_ok accepts (first layer 1x1 / 3x3 / 5x5, stride 1 / 2, 64..320 input channels, odd image
    sizes, batch 1..5, ragged second-layer widths, both tile forms, both dtypes): bit-identical to the two launches, and within the
    16-bit tolerance of the fp32 torch reference of the same two layers."""
    from msod_amd import ops
    rng = np.random.RandomState(1000 + seed)
    dtype = (torch.bfloat16, torch.float16)[seed % 2]
    n1 = (128, 256)[int(rng.randint(2))]
    k = (1, 3, 3, 5)[int(rng.randint(4))]
    s = int(rng.randint(1, 3))
    cin = 64 * int(rng.randint(1, 6 if k < 5 else 3))
    B, H, W = int(rng.randint(1, 6)), int(rng.randint(5, 60)), int(rng.randint(5, 60))
    n2 = 8 * int(rng.randint(1, n1 // 8 + 1))
    xf = _q(_rnd(B, cin, H, W, seed=seed), dtype)
    w1, b1 = _rnd(n1, cin, k, k, seed=seed + 1) * (2.0 / (cin * k * k)) ** 0.5, _rnd(n1, seed=seed + 2) * 0.1
    w2, b2 = _rnd(n2, n1, 1, 1, seed=seed + 3) * (2.0 / n1) ** 0.5, _rnd(n2, seed=seed + 4) * 0.1
    x = to_dev_nhwc(xf, dev, dtype)
    pk1, pk2 = ops.pack_conv(w1, b1, dtype, s=s, device=dev), ops.pack_conv(w2, b2, dtype, device=dev)
    assert ops.conv2d_chain_ok(x, pk1, pk2), (n1, k, s, cin, B, H, W, n2)
    one = ops.conv2d_chain(x, pk1, pk2, ops.ACT_SILU)
    two = ops.conv2d(ops.conv2d(x, pk1, ops.ACT_SILU), pk2, ops.ACT_SILU)
    torch.cuda.synchronize()
    assert torch.equal(one, two), (n1, k, s, cin, B, H, W, n2)
    mid = _q(F.silu(F.conv2d(xf, _q(w1, dtype), b1, stride=s, padding=k // 2)), dtype)
    ref = F.silu(F.conv2d(mid, _q(w2, dtype), b2))
    assert rel_err(to_cpu_f32(one), ref) < tol(dtype)


@pytest.mark.parametrize("C", [64, 256, 320, 1024, 1280])
def test_layernorm(dev, C):
    from msod_amd import ops
    x = _rnd(300, C, seed=22) * 3 + 1
    g, b = _rnd(C, seed=23) * 0.2 + 1, _rnd(C, seed=24) * 0.1
    ref = F.layer_norm(x, (C,), g, b, 1e-5)
    y = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), torch.float32)
    yb = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), torch.bfloat16)
    yh = ops.layernorm(x.to(dev), g.to(dev), b.to(dev), torch.float16)
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), ref) < 1e-5
    assert rel_err(yb.float().cpu(), ref) < 5e-3
    assert rel_err(yh.float().cpu(), ref) < 6e-4


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("d_model", [64, 256, 512, 1024, 1280, 160])
def test_self_attention_module(dev, dtype, d_model):
    """SelfAttention (fused QKV GEMM + attention core + out-proj) against the oracle; covers head
    widths 8..160 incl. the padded ones (8, 20) and forces peaked softmax rows via larger weights."""
    from msod_amd.models.common import SelfAttention
    from oracle import cft_oracle as O
    B, h = 2, 8
    sa = SelfAttention(d_model, d_model, d_model, h).eval()
    g = torch.Generator().manual_seed(25)
    with torch.no_grad():
        for lin in (sa.que_proj, sa.key_proj, sa.val_proj, sa.out_proj):
            lin.weight.copy_(_q(torch.randn(lin.weight.shape, generator=g) * (2.0 / math.sqrt(d_model)), dtype))
            lin.bias.copy_(torch.randn(lin.bias.shape, generator=g) * 0.1)
    x = _q(_rnd(B, 128, d_model, seed=26), dtype)
    sd = {"sa." + k: v for k, v in sa.state_dict().items()}
    ref = O.self_attention(sd, "sa.", x, h)
    sa = sa.to(dev)
    y = sa(x.view(B * 128, d_model).to(dev).to(dtype))
    torch.cuda.synchronize()
    got = to_cpu_f32(y).view(B, 128, -1)[..., :d_model]
    # 16-bit: q,k,v, P and the attention output are each rounded once -> a few 2^-8 (bf16) / 2^-11 (fp16) steps
    lim = {torch.float32: 5e-5, torch.bfloat16: 3e-2, torch.float16: 4e-3}[dtype]
    assert rel_err(got, ref) < lim, f"rel err {rel_err(got, ref):.3e}"


def test_detect_decode(dev):
    from msod_amd import ops
    B, ny, nx, na, no = 2, 5, 7, 3, 8
    logits = _rnd(B, 24, ny, nx, seed=27)
    anchors = torch.tensor([10., 13., 16., 30., 33., 23.])
    d = to_dev_nhwc(logits, dev, torch.float32)
    raw = torch.empty(B, na, ny, nx, no, device=dev)
    pred = torch.zeros(B, 200, no, device=dev)
    ops.detect_decode(d, raw, pred, anchors.to(dev), na, no, 16.0, 50)
    torch.cuda.synchronize()
    y = logits.view(B, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
    assert torch.equal(raw.cpu(), y)
    yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
    grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
    s = y.sigmoid()
    xy = (s[..., 0:2] * 2 - 0.5 + grid) * 16.0
    wh = (s[..., 2:4] * 2) ** 2 * anchors.view(1, na, 1, 1, 2)
    ref = torch.cat((xy, wh, s[..., 4:]), -1).view(B, -1, no)
    got = pred.cpu()
    assert rel_err(got[:, 50:50 + na * ny * nx], ref) < 1e-5
    assert got[:, :50].abs().max() == 0 and got[:, 50 + na * ny * nx:].abs().max() == 0


def test_errors_are_loud(dev):
    """Bad arguments must raise, never fall back or silently compute something else."""
    from msod_amd import ops
    pk = ops.pack_conv(torch.zeros(8, 8, 1, 1), None, torch.bfloat16, device=dev)
    with pytest.raises(ValueError):
        ops.conv2d(torch.zeros(1, 16, 4, 4, device=dev, dtype=torch.bfloat16), pk, 0)
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 8, 4, 4, dtype=torch.bfloat16), pk, 0)     # CPU tensor
    with pytest.raises(TypeError):
        ops.conv2d(torch.zeros(1, 8, 4, 4, device=dev, dtype=torch.float64), pk, 0)
    with pytest.raises(RuntimeError, match="compute dtype"):
        ops.conv2d(torch.zeros(1, 8, 4, 4, device=dev, dtype=torch.float16), pk, 0, out_dtype=torch.bfloat16)   # fp16 in, bf16 out


TILE_VARIANTS = [1, 2, 4, 6, 7, 8, 23, 27, 30, 32, 33, 51, 60, 63, 70, 71, 72, 73, 74, 75]


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("variant", TILE_VARIANTS)
def test_conv2d_every_tile_configuration(dev, dtype, variant):
    """Every tile configuration of the GEMM family (incl. the 16-wave 256x256 / 512x128 ones that the
    dispatcher only picks at large batch) on a problem with ragged M (2*33*31 rows), an N tail
    (320 = 256 + 64) and border taps; all must agree with the oracle."""
    from msod_amd import _lib, ops
    B, H, W, Cin, Cout, k = 2, 33, 31, 64, 320, 3
    x = _q(_rnd(B, Cin, H, W, seed=31), dtype)
    w = _q(_rnd(Cout, Cin, k, k, seed=32, scale=1.0 / math.sqrt(Cin * k * k)), dtype)
    b = _rnd(Cout, seed=33, scale=0.5)
    res = _q(_rnd(B, Cout, H, W, seed=34), dtype)
    ref = F.silu(F.conv2d(x, w, b, 1, 1)) + res
    pk = ops.pack_conv(w, b, dtype, device=dev)
    lib = _lib.load()
    lib.cft_set_conv_variant(variant)
    try:
        y = ops.conv2d(to_dev_nhwc(x, dev, dtype), pk, 1, residual=to_dev_nhwc(res, dev, dtype))
        torch.cuda.synchronize()
    finally:
        lib.cft_set_conv_variant(0)
    assert rel_err(to_cpu_f32(y), ref) < tol(dtype), f"variant {variant}: rel err {rel_err(to_cpu_f32(y), ref):.3e}"


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(64, 40, 256, 256, True), (64, 20, 512, 512, False)], ids=["p4_256ch", "p5_512ch"])
def test_conv3x3_wide_layers_at_the_benchmarked_row_count(dev, dtype, shape):
    """The kernel configuration bench.py actually times on the wide 3x3 layers: M = 64 x 40 x 40 =
    102 400 / 64 x 20 x 20 = 25 600 rows with Cin in {256, 512}, where the dispatcher picks the 256x256 16-wave tile AND
    the chunk-major K walk (Cin >= 256).  The automatic choice and the forced big-tile variants (27 = 256x256, 60 =
    128x256, 51 = 192x128) against F.conv2d, and bit-identical among themselves (every variant walks K alike)."""
    from msod_amd import _lib, ops
    B, HW, Cin, Cout, use_res = shape
    x = _q(_rnd(B, Cin, HW, HW, seed=51), dtype)
    w = _q(_rnd(Cout, Cin, 3, 3, seed=52, scale=1.0 / math.sqrt(Cin * 9)), dtype)
    b = _rnd(Cout, seed=53, scale=0.5)
    ref = F.silu(F.conv2d(x, w, b, 1, 1))
    res = None
    if use_res:
        res = _q(_rnd(*ref.shape, seed=54), dtype)
        ref = ref + res
    pk = ops.pack_conv(w, b, dtype, device=dev)
    xd = to_dev_nhwc(x, dev, dtype)
    rd = None if res is None else to_dev_nhwc(res, dev, dtype)
    lib = _lib.load()
    outs = {}
    try:
        for variant in (0, 27, 60, 51):
            lib.cft_set_conv_variant(variant)
            y = ops.conv2d(xd, pk, 1, residual=rd)
            torch.cuda.synchronize()
            outs[variant] = to_cpu_f32(y)
    finally:
        lib.cft_set_conv_variant(0)
    for variant, got in outs.items():
        assert rel_err(got, ref) < tol(dtype), f"variant {variant}: rel err {rel_err(got, ref):.3e}"
        assert torch.equal(got, outs[0]), f"variant {variant} differs from the automatic choice"


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
def test_focus_uint8_pair(dev, dtype):
    """The reference's callers hold the pair as one uint8 [B,6,H,W] tensor (test.py:106-113); slicing it and
    handing the uint8 views to Focus must equal `.float()/255` + split + the fp32 Focus path."""
    from msod_amd import ops
    img6 = torch.randint(0, 256, (2, 6, 32, 48), dtype=torch.uint8, generator=torch.Generator().manual_seed(40))
    d6 = img6.to(dev)
    ref = img6.float() / 255.0
    for lo in (0, 3):
        z8 = ops.focus_s2d(d6[:, lo:lo + 3], dtype)
        zf = ops.focus_s2d(ref[:, lo:lo + 3].contiguous().to(dev), dtype)
        torch.cuda.synchronize()
        assert rel_err(to_cpu_f32(z8), to_cpu_f32(zf)) < (1e-6 if dtype == torch.float32 else tol(dtype))


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "f16"])
@pytest.mark.parametrize("n", [32, 48, 64, 80])
@pytest.mark.parametrize("hw", [(32, 32), (96, 160), (40, 296)])
def test_focus_conv_fused_is_bit_identical(dev, n, hw, dtype):
    """cft_focus_conv (image -> space-to-depth -> 3x3 conv -> SiLU in one kernel) against the two-kernel path
    (cft_focus_s2d + cft_conv2d), for float and uint8 images: same products, same k order -> same bits; and
    against the oracle's Focus within the bf16 tolerance."""
    from msod_amd import ops
    from oracle import cft_oracle as O
    H, W = hw
    g = torch.Generator().manual_seed(50 + n)
    sd = {"f.conv.conv.weight": _q(_rnd(n, 12, 3, 3, seed=51, scale=0.1), dtype), "f.conv.conv.bias": _rnd(n, seed=52, scale=0.1)}
    pk = ops.pack_conv(sd["f.conv.conv.weight"], sd["f.conv.conv.bias"], dtype, cin_pad=16, device=dev)
    img = torch.rand(3, 3, H, W, generator=g)
    img6 = torch.randint(0, 256, (3, 6, H, W), dtype=torch.uint8, generator=g).to(dev)
    for act in (1, 0):
        fused = ops.focus_conv(img.to(dev), pk, act, dtype)
        two = ops.conv2d(ops.focus_s2d(img.to(dev), dtype), pk, act)
        torch.cuda.synchronize()
        assert torch.equal(fused.float().cpu(), two.float().cpu()), f"float image, act={act}"
    for lo in (0, 3):
        view = img6[:, lo:lo + 3]
        fused = ops.focus_conv(view, pk, 1, dtype)
        two = ops.conv2d(ops.focus_s2d(view, dtype), pk, 1)
        torch.cuda.synchronize()
        assert torch.equal(fused.float().cpu(), two.float().cpu()), f"uint8 view at channel {lo}"
    y = ops.focus_conv(img.to(dev), pk, 1, dtype)
    torch.cuda.synchronize()
    assert rel_err(to_cpu_f32(y), O.focus(sd, "f.", _q(img, dtype), 3, 1)) < tol(dtype)
    if dtype == torch.float16:          # half images (`img.half()`, reference test.py:107) feed the fused kernel directly
        yh = ops.focus_conv(img.to(dev).half(), pk, 1, dtype)
        torch.cuda.synchronize()
        assert rel_err(to_cpu_f32(yh), O.focus(sd, "f.", img.half().float(), 3, 1)) < tol(dtype)


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "f16"])
@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("shortcut", [True, False])
@pytest.mark.parametrize("hw", [(8, 32), (24, 40), (40, 72), (80, 80), (17, 15)])
def test_bottleneck_fused_is_bit_identical(dev, shortcut, hw, dtype, C):
    """cft_bottleneck (1x1 -> SiLU -> 3x3 -> SiLU -> + shortcut in one kernel; 64 channels: 3x3 weights LDS-resident,
    128 channels: activation patch resident + weights streamed through a ring) against the two cft_conv2d launches it
    replaces: same bits; input and output as channel slices of wider buffers; maps that are not multiples of the
    16 x 16 / 8 x 32 tiles; and against the oracle's bottleneck within the 16-bit tolerance."""
    from msod_amd import ops
    from oracle import cft_oracle as O
    H, W = hw
    sd = {"m.cv1.conv.weight": _q(_rnd(C, C, 1, 1, seed=61, scale=1.2 / math.sqrt(C)), dtype), "m.cv1.conv.bias": _rnd(C, seed=62, scale=0.1),
          "m.cv2.conv.weight": _q(_rnd(C, C, 3, 3, seed=63, scale=0.4 / math.sqrt(C)), dtype), "m.cv2.conv.bias": _rnd(C, seed=64, scale=0.1)}
    pk1 = ops.pack_conv(sd["m.cv1.conv.weight"], sd["m.cv1.conv.bias"], dtype, device=dev)
    pk2 = ops.pack_conv(sd["m.cv2.conv.weight"], sd["m.cv2.conv.bias"], dtype, device=dev)
    x = _q(_rnd(3, C, H, W, seed=65), dtype)
    wide = torch.zeros(3, 2 * C, H, W)
    wide[:, :C] = x
    d = to_dev_nhwc(wide, dev, dtype)            # x = first half of a 2C-channel buffer (as inside C3)
    xin = d[:, :C]
    assert ops.bottleneck_kernel_covers(xin, pk1, pk2, 1, 1)
    two = ops.conv2d(ops.conv2d(xin, pk1, 1), pk2, 1, residual=xin if shortcut else None)
    fused = ops.bottleneck(xin, pk1, pk2, shortcut)
    ops.bottleneck(xin, pk1, pk2, shortcut, out=d[:, C:])      # disjoint slice of the same buffer
    torch.cuda.synchronize()
    assert torch.equal(fused.float().cpu(), two.float().cpu())
    assert torch.equal(d[:, C:].float().cpu(), two.float().cpu())
    assert torch.equal(d[:, :C].float().cpu(), x), "input slice untouched"
    with pytest.raises(RuntimeError):
        ops.bottleneck(xin, pk1, pk2, shortcut, out=xin)         # in-place is refused (halo reads)
    ref = O.bottleneck(sd, "m.", x, shortcut)
    assert rel_err(to_cpu_f32(fused), ref) < 2 * tol(dtype)   # two 16-bit roundings (hidden tensor, output)


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "f16"])
def test_bottleneck_pack_w2_stage_images(dev, dtype):
    """cft_bottleneck_pack_w2: stage u of the image is k = 32u .. 32u + 31 of every row, row n at n * 64 B, k-granule kg of
    the stage in 16-byte slot kg ^ h((n / 4) & 3), h = (0, 2, 3, 1) - the order the 128-channel kernel keeps a stage in LDS."""
    from msod_amd import _lib, ops
    C = 128
    g = torch.Generator().manual_seed(11)
    pk2 = ops.pack_conv(torch.randn(C, C, 3, 3, generator=g), None, dtype, device=dev)
    stages = torch.empty_like(pk2.w)
    lib = _lib.load()
    _lib.check(lib.cft_bottleneck_pack_w2(pk2.w.data_ptr(), pk2.kpad, C, stages.data_ptr(), ops._dt(dtype), None), "pack")
    torch.cuda.synchronize()
    w = pk2.w.cpu().view(torch.int16).view(C, 36, 4, 8)                  # [n][u][kg][8]
    got = stages.cpu().view(torch.int16).view(36, C, 4, 8)              # [u][n][slot][8]
    h = torch.tensor([0, 2, 3, 1])
    n = torch.arange(C)
    for slot in range(4):
        kg = slot ^ h[(n >> 2) & 3]                                      # [n]
        want = w[n, :, kg, :].permute(1, 0, 2)                           # [u][n][8]
        assert torch.equal(got[:, :, slot, :], want), slot


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "f16"])
def test_bottleneck128_many_tiles_per_workgroup(dev, dtype):
    """More tiles than workgroup slots (12 x 50 = 600 > 512, so workgroups start while others are mid-tile), at the
    BASELINE map size of the 128-channel stage: the fused kernel equals the two cft_conv2d launches bit for bit."""
    from msod_amd import _lib, ops
    C, B, H, W = 128, 12, 80, 80
    g = torch.Generator().manual_seed(5)
    pk1 = ops.pack_conv(torch.randn(C, C, 1, 1, generator=g) / C ** 0.5, torch.randn(C, generator=g) * 0.1, dtype, device=dev)
    pk2 = ops.pack_conv(torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5), torch.randn(C, generator=g) * 0.1, dtype, device=dev)
    x = ops.new_nhwc(B, H, W, C, dtype, dev)
    x.copy_(torch.randn(x.shape, generator=g).to(dev))
    two = ops.conv2d(ops.conv2d(x, pk1, 1), pk2, 1, residual=x)
    lib = _lib.load()
    for variant in (0,):
        lib.cft_set_conv_variant(variant)
        try:
            y = ops.bottleneck(x, pk1, pk2, True)
            torch.cuda.synchronize()
        finally:
            lib.cft_set_conv_variant(0)
        assert torch.equal(y.float().cpu(), two.float().cpu()), variant


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("ks", [(5, 9, 13), (3, 5, 7), (3, 5, 9), (5, 7, 13)])
def test_spp_maxpool_kernel_sizes(dev, dtype, ks):
    """Chained fast path (k, 2k-1, 3k-2) and the generic kernel: both exact against F.max_pool2d, on a
    64-channel map (four-granule workgroups) that is not square."""
    from msod_amd import ops
    H, W, C = 20, 12, 64
    x = _q(_rnd(3, C, H, W, seed=70), dtype)
    buf = torch.zeros(3, 4 * C, H, W)
    buf[:, :C] = x
    d = to_dev_nhwc(buf, dev, dtype)
    ops.spp_maxpool(d, C, ks)
    torch.cuda.synchronize()
    got = to_cpu_f32(d)
    for i, k in enumerate(ks):
        assert torch.equal(got[:, (i + 1) * C:(i + 2) * C], F.max_pool2d(x, k, 1, k // 2)), f"k={k}"
    assert torch.equal(got[:, :C], x)


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
def test_plain_nchw_tensor_into_a_module(dev, dtype):
    """Boundary: a module of models/common.py called with an ordinary contiguous NCHW tensor (as the reference's
    modules are) converts it with the cft_to_nhwc kernel - no ATen copy - and computes the same as on NHWC input;
    also the dtype-converting form (fp32 NCHW -> compute dtype NHWC with zero channel padding)."""
    from msod_amd import ops
    x = _q(_rnd(2, 32, 9, 11, seed=80), dtype)
    w = _q(_rnd(16, 32, 3, 3, seed=81, scale=0.08), dtype)
    pk = ops.pack_conv(w, None, dtype, device=dev)
    y_nchw = ops.conv2d(x.to(dev).to(dtype), pk, 1)                      # contiguous NCHW in
    y_nhwc = ops.conv2d(to_dev_nhwc(x, dev, dtype), pk, 1)
    torch.cuda.synchronize()
    assert torch.equal(y_nchw.float().cpu(), y_nhwc.float().cpu())
    z = ops.to_nhwc(_rnd(2, 12, 5, 7, seed=82).to(dev), dtype, cpad=16)   # fp32 NCHW, 12 -> 16 channels
    torch.cuda.synchronize()
    assert z.shape == (2, 16, 5, 7) and z.stride(1) == 1
    assert torch.equal(z[:, :12].float().cpu(), _q(_rnd(2, 12, 5, 7, seed=82), dtype)) and z[:, 12:].abs().max() == 0


ALT_VARIANTS = [0, 27, 51]   # automatic choice / forced big tiles with the uniform-K-walk address path (UNIK) where the layer allows it,
# against variant 900 = the generic per-thread address path
STAGGERED_CASES = [
    # B, H, W, Cin, Cout, k, s, residual
    (2, 33, 31, 64, 320, 3, 1, True),       # ragged M, N tail over three / two tiles, border taps
    (3, 20, 20, 32, 256, 3, 2, False),      # Cin < K step: several taps per K tile (the non-branch-free advance), stride 2
    (1, 16, 16, 1024, 512, 1, 1, False),    # 1x1, long K (16 K tiles)
    (2, 12, 12, 128, 128, 3, 1, True),      # the 128-channel Bottleneck shape
    (1, 9, 11, 80, 160, 3, 1, False),       # K = 720 -> Kpad 768: K tail inside the last tile, taps straddle K tiles
    (1, 8, 8, 64, 8, 1, 1, False),          # ONE K tile (shorter than the prologue's run-ahead), N = 8
    (1, 12, 12, 256, 64, 3, 1, True),       # Cin >= 256: the chunk-major K walk (36 K steps), border taps on every side
    (2, 9, 9, 512, 40, 3, 2, False),        # chunk-major, stride 2, N tail inside a 64-wide tile
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("variant", ALT_VARIANTS)
@pytest.mark.parametrize("case", STAGGERED_CASES, ids=[f"s{i}" for i in range(len(STAGGERED_CASES))])
def test_alternative_gemm_variants_are_bit_identical(dev, dtype, variant, case):
    """The uniform-K-walk address path (constant per-thread pointers + scalar tap / chunk walk, taken automatically when
    Cin is a multiple of the K step) against the generic per-thread address path (variant 900): same fetches, same LDS
    image, same k order -> same bits; on shapes that stress K tails, taps that straddle K tiles (those stay on the generic
    path), 1 .. 72 K steps, the chunk-major walk, borders and tile edges."""
    from msod_amd import _lib, ops
    B, H, W, Cin, Cout, k, s_, use_res = case
    x = _q(_rnd(B, Cin, H, W, seed=91), dtype)
    w = _q(_rnd(Cout, Cin, k, k, seed=92, scale=1.0 / math.sqrt(Cin * k * k)), dtype)
    b = _rnd(Cout, seed=93, scale=0.5)
    pk = ops.pack_conv(w, b, dtype, s=s_, device=dev)
    xd = to_dev_nhwc(x, dev, dtype)
    lib = _lib.load()
    outs = {}
    for v in (900, variant):
        lib.cft_set_conv_variant(v)
        try:
            y0 = ops.conv2d(xd, pk, 1)
            res = y0.clone() if use_res else None
            outs[v] = ops.conv2d(xd, pk, 1, residual=res)
            torch.cuda.synchronize()
        finally:
            lib.cft_set_conv_variant(0)
    assert torch.equal(outs[900].float().cpu(), outs[variant].float().cpu())
    ref = F.silu(F.conv2d(x, w, b, s_, k // 2))
    got = to_cpu_f32(outs[variant])[:, :Cout]
    if not use_res:
        assert rel_err(got, ref) < tol(dtype)


ASM_CASES = [
    # B, H, W, Cin, Cout, k, s, residual      (the hand-scheduled 8-wave kernel takes 16-bit layers with Cin % 64 == 0)
    (2, 24, 24, 256, 512, 3, 1, True),      # chunk-major walk (36 K steps), two N tiles, border taps on every side, ragged M (1152 rows = 4.5 tiles)
    (1, 40, 40, 512, 256, 1, 1, False),     # pointwise: the unmasked form, 8 K steps
    (2, 33, 31, 64, 320, 3, 1, True),       # tap-major walk with ONE chunk per tap (9 steps: odd -> the zero padding step), N tail, ragged M
    (3, 20, 20, 128, 256, 3, 2, False),     # stride 2, tap-major with two chunks per tap
    (1, 16, 16, 1024, 512, 1, 1, False),    # long pointwise K (16 steps)
    (2, 9, 9, 512, 40, 3, 2, False),        # chunk-major, stride 2, N tail inside the first tile (72 steps)
    (1, 20, 20, 320, 256, 1, 1, True),      # pointwise with an odd step count (5): the masked form with the zero step
    (1, 7, 5, 128, 264, 1, 1, False),       # 35 rows: one ragged tile, two K steps (the shortest walk the kernel takes)
    (2, 13, 17, 64, 256, 5, 2, False),      # 5x5, stride 2: 25 taps (odd step count), border taps two pixels out (the descriptor's head room is pad-aware)
]


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "f16"])
@pytest.mark.parametrize("avariant", [96, 961, 962, 963, 964], ids=["h256", "h224", "h208", "h192", "h128"])
@pytest.mark.parametrize("case", ASM_CASES, ids=[f"a{i}" for i in range(len(ASM_CASES))])
def test_asm_gemm_kernel_is_bit_identical(dev, dtype, case, avariant):
    """The 8-wave kernel with the hand-scheduled (inline-asm) K loop (csrc/conv_gemm_asm.hip; variants 96 / 961-964 = wherever eligible with tiles
    of 256 / 224 / 208 / 192 / 128 rows - the automatic choice takes it for the wide layers with the height that fills the CUs' rounds) against
    the generic address path of the 16-wave kernel (variant 900): same fetches, same LDS image, same k order per accumulator -> the same bits,
    with and without a shortcut, and close to torch's fp32 convolution."""
    from msod_amd import _lib, ops
    B, H, W, Cin, Cout, k, s_, use_res = case
    x = _q(_rnd(B, Cin, H, W, seed=91), dtype)
    w = _q(_rnd(Cout, Cin, k, k, seed=92, scale=1.0 / math.sqrt(Cin * k * k)), dtype)
    b = _rnd(Cout, seed=93, scale=0.5)
    pk = ops.pack_conv(w, b, dtype, s=s_, device=dev)
    xd = to_dev_nhwc(x, dev, dtype)
    lib = _lib.load()
    outs = {}
    for v in (900, avariant):
        lib.cft_set_conv_variant(v)
        try:
            y0 = ops.conv2d(xd, pk, 1)
            res = y0.clone() if use_res else None
            outs[v] = (y0, ops.conv2d(xd, pk, 1, residual=res))
            torch.cuda.synchronize()
        finally:
            lib.cft_set_conv_variant(0)
    for a, c in zip(outs[900], outs[avariant]):
        assert torch.equal(a.float().cpu(), c.float().cpu())
    ref = F.silu(F.conv2d(x, w, b, s_, k // 2))
    assert rel_err(to_cpu_f32(outs[avariant][0])[:, :Cout], ref) < tol(dtype)


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "f16"])
@pytest.mark.parametrize("rows,K,N", [(8192, 1024, 4096), (1000, 4096, 1024), (8192, 512, 512)])
def test_asm_gemm_kernel_linear_forms(dev, dtype, rows, K, N):
    """The nn.Linear forms the CFT block runs on the asm kernel (reference models/common.py:511, :532-538): GELU, fp32 output added to the fp32
    residual stream, plain - bit-identical to the 16-wave kernel's generic path, many times over (a stale LDS tile would show as a flake)."""
    from msod_amd import _lib, ops
    x = _q(_rnd(rows, K, seed=5), dtype).to(dev).to(dtype)
    w = _q(_rnd(N, K, seed=6, scale=1.0 / math.sqrt(K)), dtype)
    pk = ops.pack_conv(w, _rnd(N, seed=7, scale=0.3), dtype, device=dev)
    resid = _rnd(rows, N, seed=8).to(dev)
    lib = _lib.load()
    outs = {}
    for v in (900, 96, 0, 964):
        lib.cft_set_conv_variant(v)
        try:
            o = [ops.linear(x, pk, ops.ACT_GELU), ops.linear(x, pk, ops.ACT_NONE, residual=resid, out_dtype=torch.float32), ops.linear(x, pk)]
            if v == 96:
                o += [ops.linear(x, pk) for _ in range(20)]
            torch.cuda.synchronize()
            outs[v] = o
        finally:
            lib.cft_set_conv_variant(0)
    for v in (96, 0, 964):                      # (0: the automatic choice - the tile height that fills the CUs' rounds)
        for a, c in zip(outs[900], outs[v][:3]):
            assert torch.equal(a.float().cpu(), c.float().cpu())
    for c in outs[96][3:]:
        assert torch.equal(c, outs[96][2])


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "f16"])
def test_probe_build_8wave_kernel_is_bit_identical(dev, dtype):
    """The 8-wave register-double-buffered 256x256 kernel (csrc/probes/conv_ring.hip, probe build only: it is slower than the shipped 16-wave
    kernel, profiles/r04_gemm_experiments.md) against the generic path of the same library, bit for bit, on the eligible cases above
    plus a wide layer with two N tiles.  Skipped when libcft_hip_probes.so has not been built (tools/build_probes.sh)."""
    import os
    from msod_amd import _lib, ops
    probes = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcft_hip_probes.so")
    if not os.path.exists(probes):
        pytest.skip("probe build absent")
    saved = (_lib._lib, _lib.LIB_PATH)
    _lib._lib, _lib.LIB_PATH = None, probes
    try:
        try:
            lib = _lib.load()
            stale = lib.cft_abi_version() != _lib.ABI_VERSION
        except AttributeError:          # built from older sources: an export is missing
            stale = True
        if stale:
            pytest.skip("probe build is older than the product library (re-run tools/build_probes.sh)")
        for case in [c for c in STAGGERED_CASES if c[3] % 64 == 0] + [(2, 24, 24, 256, 512, 3, 1, True), (1, 40, 40, 512, 256, 1, 1, False)]:
            B, H, W, Cin, Cout, k, s_, use_res = case
            x = _q(_rnd(B, Cin, H, W, seed=91), dtype)
            w = _q(_rnd(Cout, Cin, k, k, seed=92, scale=1.0 / math.sqrt(Cin * k * k)), dtype)
            pk = ops.pack_conv(w, _rnd(Cout, seed=93, scale=0.5), dtype, s=s_, device=dev)
            xd = to_dev_nhwc(x, dev, dtype)
            outs = {}
            for v in (900, 91):
                lib.cft_set_conv_variant(v)
                try:
                    y0 = ops.conv2d(xd, pk, 1)
                    outs[v] = ops.conv2d(xd, pk, 1, residual=y0.clone() if use_res else None)
                    torch.cuda.synchronize()
                finally:
                    lib.cft_set_conv_variant(0)
            assert torch.equal(outs[900], outs[91]), case
    finally:
        _lib._lib, _lib.LIB_PATH = saved


def test_clock_probe_reports_ticks_and_work(dev):
    """cft_clock_probe (bench.py's shader-clock reading): spins for the requested wall-clock time and reports shader-clock
    ticks, wall-clock ticks and the dependent FMAs executed; the shader clock it implies is a plausible GPU clock."""
    import ctypes
    from msod_amd import _lib
    lib = _lib.load()
    out = torch.zeros(4, dtype=torch.int64, device=dev)
    khz = ctypes.c_int(0)
    st = lib.cft_clock_probe(out.data_ptr(), 500, ctypes.byref(khz), torch.cuda.current_stream().cuda_stream)
    assert st == 0, lib.cft_last_error()
    torch.cuda.synchronize()
    ticks, wall, fmas, _ = out.tolist()
    assert khz.value > 0
    us = wall / (khz.value / 1e3)
    assert 500 <= us < 2000, us                      # at least the requested spin, not much more
    assert fmas >= 256 and ticks > 0
    mhz = ticks / us
    assert 50 <= mhz <= 4000, mhz                    # s_memtime: the shader clock or a constant reference clock, never garbage
    assert lib.cft_clock_probe(None, 500, ctypes.byref(khz), None) != 0 and b"cft_clock_probe" in lib.cft_last_error()
