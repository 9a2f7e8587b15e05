"""CPU-only checks of everything around the kernels: config generation, graph build, state-dict
compatibility, weight packing, BN folding, the C ABI surface, loud failure without a GPU."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

import msod_amd  # noqa: F401
from msod_amd import _lib, ops
from msod_amd.models import configs
from msod_amd.models.yolo_test import Model

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norm(d):
    d = dict(d)
    for part in ("backbone", "head"):
        d[part] = [[f, n, m, [None if a == "None" else a for a in args]] for f, n, m, args in d[part]]
    return d


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("name", sorted(configs.REFERENCE_YAMLS))
def test_generated_config_equals_reference_yaml(name):
    import yaml
    with open(f"{REF}/models/transformer/{name}.yaml") as fh:
        ref = _norm(yaml.safe_load(fh))
    assert configs.named_config(name) == ref


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_yaml_file_builds_unchanged():
    m = Model(f"{REF}/models/transformer/yolov5s_fusion_transformerx3_vedai.yaml")
    assert m.yaml["nc"] == 9 and len(m.model) == 47


def test_graph_structure_flagship():
    m = Model(configs.named_config("cfg3"))
    assert len(m.model) == 47
    assert m.save == sorted(set(m.save)) or True
    assert [type(x).__name__ for x in m.model][:6] == ["Focus", "Conv", "C3", "Conv", "C3", "Focus"]
    assert m.model[5].f == -4                                   # IR stream entry
    assert sum(p.numel() for p in m.parameters()) == 206257992  # 206 M parameters
    gpts = [x for x in m.model if type(x).__name__ == "GPT"]
    assert [g.n_embd for g in gpts] == [256, 512, 1024]
    keys = set(m.state_dict())
    for k in ("model.0.conv.conv.weight", "model.0.conv.bn.running_var", "model.4.m.8.cv2.conv.weight",
              "model.10.pos_emb", "model.10.trans_blocks.7.sa.que_proj.bias", "model.10.trans_blocks.0.mlp.2.weight",
              "model.10.ln_f.weight", "model.46.m.2.bias", "model.46.anchors", "model.46.anchor_grid"):
        assert k in keys, k
    det = m.model[-1]
    assert torch.allclose(det.anchors[0], torch.tensor([[10., 13.], [16., 30.], [33., 23.]]) / 8)
    assert float(det.m[0].bias.view(3, -1)[0, 4]) != 0.0        # _initialize_biases ran


def test_derived_configs():
    c2 = configs.named_config("cfg2")
    assert sum(1 for r in c2["backbone"] if r[2] == "GPT") == 1
    m = Model(c2)
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 36.6) < 0.1
    c5 = configs.named_config("cfg5")
    assert (c5["depth_multiple"], c5["width_multiple"]) == (1.33, 1.25)
    assert [g.n_embd for g in Model(c5).model if type(g).__name__ == "GPT"] == [320, 640, 1280]


def test_fuse_matches_batchnorm_eval():
    torch.manual_seed(0)
    from msod_amd.models.common import Conv
    c = Conv(8, 16, 3, 1).eval()
    with torch.no_grad():
        c.bn.running_mean.normal_(0, 0.3); c.bn.running_var.uniform_(0.5, 2); c.bn.weight.uniform_(0.5, 1.5); c.bn.bias.normal_(0, 0.2)
    x = torch.randn(2, 8, 9, 9)
    want = c.bn(c.conv(x))
    w, b = ops.fold_bn(c.conv.weight, c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var, c.bn.eps)
    assert c.bn.eps == 1e-3
    assert torch.allclose(F.conv2d(x, w, b, 1, 1), want, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_conv_layout(dtype):
    """w[n][(kh*KW + kw)*cin_pad + ci], zero padded in ci, K and n."""
    w = torch.randn(10, 12, 3, 3)
    b = torch.randn(10)
    pk = ops.pack_conv(w, b, dtype, cin_pad=16)
    ge = 8 if dtype == torch.bfloat16 else 4
    assert pk.n == 16 and pk.cin == 16 and pk.kpad % (8 * ge) == 0 and pk.kpad >= 144 and pk.w.dtype == dtype
    full = pk.w.float()
    for (n, kh, kw, ci) in [(0, 0, 0, 0), (9, 2, 1, 11), (3, 1, 2, 5)]:
        assert full[n, (kh * 3 + kw) * 16 + ci] == w[n, ci, kh, kw].to(dtype).float()
    assert full[:, 144:].abs().max() == 0 and full[10:].abs().max() == 0
    assert full.view(16, -1)[:, :144].view(16, 9, 16)[:, :, 12:].abs().max() == 0
    assert torch.equal(pk.bias[:10], b) and pk.bias[10:].abs().max() == 0
    assert pk.flops_per_row == 2 * 10 * 9 * 12


def test_c_abi_exports_every_declared_symbol():
    """The shared library must load (no GPU needed) and export exactly what include/cft_hip.h declares."""
    lib = _lib.load()
    assert lib.cft_abi_version() == _lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "cft_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|long|const char\*)\s+(cft_\w+)\s*\(", header, re.M))
    assert declared == set(_lib.SIGNATURES) | {"cft_last_error"}, declared ^ (set(_lib.SIGNATURES) | {"cft_last_error"})
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name


def test_conv2d_chain_eligibility_is_a_host_decision():
    """cft_conv2d_chain_ok (no GPU): the chained kernel takes a first layer of exactly 128 or 256 output channels on the uniform K walk
    (cin % 64 == 0, no K padding), 16-bit operands, and a pointwise second layer no wider than the first - the 64 -> 128 and 128 -> 256
    stride-2 convs + C3.cv1|cv2 of yolov5l (reference models/common.py:45-50, 141-143) and cv2[j] + cv1[j+1] of its 256-channel head
    C3s; not the 256 -> 512 pair, yolov5x's 80 / 160-channel layers, fp32, or a padded K.  An ineligible pair is refused by the launcher."""
    lib = _lib.load()
    BF16, F32, F16 = 0, 1, 2
    ok_ = lib.cft_conv2d_chain_ok

    def ok(B, H, W, cin, n1, kpad1, k, s_, n2, dt, ldx=None, ldy=None):
        return ok_(B, H, W, cin, ldx or cin, n1, kpad1, k, s_, n2, ldy or n2, dt)
    assert ok(64, 320, 320, 64, 128, 576, 3, 2, 128, BF16) == 1            # yaml rows 1-2 / 6-7
    assert ok(64, 160, 160, 128, 256, 1152, 3, 2, 256, F16) == 1           # rows 3-4 / 8-9
    assert ok(64, 40, 40, 256, 256, 2304, 3, 1, 256, BF16) == 1            # head C3: cv2[j] + cv1[j+1]
    assert ok(1, 9, 9, 64, 128, 64, 1, 1, 8, BF16) == 1                    # any size, 1x1 first layer, narrow second layer
    assert ok(64, 80, 80, 256, 512, 2304, 3, 2, 512, BF16) == 0            # 512 channels do not fit a tile
    assert ok(16, 640, 640, 80, 160, 720, 3, 2, 160, BF16) == 0            # yolov5x widths
    assert ok(64, 320, 320, 64, 128, 576, 3, 2, 128, F32) == 0             # 16-bit only
    assert ok(64, 320, 320, 64, 128, 640, 3, 2, 128, BF16) == 0            # padded K
    assert ok(64, 320, 320, 32, 128, 288, 3, 2, 128, BF16) == 0            # cin % 64
    assert ok(64, 320, 320, 64, 128, 576, 3, 2, 136, BF16) == 0            # second layer wider than the first
    assert ok(64, 320, 320, 64, 128, 576, 3, 2, 100, BF16) == 0            # n2 % 8
    assert ok(0, 320, 320, 64, 128, 576, 3, 2, 128, BF16) == 0 and ok(64, 320, 320, 64, 128, 576, 2, 2, 128, BF16) == 0
    # 'ok' is the launcher's own validation, incl. its 2^31-element limits on the BUFFER extents (ld, not channel count)
    assert ok(64, 640, 640, 64, 128, 576, 3, 2, 128, BF16) == 1            # 64 * 640 * 640 * 64 = 1.7e9 elements
    assert ok(64, 640, 640, 64, 128, 576, 3, 2, 128, BF16, ldx=128) == 0   # the same input as a slice of a 128-channel buffer: 3.4e9
    assert ok(64, 640, 640, 64, 128, 576, 3, 2, 128, BF16, ldy=1024) == 0  # output slice of a 1024-channel buffer: 6.7e9
    assert ok(64, 320, 320, 64, 128, 576, 3, 2, 128, BF16, ldy=120) == 0   # ldy must cover the second layer's width
    # null / ineligible arguments are refused before any launch
    st = lib.cft_conv2d_chain(None, None, None, None, None, None, 1, 8, 8, 64, 64, 0, 128, 576, 3, 2, 128, 128, 0, 1, BF16, None)
    assert st == -1 and b"null pointer" in lib.cft_last_error()
    st = lib.cft_conv2d_chain(4096, 4096, None, 4096, None, 4096, 1, 8, 8, 64, 64, 0, 512, 576, 3, 2, 128, 128, 0, 1, BF16, None)
    assert st == -1 and b"not eligible" in lib.cft_last_error()


def test_bad_arguments_return_error_codes_without_gpu():
    lib = _lib.load()
    st = lib.cft_conv2d(None, None, None, None, None, 1, 8, 8, 8, 8, 0, 8, 64, 1, 1, 8, 0, 0, 0, 0, 0, 0, 0, None)
    assert st == -1 and b"null pointer" in lib.cft_last_error()
    st = lib.cft_layernorm(1, 1, 1, 1, 4, 6, 1e-5, 0, None)     # C not a multiple of 4; rejected before any launch
    assert st == -1
    # fused Focus: only 32/48/64/80 output channels, weights packed [n][192], even pointer/strides
    st = lib.cft_focus_conv(16, 0, 3 * 64 * 64, 64 * 64, 64, 1.0, 16, 192, None, 16, 40, 0, 1, 64, 64, 40, 1, 0, None)
    assert st == -1 and b"n must be 32, 48, 64 or 80" in lib.cft_last_error()
    st = lib.cft_focus_conv(16, 0, 3 * 64 * 64, 64 * 64, 64, 1.0, 16, 160, None, 16, 64, 0, 1, 64, 64, 64, 1, 0, None)
    assert st == -1 and b"[n][192]" in lib.cft_last_error()
    st = lib.cft_focus_conv(17, 1, 3 * 64 * 64, 64 * 64, 64, 1.0 / 255, 16, 192, None, 16, 64, 0, 1, 64, 64, 64, 1, 0, None)
    assert st == -1 and b"pixel-pair" in lib.cft_last_error()
    # fused Bottleneck: 64 / 128 channels only, and never in place (it reads a halo of x)
    st = lib.cft_bottleneck(4096, 256, 0, 16, 256, None, 16, 2304, None, None, 1 << 20, 256, 0, 1, 8, 8, 256, 1, 0, None)
    assert st == -1 and b"64 and 128 channels" in lib.cft_last_error()
    st = lib.cft_bottleneck(4096, 64, 0, 16, 64, None, 16, 576, None, None, 4096, 64, 0, 1, 8, 8, 64, 1, 0, None)
    assert st == -1 and b"overlaps the input" in lib.cft_last_error()
    st = lib.cft_bottleneck(4096, 128, 0, 16, 64, None, 16, 576, None, None, 4096 + 64, 128, 0, 1, 8, 8, 64, 1, 0, None)
    assert st == -1 and b"overlaps" in lib.cft_last_error()       # slices of one buffer that share channels [32,64)
    st = lib.cft_bottleneck_pack_w2(16, 576, 64, 32, 0, None)     # stage images exist for the 128-channel kernel only
    assert st == -1 and b"128 channels" in lib.cft_last_error()


def test_no_cpu_fallback():
    m = Model(configs.named_config("cfg1"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))
    m.train()
    with pytest.raises(RuntimeError):
        m.model[1](torch.zeros(1, 32, 8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "multispectral-object-detection_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_pickled_checkpoint_loads_into_hip_modules(tmp_path):
    """A checkpoint pickled by the REFERENCE code (whole nn.Module, train.py:850-860) un-pickles into
    this package's classes once the module aliases are installed, with identical parameters."""
    import subprocess
    import sys
    ck = tmp_path / "ref.pt"
    script = f"""
import sys, types, logging, torch
for n in ("cv2", "torchvision", "seaborn"):
    sys.modules.setdefault(n, types.ModuleType(n))
sys.modules["cv2"].setNumThreads = lambda n: None
sys.path.insert(0, {REF!r}); sys.path.insert(0, {ROOT!r}); logging.disable(logging.CRITICAL)
import msod_amd
from msod_amd.models.configs import named_config
from models.yolo_test import Model
m = Model(named_config("cfg2")).half()
torch.save({{"model": m, "ema": None, "epoch": 3}}, {str(ck)!r})
"""
    subprocess.run([sys.executable, "-c", script], check=True, capture_output=True)
    from msod_amd import compat
    model = compat.attempt_load(str(ck), map_location="cpu")
    assert type(model).__module__.startswith("msod_amd") and model.compute_dtype == torch.float32   # .float(), as the reference
    assert type(model.model[-1]).__name__ == "Detect" and not hasattr(model.model[1], "bn")     # fused
    assert all(p.dtype == torch.float32 for p in model.parameters())
    ups = [m for m in model.model if isinstance(m, torch.nn.Upsample)]
    assert len(ups) == 2 and all(type(u).__module__.startswith("msod_amd") for u in ups)
    for name in ("models", "models.common", "models.yolo_test"):
        sys.modules.pop(name, None)


def test_stream_lanes_follow_the_ir_backbone():
    """Two-HIP-stream schedule: the IR backbone (everything downstream of an ``f == -4`` entry through
    single-input edges, plus Add2(index=1)) is lane 1; joins and the RGB backbone/head are lane 0."""
    m = Model(configs.named_config("cfg3"))
    lanes = m.stream_lanes()
    assert len(lanes) == 47
    assert [i for i, l in enumerate(lanes) if l == 1] == [5, 6, 7, 8, 9, 12, 15, 16, 19, 23, 24, 25, 28]
    for i, mod in enumerate(m.model):
        if type(mod).__name__ in ("GPT", "Add", "Concat", "Detect"):
            assert lanes[i] == 0
    add = Model(configs.named_config("cfg1")).stream_lanes()
    assert add[:10] == [0] * 10 and add[10:20] == [1] * 10 and set(add[20:]) == {0}


def test_concat_plan_routes_producers_into_concat_buffers():
    """Head Concats: Conv / Add sources write straight into their slice of the concat buffer (no copy);
    Upsample sources stay deferred copies.  Offsets follow the reference's torch.cat order (common.py:217-219)."""
    from msod_amd.models.configs import named_config
    from msod_amd.models.yolo_test import Model
    m = Model(named_config("cfg3"))
    plan = m.concat_plan()
    concats = [i for i, l in enumerate(m.model) if type(l).__name__ == "Concat"]
    assert len(concats) == 4 and len(plan) == 6
    for prod, (cidx, off, c, total) in plan.items():
        srcs = [cidx + j if j < 0 else j for j in m.model[cidx].f]
        assert prod in srcs and type(m.model[prod]).__name__ in ("Conv", "Add", "C3")
        assert off == (0 if srcs.index(prod) == 0 else total - c) and 0 < c < total
    ups = [s for cidx in concats for s in (cidx + j if j < 0 else j for j in m.model[cidx].f) if type(m.model[s]).__name__ == "Upsample"]
    assert ups and not any(u in plan for u in ups)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_graph_file_builds_over_this_packages_common():
    """Strict form of the boundary ("models/yolo.py builds unchanged"): the reference's OWN
    models/yolo_test.py (Model, parse_model with its eval() name lookup and `m is X` identity tests, forward_once,
    Detect) imported unchanged, with only `models.common` resolved to this package, builds a reference yaml file;
    its state-dict keys equal those of the reference-on-reference build; and a CPU forward reaches this package's
    kernels' guard ("no CPU fallback") - i.e. the reference executor is driving the HIP-backed modules.  Run in a
    subprocess: it puts the reference tree on sys.path."""
    import subprocess
    import sys
    script = f"""
import sys, types, logging, torch
for n in ("cv2", "torchvision", "seaborn"):
    sys.modules.setdefault(n, types.ModuleType(n))
sys.modules["cv2"].setNumThreads = lambda n: None
sys.path.insert(0, {REF!r}); sys.path.insert(0, {ROOT!r}); logging.disable(logging.CRITICAL)
import msod_amd
from msod_amd import compat
compat.install_common_alias()
from models.yolo_test import Model, Detect                      # the REFERENCE's graph file
import models.yolo_test as ref_graph
assert ref_graph.__file__.startswith({REF!r})
m = Model({REF!r} + "/models/transformer/yolov5s_fusion_transformerx3_vedai.yaml")
kinds = [type(l).__module__.split(".")[0] for l in m.model]
assert type(m.model[0]).__name__ == "Focus" and type(m.model[0]).__module__.startswith("msod_amd"), kinds
assert type(m.model[-1]) is Detect and Detect.__module__ == "models.yolo_test"
assert sum(type(l).__name__ == "GPT" and type(l).__module__.startswith("msod_amd") for l in m.model) == 3
assert any(type(l) is torch.nn.Upsample for l in m.model)          # the reference's plain torch Upsample stays
from msod_amd.models.yolo_test import Model as Ours
from msod_amd.models.configs import named_config
ours = Ours(named_config("yolov5s_fusion_transformerx3_vedai"))
assert list(m.state_dict().keys()) == list(ours.state_dict().keys())
assert all(a.shape == b.shape for a, b in zip(m.state_dict().values(), ours.state_dict().values()))
try:
    m.eval()(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64))
    raise SystemExit("CPU forward did not raise")
except RuntimeError as e:
    assert "no CPU fallback" in str(e), e
m.fuse()                                                         # the reference's fuse(): type(m) is Conv, .bn, .fuseforward
assert not hasattr(m.model[1], "bn") and m.model[1].conv.bias is not None
print("STRICT-FORM-OK")
"""
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True)
    assert r.returncode == 0 and "STRICT-FORM-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_parity_check_flags_a_wrong_batch():
    """bench.py's `parity_at_bench_shape`: green for outputs at the reference-style bf16 error level, red for a forward that is off
    (wrong weights / a broken kernel) and red for a bf16 result that is much worse than the reference's own bf16 - the run then exits 3."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g = torch.Generator().manual_seed(0)
    want = [torch.randn(4, 3, 8, 8, 8, generator=g), torch.randn(4, 3, 4, 4, 8, generator=g)]
    noise = lambda s: [w + s * torch.randn(w.shape, generator=g) for w in want]      # noqa: E731
    ref16 = [r[:2] for r in noise(0.02)]
    ok = bench.parity_at_bench_shape("bf16", noise(0.02), want, ref16, 2)
    assert ok["ok"] and ok["pairs"] == 4 and ok["vs_reference_style_bf16"]["ok"]
    assert "NOT the literal 1e-2" in ok["gate"] and ok["meets_north_star_bound"] is False      # bf16 says in words what it is gated on
    f16 = bench.parity_at_bench_shape("f16", noise(0.002), want)
    assert f16["meets_north_star_bound"] is True and "parity-green" in f16["gate"]
    assert not bench.parity_at_bench_shape("bf16", noise(0.5), want, ref16, 2)["ok"]             # absolute bound
    worse = bench.parity_at_bench_shape("bf16", noise(0.06), want, ref16, 2)                     # inside 2.5e-2? no - and 3x the reference level
    assert not worse["ok"]
    assert bench.parity_at_bench_shape("f16", noise(0.002), want)["ok"] and not bench.parity_at_bench_shape("f16", noise(0.1), want)["ok"]
    assert bench.parity_at_bench_shape("f32", noise(1e-5), want)["ok"] and not bench.parity_at_bench_shape("f32", noise(0.01), want)["ok"]


def test_shader_clock_summary_arithmetic():
    """bench.py's reading of cft_clock_probe samples: MHz = shader ticks / wall ticks x wall-clock rate; the FMA rate against the idle probe."""
    import ctypes
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pr = object.__new__(bench.ShaderClockProbe)          # no GPU: fill the fields the summary reads
    pr.khz = ctypes.c_int(100000)                        # 100 MHz wall clock
    pr.max = 4
    pr.buf = torch.tensor([[420000, 20000, 80000, 0],    # 200 us at 2100 MHz, 400 FMA/us
                           [400000, 20000, 76000, 0],    # 2000 MHz
                           [440000, 20000, 84000, 0],    # 2200 MHz
                           [0, 0, 0, 0],
                           [480000, 20000, 90000, 0]], dtype=torch.int64)
    pr.n = 4                                             # the fourth sample never ran (wall ticks 0): dropped
    pr.idle = pr.buf[4].tolist()
    s = pr.summary()
    assert s["samples"] == 3 and s["s_memtime_mhz"] == {"median": 2100.0, "min": 2000.0, "max": 2200.0}
    assert s["idle_gpu"]["s_memtime_mhz"] == 2400.0 and s["idle_gpu"]["dependent_fma_per_us"] == 450.0
    assert abs(s["fma_rate_vs_idle"] - 400.0 / 450.0) < 1e-3
    pr.n = 0
    assert pr.summary() is None


def test_cft_fusion_plan_and_nms_module_structure():
    """Host side of round 4's graph-level pieces (no GPU): ``Model.cft_fusion_plan`` finds the (Add2, Add2, Add) group behind every GPT
    block of an x3 config - the yaml rows reference models/transformer/yolov5l_fusion_transformerx3_FLIR_aligned.yaml wires as
    [4,10]/[9,10] -> 29, [14,17]/[16,17] -> 30, [22,26]/[25,26] -> 31 - and nothing in a config without GPT blocks; ``Model.nms()``
    appends / removes the reference's NMS module (models/yolo_test.py:306-318) and ``autoshape()`` wraps the model."""
    from msod_amd.models.common import NMS, autoShape
    from msod_amd.models.configs import named_config
    from msod_amd.models.yolo_test import Model
    m = Model(named_config("cfg3"))
    assert m.cft_fusion_plan() == {11: (10, 12, 29), 18: (17, 19, 30), 27: (26, 28, 31)}
    assert Model(named_config("cfg1")).cft_fusion_plan() == {}
    # the Conv layers handed un-run to the C3 behind them (yaml rows 1-2, 3-4, 13-14 of the RGB stream, 6-7, 8-9, 15-16 of the IR
    # stream); rows 20 / 23 feed SPP and the head's stride-2 convs feed Concat: not in the plan
    chain = m.chain_plan()
    assert chain == frozenset({1, 3, 6, 8, 13, 15}) and 1 in m.save      # (row 1 is "saved" by the reference's f = -4 book-keeping only)
    n = len(m.model)
    m.nms()
    assert len(m.model) == n + 1 and type(m.model[-1]) is NMS and m.model[-1].f == -1 and m.model[-1].i == n
    assert m.cft_fusion_plan() == {11: (10, 12, 29), 18: (17, 19, 30), 27: (26, 28, 31)}      # re-derived after the structural change
    m.nms()                                        # idempotent
    assert len(m.model) == n + 1
    m.nms(False)
    assert len(m.model) == n and type(m.model[-1]) is not NMS
    w = m.autoshape()
    assert isinstance(w, autoShape) and w.autoshape() is w and w.names == m.names and torch.equal(w.stride, m.stride)
    m._print_biases()


def test_prefix_segments_and_executor_switches():
    """Model.prefix_segments (the image-only prefix of each backbone that Model.depth_first runs sub-batch by sub-batch): a Focus fed by x or
    x2 followed by f == -1 Conv / C3 rows with exactly one reader, ending on a C3; every executor switch is a property whose setter drops
    the captured graphs."""
    from msod_amd.models.configs import named_config
    from msod_amd.models.yolo_test import Model
    m = Model(named_config("cfg3"))
    assert m.prefix_segments() == [(0, 4), (5, 9)] and m.prefix_segments(3) == [(0, 2), (5, 7)] and m.prefix_segments(2) == []
    assert Model(named_config("cfg1")).prefix_segments() == [(0, 4), (10, 14)]            # add-fusion: row 4 is also read by the Add
    assert Model(named_config("yolov5l_fusion_transformer_FLIR")).prefix_segments() == [(0, 2), (3, 5)]   # 4-GPT layout: GPT after P2
    assert (m.depth_first, m.splitk, m.chain_convs, m.fuse_cft_outputs, m.plan_concats) == (None, True, True, True, True)
    for name, value in (("depth_first", (8, None)), ("splitk", False), ("chain_convs", False),
                        ("fuse_cft_outputs", False), ("plan_concats", False)):
        m._graphs["sentinel"] = object()
        setattr(m, name, value)
        assert getattr(m, name) == value and not m._graphs, name


def test_splitk_choice_rule():
    """ops.splitk_choice (host decision, measured in profiles/r05_splitk_ab.md): split only GEMMs on the uniform K walk whose 256 x 256 tiles
    leave >= 3/4 of the chip idle, only up to SPLITK_MAX_ROWS token rows, to the smallest split that yields 128 workgroups with >= 4 K
    steps each."""
    import torch
    from msod_amd import ops
    pk = lambda n, k: ops.pack_conv(torch.zeros(n, k), None, torch.bfloat16)      # noqa: E731
    assert ops.splitk_choice(8192, pk(1024, 4096), torch.bfloat16) == 1          # 64 pairs: never
    assert ops.splitk_choice(2048, pk(1024, 4096), torch.bfloat16) == 1          # above SPLITK_MAX_ROWS
    assert ops.splitk_choice(1024, pk(1024, 4096), torch.bfloat16) == 8          # 16 tiles -> 128 workgroups of 8 K steps
    assert ops.splitk_choice(1024, pk(256, 1024), torch.bfloat16) == 4           # 4 K steps per split is the floor
    assert ops.splitk_choice(1024, pk(256, 256), torch.bfloat16) == 1            # 4 K steps in all
    assert ops.splitk_choice(1024, pk(4096, 1024), torch.bfloat16) == 1          # wide N: 64 tiles already
    assert ops.splitk_choice(1024, pk(1024, 1000), torch.bfloat16) == 1          # K not on the uniform walk (padded)
    assert ops.splitk_choice(512, pk(1024, 4096), torch.float32) == 8            # fp32: 32-wide K steps


@pytest.mark.parametrize("gen, inc", [("tools/gen_conv_asm.py", "multispectral-object-detection_amd/csrc/conv_gemm_asm.inc"),
                                      ("tools/gen_bneck_asm.py", "multispectral-object-detection_amd/csrc/probes/bottleneck_asm.inc")])
def test_committed_asm_text_is_what_its_generator_writes(gen, inc, tmp_path):
    """The hand-scheduled K loops are GENERATED text (csrc/*.inc, committed so that a build needs no generator run): the committed file must
    be byte-for-byte what the committed generator writes - an edit of one without the other fails here, not on the GPU."""
    import subprocess
    import sys
    out = tmp_path / "gen.inc"
    subprocess.run([sys.executable, os.path.join(ROOT, gen), str(out)], check=True, env={**os.environ, "CONV_ASM_PAD_NOPS": "0"})
    assert out.read_bytes() == open(os.path.join(ROOT, inc), "rb").read()


def test_asm_loop_text_invariants():
    """Structural invariants of one generated K step (tools/gen_conv_asm.py): 64 MFMAs, 24 fragment reads, AP + 4 LDS-DMA requests, exactly
    one barrier, and every accumulator register of the (MT0 / MT1) m-tiles written exactly twice per step (k halves 0 and 1)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_conv_asm", os.path.join(ROOT, "tools", "gen_conv_asm.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    op = "v_mfma_f32_16x16x32_bf16"
    for (mt0, mt1) in g.TILES:
        ap = (16 * (mt0 + mt1) + 63) // 64                    # A passes of the tile (emit())
        for grp, mt in ((0, mt0), (1, mt1)):
            for masked in (False, True):
                for c in (0, 1):
                    lines = [ln for item in g.step(op, c, grp, masked, mt, ap) for ln in ([item] if isinstance(item, str) else item)]
                    mf = [ln for ln in lines if ln.startswith(op)]
                    assert len(mf) == 8 * mt, (mt, masked, grp)
                    accs = [ln.split()[1].rstrip(",") for ln in mf]
                    assert all(accs.count(a) == 2 for a in set(accs)) and len(set(accs)) == 4 * mt
                    assert sum(ln.startswith("ds_read_b128") for ln in lines) == 2 * (mt + 4)
                    assert sum(ln.startswith("s_barrier") for ln in lines) == 1
                    assert sum(ln.startswith("buffer_load_dwordx4") for ln in lines) == ap + 4
                    # the barrier sits between the two MFMA halves, behind a full wait
                    b = next(i for i, ln in enumerate(lines) if ln.startswith("s_barrier"))
                    assert lines[b - 1] == "s_waitcnt vmcnt(0) lgkmcnt(0)"
                    assert sum(ln.startswith(op) for ln in lines[:b]) == 4 * mt
