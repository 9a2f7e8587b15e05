"""End-to-end parity of the HIP model against the CPU oracle (same seeded weights and inputs) and
the committed golden vectors, at the golden shapes AND at the BASELINE shapes, plus size-independent
properties.

Tolerances (BASELINE.json north_star: 1e-3 fp32 / 1e-2 for the 16-bit path; SURVEY.md D8: an absolute
bound is only meaningful in sigmoid space, pixel-space columns get a relative one):

  fp32   raw head logits |err| <= 1e-3 absolute; decoded pred allclose(rtol=1e-3, atol=1e-3) - vs the oracle
         AND vs the reference's own outputs (tests/golden/*.pt)
  fp16   sigmoid-space max-abs (conf/cls and sigma(box logits)) <= 1e-2 vs the fp32 oracle.  fp16 is the
         precision the reference itself runs on a GPU (test.py:66-68 `model.half()`).
  bf16   (a) the HIP error is what bf16 STORAGE predicts: oracle/lowp_oracle.py evaluates the fp32 oracle with
         a bf16 rounding wherever the product stores a bf16 tensor; the HIP result must be as close to fp32
         as that model is - rms error <= 1.25 x the model's, max error <= 1.75 x the model's (measured on
         MI355X: rms ratio 0.89-1.0, max ratio 0.9-1.45; two different bf16 realisations of a 100-layer
         network are themselves ~1e-2 apart, roundings flip and cascade, so the comparison is of error
         LEVELS, not element by element);
         (b) vs the fp32 oracle <= BF16_SIGMOID_ATOL = 2.5e-2.  bf16 keeps 8 significand bits; ~100 stored
         layers deep, ANY implementation with bf16 storage is 0.9-2.0e-2 away from fp32 on these
         deliberately lively weights (storage model: 0.85-1.98e-2), so 1e-2 is not attainable in bf16 -
         fp16 is the 16-bit type that meets it.
"""
import glob
import os

import pytest
import torch

from test_oracle_golden import load_case

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
GOLDEN = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "*.pt"))
                if not os.path.basename(p).startswith(("nms_", "ref_ckpt", "s_x3_train", "letterbox_", "lowp_", "survey_", "ladder_")))
IDS = [os.path.basename(p)[:-3] for p in GOLDEN]

F16_SIGMOID_ATOL = 1e-2     # north_star's 16-bit bound, met in fp16 (measured <= 2e-3)
F16_LOGIT_RMS = 5e-3
BF16_SIGMOID_ATOL = 2.5e-2  # bf16 storage floor is 1.0-1.5e-2 on these weights (see module docstring)
BF16_LOGIT_RMS = 3e-2
BF16_RMS_RATIO, BF16_MAX_RATIO = 1.25, 1.75   # HIP bf16 error vs the error of the bf16-storage model of the oracle


_ORACLE_CACHE = {}


def _oracle_for(path, kind="fp32"):
    """The CPU oracle's (pred, raw) for a golden case, computed once per session and kind ("fp32" = OracleModel, "lowp_bf16" = the bf16
    storage model): the three precision tests of a case share it (the CPU work dominates the suite's wall time)."""
    key = (path, kind)
    if key not in _ORACLE_CACHE:
        from oracle.cft_oracle import OracleModel
        from oracle.lowp_oracle import LowpOracle
        g, cfg, model, rgb, ir = load_case(path)
        sd = model.state_dict()
        _ORACLE_CACHE[key] = OracleModel(cfg)(sd, rgb, ir) if kind == "fp32" else LowpOracle(cfg, torch.bfloat16)(sd, rgb, ir)
    return _ORACLE_CACHE[key]


def _run(model, rgb, ir, dev, dtype):
    model = model.to(dev).set_compute_dtype(dtype)
    with torch.no_grad():
        pred, raw = model(rgb.to(dev), ir.to(dev))
    torch.cuda.synchronize()
    return pred.cpu(), [r.cpu() for r in raw]


def _flat(raws):
    return torch.cat([r.reshape(-1) for r in raws])


def _sig_err(raw, want_raw):
    return (_flat(raw).sigmoid() - _flat(want_raw).sigmoid()).abs().max().item()


def _rms_rel(raw, want_raw):
    a, b = _flat(raw), _flat(want_raw)
    return ((a - b).pow(2).mean().sqrt() / b.std()).item()


def _check_fp32(pred, raw, want_pred, want_raw):
    for a, b in zip(raw, want_raw):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 1e-3
    assert torch.allclose(pred, want_pred, rtol=1e-3, atol=1e-3)


def _check_f16(pred, raw, want_pred, want_raw):
    assert _sig_err(raw, want_raw) <= F16_SIGMOID_ATOL
    assert _rms_rel(raw, want_raw) <= F16_LOGIT_RMS
    assert (pred[..., 4:] - want_pred[..., 4:]).abs().max().item() <= F16_SIGMOID_ATOL
    assert torch.allclose(pred[..., :4], want_pred[..., :4], rtol=2e-2, atol=1.0)   # pixel columns: relative + 1 px


def _check_bf16(pred, raw, want_pred, want_raw, lowp_raw=None):
    assert _sig_err(raw, want_raw) <= BF16_SIGMOID_ATOL
    assert _rms_rel(raw, want_raw) <= BF16_LOGIT_RMS
    assert (pred[..., 4:] - want_pred[..., 4:]).abs().max().item() <= BF16_SIGMOID_ATOL
    if lowp_raw is not None:      # the error LEVEL is the one bf16 storage predicts
        assert _rms_rel(raw, want_raw) <= BF16_RMS_RATIO * _rms_rel(lowp_raw, want_raw) + 5e-4
        assert _sig_err(raw, want_raw) <= BF16_MAX_RATIO * _sig_err(lowp_raw, want_raw) + 1e-3


# ------------------------------------------------------------------------- golden shapes, every config
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_fp32_matches_oracle_and_golden(dev, path):
    g, cfg, model, rgb, ir = load_case(path)
    want_pred, want_raw = _oracle_for(path)
    pred, raw = _run(model, rgb, ir, dev, torch.float32)
    _check_fp32(pred, raw, want_pred, want_raw)
    _check_fp32(pred, raw, g["pred"], g["raw"])            # and against the reference's own output


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_f16_matches_oracle_within_1e2(dev, path):
    g, cfg, model, rgb, ir = load_case(path)
    want_pred, want_raw = _oracle_for(path)
    pred, raw = _run(model, rgb, ir, dev, torch.float16)
    _check_f16(pred, raw, want_pred, want_raw)
    _check_f16(pred, raw, g["pred"], g["raw"])


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_bf16_matches_its_storage_model_and_oracle(dev, path):
    g, cfg, model, rgb, ir = load_case(path)
    want_pred, want_raw = _oracle_for(path)
    _, lowp_raw = _oracle_for(path, "lowp_bf16")
    pred, raw = _run(model, rgb, ir, dev, torch.bfloat16)
    _check_bf16(pred, raw, want_pred, want_raw, lowp_raw)


# ------------------------------------------------------------------------- BASELINE shapes
def _seeded(cfg_name, seed):
    from msod_amd.models.configs import named_config
    from msod_amd.models.yolo_test import Model
    from msod_amd.utils.seeded import seeded_state_dict
    cfg = named_config(cfg_name)
    model = Model(cfg)
    sd = seeded_state_dict(model.state_dict(), seed)
    model.load_state_dict(sd)
    return cfg, model, sd


def test_cfg2_full_shape_fp32_tolerance_check(dev):
    """BASELINE config 2 as specified: yolov5s + 1 CFT block, 640x640, batch 16, fp32, every pair vs the oracle."""
    from msod_amd.utils.seeded import seeded_inputs
    from oracle.cft_oracle import OracleModel
    cfg, model, sd = _seeded("cfg2", 42)
    rgb, ir = seeded_inputs(16, 640, 640, 42)
    want_pred, want_raw = OracleModel(cfg)(sd, rgb, ir)
    pred, raw = _run(model.fuse(), rgb, ir, dev, torch.float32)
    assert pred.shape == (16, 25200, 14)
    _check_fp32(pred, raw, want_pred, want_raw)


def test_cfg3_full_shape_all_precisions(dev):
    """BASELINE config 3 network at its own shape (yolov5l + CFTx3 FLIR, 640x640): two pairs, fp32 / fp16 / bf16
    vs the oracle; the 16-bit runs use the deployed form (BN folded) like bench.py."""
    from msod_amd.utils.seeded import seeded_inputs
    from oracle.cft_oracle import OracleModel
    from oracle.lowp_oracle import LowpOracle
    cfg, model, sd = _seeded("cfg3", 0)
    rgb, ir = seeded_inputs(2, 640, 640, 0)
    want_pred, want_raw = OracleModel(cfg)(sd, rgb, ir)
    model.fuse()
    pred, raw = _run(model, rgb, ir, dev, torch.float32)
    assert pred.shape == (2, 25200, 8)
    _check_fp32(pred, raw, want_pred, want_raw)
    pred, raw = _run(model, rgb, ir, dev, torch.float16)
    _check_f16(pred, raw, want_pred, want_raw)
    _, lowp_raw = LowpOracle(cfg, torch.bfloat16)(sd, rgb[:1], ir[:1])
    pred, raw = _run(model, rgb, ir, dev, torch.bfloat16)
    _check_bf16(pred, raw, want_pred, want_raw)
    _check_bf16(pred[:1], [r[:1] for r in raw], want_pred[:1], [r[:1] for r in want_raw], lowp_raw)


def test_cfg5_one_pair_at_1280(dev):
    """BASELINE config 5 network (yolov5x x3 CFT, 80/160/320/640/1280 channels, head widths 40/80/160) at
    1280x1280, one pair: fp32 and fp16 vs the oracle."""
    from msod_amd.utils.seeded import seeded_inputs
    from oracle.cft_oracle import OracleModel
    cfg, model, sd = _seeded("cfg5", 5)
    rgb, ir = seeded_inputs(1, 1280, 1280, 5)
    want_pred, want_raw = OracleModel(cfg)(sd, rgb, ir)
    model.fuse()
    pred, raw = _run(model, rgb, ir, dev, torch.float32)
    assert pred.shape == (1, 100800, 8)
    _check_fp32(pred, raw, want_pred, want_raw)
    pred, raw = _run(model, rgb, ir, dev, torch.float16)
    _check_f16(pred, raw, want_pred, want_raw)


# ------------------------------------------------------------------------- the 16-bit bound, pinned to the REFERENCE's own bf16 forward
# tests/golden/lowp_ref.pt (make_golden.py `lowp`): the reference's unmodified Model under torch.autocast("cpu", bfloat16) on
# the golden weights/inputs, plus four cases with the reference CONSTRUCTOR's weight distributions (`dinit_*`).
LOWP_REF = torch.load(os.path.join(HERE, "golden", "lowp_ref.pt"), weights_only=False)
REF_BF16_RMS_RATIO, REF_BF16_MAX_RATIO = 1.10, 1.30   # HIP bf16 error level vs the reference's own bf16 error level


def _lowp_case(name):
    from msod_amd.models.configs import named_config
    from msod_amd.models.yolo_test import Model
    from msod_amd.utils.seeded import default_init_state_dict, seeded_inputs, seeded_state_dict
    rec = LOWP_REF[name]
    c = rec["case"]
    cfg = named_config(c["cfg"])
    model = Model(cfg)
    model.load_state_dict((default_init_state_dict if c["dinit"] else seeded_state_dict)(model.state_dict(), c["seed"]))
    if c["fused"]:
        model.fuse()
    rgb, ir = seeded_inputs(c["batch"], c["height"], c["width"], c["seed"])
    return rec, cfg, model, rgb, ir


@pytest.mark.parametrize("name", [n for n in LOWP_REF if not n.startswith("dinit_")])
def test_bf16_is_at_least_as_close_to_fp32_as_the_references_own_bf16(dev, name):
    """HIP bf16 vs the reference's fp32 output, against the reference's OWN bf16-autocast forward vs the same fp32
    output (recorded in the build container): rms error <= 1.10 x, max sigmoid-space error <= 1.30 x (+1e-3) the
    reference's.  (Two bf16 realisations differ element by element - roundings flip and cascade - so the comparison is
    of error levels.)"""
    rec, cfg, model, rgb, ir = _lowp_case(name)
    g = torch.load(os.path.join(HERE, "golden", name + ".pt"), weights_only=False)
    ref16 = [r.float() for r in rec["raw_bf16"]]
    pred, raw = _run(model, rgb, ir, dev, torch.bfloat16)
    assert _rms_rel(raw, g["raw"]) <= REF_BF16_RMS_RATIO * _rms_rel(ref16, g["raw"]) + 2e-4
    assert _sig_err(raw, g["raw"]) <= REF_BF16_MAX_RATIO * _sig_err(ref16, g["raw"]) + 1e-3


@pytest.mark.parametrize("name", [n for n in LOWP_REF if n.startswith("dinit_")])
def test_reference_constructor_weights_meet_1e2_in_bf16(dev, name):
    """north_star's bounds asserted outright - fp32 1e-3, fp16 1e-2, **bf16 1e-2** in sigmoid space - on weights drawn
    from the reference constructor's own distributions (kaiming-uniform convs, N(0, 0.02) linears; BatchNorm statistics
    and pos_emb mildly seeded), for the l, s, 4-GPT and x (cfg5) networks, vs the REFERENCE's recorded fp32 forward.
    The reference's own bf16 forward is 1.1-1.4e-3 away there."""
    from oracle.cft_oracle import OracleModel
    rec, cfg, model, rgb, ir = _lowp_case(name)
    want_pred, want_raw = OracleModel(cfg)(model.state_dict(), rgb, ir)
    for a, b in zip(want_raw, rec["raw"]):
        assert (a - b).abs().max().item() <= 2e-4            # the oracle reproduces the reference on these weights too
    pred, raw = _run(model, rgb, ir, dev, torch.float32)
    _check_fp32(pred, raw, want_pred, rec["raw"])
    pred, raw = _run(model, rgb, ir, dev, torch.float16)
    assert _sig_err(raw, rec["raw"]) <= 1e-2
    pred, raw = _run(model, rgb, ir, dev, torch.bfloat16)
    assert _sig_err(raw, rec["raw"]) <= 1e-2, f"bf16 sigmoid-space error {_sig_err(raw, rec['raw']):.3e}"
    assert (pred[..., 4:] - want_pred[..., 4:]).abs().max().item() <= 1e-2
    ref16 = [r.float() for r in rec["raw_bf16"]]
    assert _rms_rel(raw, rec["raw"]) <= 1.25 * _rms_rel(ref16, rec["raw"]) + 2e-4


# ------------------------------------------------------------------------- SURVEY.md section 8c weights, literally
@pytest.mark.parametrize("name", ["survey_l_x3_flir_256", "survey_l_x3_flir_unfused_192", "survey_s_1cft_256"])
def test_survey_weights_meet_the_literal_bounds(dev, name):
    """The golden-vector plan of SURVEY.md section 8c as written: torch.manual_seed(0) CONSTRUCTOR weights, BatchNorm running_mean ~ N(0, .1),
    running_var / weight ~ U(.5, 1.5), bias ~ N(0, .1), pos_emb ~ N(0, .02) (utils/seeded.survey_state_dict; this package's
    constructor draws the reference constructor's values - fingerprint pinned in tests/test_oracle_golden.py).  north_star's bounds
    asserted OUTRIGHT against the reference's recorded fp32 forward: 1e-3 fp32, 1e-2 fp16, **1e-2 bf16** (sigmoid space)."""
    from test_oracle_golden import survey_case
    from oracle.cft_oracle import OracleModel
    rec, cfg, model, sd, rgb, ir = survey_case(name)
    want_pred, _ = OracleModel(cfg)(model.state_dict(), rgb, ir)
    pred, raw = _run(model, rgb, ir, dev, torch.float32)
    _check_fp32(pred, raw, want_pred, rec["raw"])
    pred, raw = _run(model, rgb, ir, dev, torch.float16)
    assert _sig_err(raw, rec["raw"]) <= 1e-2
    pred, raw = _run(model, rgb, ir, dev, torch.bfloat16)
    assert _sig_err(raw, rec["raw"]) <= 1e-2, f"bf16 sigmoid-space error {_sig_err(raw, rec['raw']):.3e}"
    assert (pred[..., 4:] - want_pred[..., 4:]).abs().max().item() <= 1e-2


# ------------------------------------------------------------------------- the gain ladder: a bf16 bound that can fail
from test_oracle_golden import _ladder_ref, ladder_case  # noqa: E402


def _ladder_gains():
    from msod_amd.utils.seeded import LADDER_GAINS
    return [g for g in LADDER_GAINS if g not in (1.0, 1.35)]        # (the CPU suite checks the whole ladder; five rungs keep the GPU suite short)


@pytest.mark.parametrize("gain", _ladder_gains())
def test_bf16_on_the_gain_ladder(dev, gain):
    """cfg3 at 256 x 256 on every rung of the gain ladder (utils/seeded.ladder_state_dict), against the REFERENCE's recorded fp32 forward
    (tests/golden/ladder_ref.pt).  On the rungs where the reference's own bf16-autocast forward meets north_star's 1e-2 (sigmoid space) AND
    the output moves by >= 1e-2 rms when the images change - the bound can fail there for a kernel bug anywhere in the network - HIP bf16
    is held to **1e-2 outright**; above (the stress weights), to 1.3 x the reference's own bf16 error; fp16 to 1e-2 on every rung, fp32 to 1e-3."""
    rec, cfg, model, rgb, ir = ladder_case(gain)
    pred, raw = _run(model, rgb, ir, dev, torch.float32)
    for a, b in zip(raw, rec["raw"]):                    # (the reference's recorded forward itself: no oracle run on the GPU box's host)
        assert a.shape == b.shape and (a - b).abs().max().item() <= 1e-3
    pred, raw = _run(model, rgb, ir, dev, torch.float16)
    assert _sig_err(raw, rec["raw"]) <= 1e-2
    pred, raw = _run(model, rgb, ir, dev, torch.bfloat16)
    err = _sig_err(raw, rec["raw"])
    if rec["sig_err_bf16"] <= 1e-2:
        assert err <= 1e-2, f"gain {gain}: bf16 sigmoid-space error {err:.3e} (reference's own bf16: {rec['sig_err_bf16']:.3e}, input sensitivity {rec['sens_logit_rms']:.2e})"
    else:
        assert err <= 1.3 * rec["sig_err_bf16"], f"gain {gain}: bf16 error {err:.3e} vs the reference's own {rec['sig_err_bf16']:.3e}"
    # ... and the HIP output itself moves with the images as the reference's does (a forward that ignored its input would pass a
    # bound on insensitive weights): same second image pair as make_golden.py
    from msod_amd.utils.seeded import seeded_inputs
    c = rec["case"]
    rgb2, ir2 = seeded_inputs(c["batch"], c["height"], c["width"], c["seed"] + 100)
    _, raw2 = _run(model, rgb2, ir2, dev, torch.bfloat16)
    moved = (_flat(raw2) - _flat(raw)).pow(2).mean().sqrt().item()
    assert 0.7 * rec["sens_logit_rms"] - 2e-3 <= moved <= 1.3 * rec["sens_logit_rms"] + 2e-3


def test_cfg3_ladder_rung_at_the_benchmarked_shape_bf16_within_1e2(dev):
    """The benchmarked configuration (yolov5l + CFTx3, 640 x 640, BN folded, HIP-graph replay, bf16; 8 of the 64 pairs) on the HIGHEST
    ladder rung whose reference-side bf16 error is below 1e-2 (gain 1.4 at 256 x 256: 8.9e-3, input sensitivity 4.2e-2 rms): **1e-2
    asserted outright** against the fp32 oracle, which tests/test_oracle_golden.py pins to the reference on this rung."""
    from msod_amd.models.configs import named_config
    from msod_amd.models.yolo_test import Model
    from msod_amd.utils.seeded import BENCH_LADDER_GAIN, ladder_state_dict, seeded_inputs
    from oracle.cft_oracle import OracleModel
    cfg = named_config("cfg3")
    model = Model(cfg)
    model.load_state_dict(ladder_state_dict(model.state_dict(), BENCH_LADDER_GAIN, 5))
    model.fuse()
    rgb, ir = seeded_inputs(8, 640, 640, 0)
    idx = [0, 7]
    want_pred, want_raw = OracleModel(cfg)(model.state_dict(), rgb[idx], ir[idx])
    model = model.to(dev)
    x, x2 = rgb.to(dev), ir.to(dev)
    model.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        model.capture(8, 640, 640)
        pred, raw = model(x, x2)
        torch.cuda.synchronize()
        pred, raw = pred[idx].cpu(), [r[idx].cpu() for r in raw]
    model.release_graphs()
    err = _sig_err(raw, want_raw)
    assert err <= 1e-2, f"bf16 sigmoid-space error {err:.3e} at the bench shape on ladder rung {BENCH_LADDER_GAIN}"
    moved = (_flat([r[:1] for r in want_raw]) - _flat([r[1:] for r in want_raw])).pow(2).mean().sqrt().item()
    assert moved >= 1e-2           # pairs 0 and 7 differ by much more than the bound: the output depends on the images


def test_cfg3_survey_weights_at_the_benchmarked_batch_of_64_bf16_within_1e2(dev):
    """The benchmarked configuration (yolov5l + CFTx3, 640x640, 64 pairs, BN folded, HIP-graph replay, bf16) on the weights SURVEY.md
    section 8c prescribes: pairs 0 and 63 vs the fp32 oracle, **1e-2 asserted outright in bf16** (and in fp16).  The lively
    `seeded_state_dict` weights of the test below stay as the stress case, gated on the reference's own bf16 level."""
    from msod_amd.models.configs import named_config
    from msod_amd.models.yolo_test import Model
    from msod_amd.utils.seeded import seeded_inputs, survey_state_dict
    from oracle.cft_oracle import OracleModel
    cfg = named_config("cfg3")
    model = Model(cfg)
    model.load_state_dict(survey_state_dict(lambda: Model(cfg), seed=0))
    model.fuse()
    rgb, ir = seeded_inputs(64, 640, 640, 0)
    idx = [0, 63]
    want_pred, want_raw = OracleModel(cfg)(model.state_dict(), rgb[idx], ir[idx])
    model = model.to(dev)
    x, x2 = rgb.to(dev), ir.to(dev)
    for dtype in (torch.bfloat16, torch.float16):
        model.set_compute_dtype(dtype)
        with torch.no_grad():
            model.capture(64, 640, 640)
            pred, raw = model(x, x2)
            torch.cuda.synchronize()
            pred, raw = pred[idx].cpu(), [r[idx].cpu() for r in raw]
        model.release_graphs()
        assert torch.isfinite(pred).all()
        err = _sig_err(raw, want_raw)
        assert err <= 1e-2, f"{dtype}: sigmoid-space error {err:.3e} at the bench shape on the survey weights"
        assert (pred[..., 4:] - want_pred[..., 4:]).abs().max().item() <= 1e-2


def test_cfg3_at_the_benchmarked_batch_of_64(dev):
    """The configuration bench.py times: yolov5l + CFTx3, 640x640, SIXTY-FOUR pairs per GPU, BN folded, HIP-graph replay - at
    this M the dispatcher picks the 256x256 16-wave tiles, the chunk-major K walk and the fused Bottleneck kernels, which
    the 2-pair test above never reaches.  Pairs 0 and 63 vs the oracle in bf16 and fp16.  fp16 is the DECLARED parity-green 16-bit
    mode (profiles/r04_bf16_sites.md): north_star's literal 1e-2 is asserted for it outright at this shape;
    bf16 is gated on the live reference-style bf16 forward (oracle under CPU autocast, pinned to the reference in
    tests/test_oracle_golden.py) - two thirds of its error is the one rounding of the weights, which no kernel can remove."""
    from msod_amd.utils.seeded import seeded_inputs
    from oracle.cft_oracle import OracleModel
    from oracle.lowp_oracle import AutocastOracle
    cfg, model, sd = _seeded("cfg3", 0)
    rgb, ir = seeded_inputs(64, 640, 640, 0)
    idx = [0, 63]
    model.fuse()
    fsd = model.state_dict()
    want_pred, want_raw = OracleModel(cfg)(fsd, rgb[idx], ir[idx])
    _, ref16 = AutocastOracle(cfg)(fsd, rgb[idx], ir[idx])
    model = model.to(dev)
    x, x2 = rgb.to(dev), ir.to(dev)
    for dtype in (torch.bfloat16, torch.float16):
        model.set_compute_dtype(dtype)
        with torch.no_grad():
            model.capture(64, 640, 640)
            pred, raw = model(x, x2)
            torch.cuda.synchronize()
            pred, raw = pred[idx].cpu(), [r[idx].cpu() for r in raw]
        model.release_graphs()
        assert torch.isfinite(pred).all()
        if dtype == torch.float16:
            _check_f16(pred, raw, want_pred, want_raw)
        else:
            _check_bf16(pred, raw, want_pred, want_raw)
            assert _rms_rel(raw, want_raw) <= REF_BF16_RMS_RATIO * _rms_rel(ref16, want_raw) + 2e-4
            assert _sig_err(raw, want_raw) <= REF_BF16_MAX_RATIO * _sig_err(ref16, want_raw) + 1e-3


def test_three_single_stream_forwards_in_flight_are_bit_stable_at_the_benchmarked_shape(dev):
    """bench.py's default step mode: THREE captured forwards in flight, each on ONE HIP stream, replayed round-robin
    through distributed.ForwardPipeline at the benchmarked shape (64 pairs, 640 x 640, bf16: the asm GEMM kernels, the chained asm pairs and
    the fused Bottlenecks all run, three kernels of different forwards at a time).  Every replay of every graph must reproduce, bit for bit,
    what the two-stream graph computes alone - a missing wait or barrier in a hand-scheduled loop shows up as a rare mismatch under exactly
    this kind of co-scheduling.  On the way: the eager forward with the asm kernels switched off (variant 97) equals the default bit for bit."""
    from msod_amd import distributed as D
    from msod_amd.graph import CapturedForward
    from msod_amd.utils.seeded import seeded_inputs
    cfg, model, sd = _seeded("cfg3", 0)
    rgb, ir = seeded_inputs(64, 640, 640, 0)
    model = model.fuse().to(dev).set_compute_dtype(torch.bfloat16)
    x, x2 = rgb.to(dev), ir.to(dev)
    with torch.no_grad():
        model.capture(64, 640, 640)
        pred0, raw0 = model(x, x2)
        torch.cuda.synchronize()
        pred0, raw0 = pred0.clone(), [r.clone() for r in raw0]
        # the same forward without the hand-scheduled kernels (variant 97 = the round-5 choice: 16-wave kernels, 16-wave chained pairs): every asm
        # kernel is bit-identical to the kernel it replaces, so the whole forward must be
        from msod_amd import _lib
        lib = _lib.load()
        old = lib.cft_set_conv_variant(97)
        try:
            pred97, raw97 = model.forward_once(x, x2)
            torch.cuda.synchronize()
        finally:
            lib.cft_set_conv_variant(old)
        assert torch.equal(pred97, pred0)
        for a, b in zip(raw97, raw0):
            assert torch.equal(a, b)
        del pred97, raw97
        model.overlap_streams = False                     # (drops the two-stream graph)
        caps = [CapturedForward(model, 64, 640, 640) for _ in range(3)]
        for c in caps:
            c.rgb.copy_(x)
            c.ir.copy_(x2)
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=dev) for _ in caps]
        pipe = D.ForwardPipeline([(lambda c=c: c.replay_static()[0]) for c in caps], streams)
        for rnd in range(4):
            for _ in range(30):
                pipe.step()
            torch.cuda.synchronize()
            for c in caps:
                assert torch.equal(c.pred, pred0), f"round {rnd}"
                for a, b in zip(c.raw, raw0):
                    assert torch.equal(a, b), f"round {rnd}"
    model.overlap_streams = True
    model.release_graphs()


def test_cfg5_bf16_and_cfg4_at_640(dev):
    """Coverage holes named by VERDICT r2: cfg5 (yolov5x x3 CFT, 1280x1280) in bf16, and cfg4 (LLVIP yaml, nc = 1) at its
    own 640x640 shape in fp32 / fp16 / bf16; bf16 against the oracle and the live reference-style bf16 level."""
    from msod_amd.utils.seeded import seeded_inputs
    from oracle.cft_oracle import OracleModel
    from oracle.lowp_oracle import AutocastOracle
    for cfg_name, size, seed in (("cfg5", 1280, 5), ("cfg4", 640, 4)):
        cfg, model, sd = _seeded(cfg_name, seed)
        rgb, ir = seeded_inputs(1, size, size, seed)
        model.fuse()
        fsd = model.state_dict()
        want_pred, want_raw = OracleModel(cfg)(fsd, rgb, ir)
        _, ref16 = AutocastOracle(cfg)(fsd, rgb, ir)
        if cfg_name == "cfg4":
            pred, raw = _run(model, rgb, ir, dev, torch.float32)
            assert pred.shape == (1, 25200, 6)
            _check_fp32(pred, raw, want_pred, want_raw)
            pred, raw = _run(model, rgb, ir, dev, torch.float16)
            _check_f16(pred, raw, want_pred, want_raw)
        pred, raw = _run(model, rgb, ir, dev, torch.bfloat16)
        _check_bf16(pred, raw, want_pred, want_raw)
        assert _rms_rel(raw, want_raw) <= REF_BF16_RMS_RATIO * _rms_rel(ref16, want_raw) + 2e-4
        assert _sig_err(raw, want_raw) <= REF_BF16_MAX_RATIO * _sig_err(ref16, want_raw) + 1e-3


def test_float_channel_slices_of_a_six_channel_batch(dev):
    """the reference's callers pass `img[:, :3]`, `img[:, 3:]` of ONE float [B,6,H,W] batch (test.py:112-113,
    train.py:716-717); fp32 compute (the non-fused Focus path), fp16 images with bf16 compute, and the training forward."""
    from oracle.cft_oracle import OracleModel
    g, cfg, model, rgb, ir = load_case([p for p in GOLDEN if p.endswith("s_1cft_256.pt")][0])
    want_pred, want_raw = OracleModel(cfg)(model.state_dict(), rgb, ir)
    f6 = torch.cat([rgb, ir], 1).to(dev)
    model = model.to(dev).set_compute_dtype(torch.float32)
    with torch.no_grad():
        pred, raw = model(f6[:, :3], f6[:, 3:])
    _check_fp32(pred.cpu(), [r.cpu() for r in raw], want_pred, want_raw)
    model.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        pred, raw = model(f6.half()[:, :3], f6.half()[:, 3:])
    _check_bf16(pred.cpu(), [r.cpu() for r in raw], want_pred, want_raw)
    from msod_amd import ops
    model.set_compute_dtype(torch.float32).train()
    with torch.no_grad():
        ops.manual_dropout_seed(5)           # (training mode: the GPT dropout masks are drawn per call)
        raws = model(f6[:, :3], f6[:, 3:])
        ops.manual_dropout_seed(5)
        raws2 = model(f6[:, :3].contiguous(), f6[:, 3:].contiguous())
    assert all(torch.equal(a, b) for a, b in zip(raws, raws2))


def test_fresh_model_runs_in_its_declared_precision(dev):
    """`Model(cfg).cuda()` without set_compute_dtype() computes in Model.compute_dtype (bf16), not in fp32."""
    from msod_amd.models.common import Focus
    from msod_amd.models.configs import named_config
    from msod_amd.models.yolo_test import Model
    model = Model(named_config("cfg1")).to(dev)
    assert model.compute_dtype == torch.bfloat16
    assert all(m.compute_dtype == torch.bfloat16 for m in model.modules() if isinstance(m, Focus))
    seen = []
    hook = model.model[1].register_forward_hook(lambda mod, inp, out: seen.append(out.dtype))
    with torch.no_grad():
        model(torch.rand(1, 3, 64, 64, device=dev), torch.rand(1, 3, 64, 64, device=dev))
    hook.remove()
    assert seen == [torch.bfloat16]


# ------------------------------------------------------------------------- structural properties
def test_unfused_equals_fused(dev):
    """BN folding happens at pack time either way: model.fuse() must not change the outputs."""
    g, cfg, model, rgb, ir = load_case([p for p in GOLDEN if p.endswith("s_x3_rect.pt")][0])
    p0, _ = _run(model, rgb, ir, dev, torch.float32)
    model.fuse()
    p1, _ = _run(model, rgb, ir, dev, torch.float32)
    assert torch.allclose(p0, p1, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_graph_replay_is_bit_identical_to_eager(dev, dtype):
    g, cfg, model, rgb, ir = load_case([p for p in GOLDEN if p.endswith("s_1cft_256.pt")][0])
    model = model.to(dev).set_compute_dtype(dtype)
    with torch.no_grad():
        e_pred, e_raw = model.forward_once(rgb.to(dev), ir.to(dev))
        e_pred = e_pred.clone()
        model.capture(rgb.shape[0], rgb.shape[2], rgb.shape[3])
        for _ in range(2):
            g_pred, g_raw = model(rgb.to(dev), ir.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(e_pred, g_pred)
    model.release_graphs()


def test_two_forwards_in_flight_do_not_interfere(dev):
    """bench.py's step mode (distributed.ForwardPipeline): two captured graphs with their OWN static buffers and DIFFERENT inputs,
    replayed round-robin on two streams so that they overlap on the GPU - every replay must reproduce the eager result of its
    own input bit for bit (no shared scratch between forwards in flight)."""
    from msod_amd import distributed as D
    from msod_amd.graph import CapturedForward
    from msod_amd.utils.seeded import seeded_inputs
    g, cfg, model, rgb, ir = load_case([p for p in GOLDEN if p.endswith("s_x3_320.pt")][0])
    model = model.to(dev).set_compute_dtype(torch.bfloat16)
    ins = [seeded_inputs(4, 320, 320, s) for s in (71, 72)]
    with torch.no_grad():
        want = [model.forward_once(a.to(dev), b.to(dev))[0].clone() for a, b in ins]
        caps = [CapturedForward(model, 4, 320, 320) for _ in ins]
        for c, (a, b) in zip(caps, ins):
            c.rgb.copy_(a)
            c.ir.copy_(b)
        runners = [(lambda c=c: c.replay_static()[0]) for c in caps]
        streams, table = D.ForwardPipeline.pick_streams(runners, dev, groups=2, probe_steps=2)
        pipe = D.ForwardPipeline(runners, streams)
        assert len(table) == 2 and all(t > 0 for t in table)
        for _ in range(12):
            pipe.step()
        torch.cuda.synchronize()
    assert not torch.equal(want[0], want[1])
    for c, w in zip(caps, want):
        assert torch.equal(c.pred, w)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_depth_first_prefix_is_bit_identical(dev, dtype):
    """Model.depth_first = (chunks, rows): the image-only prefix of each backbone runs sub-batch by sub-batch (Infinity-Cache
    residency of the P1 / P2 tensors).  Per-image arithmetic is untouched, so outputs are bit-identical to the layer-by-layer walk:
    even and ragged chunk counts, the whole prefix or its first three rows, one lane or two, eager and HIP-graph replay."""
    from msod_amd.utils.seeded import seeded_inputs
    g, cfg, model, _, _ = load_case([p for p in GOLDEN if p.endswith("s_x3_320.pt")][0])
    model = model.to(dev).set_compute_dtype(dtype)
    assert model.prefix_segments() == [(0, 4), (5, 9)] and model.prefix_segments(3) == [(0, 2), (5, 7)]
    rgb, ir = (t.to(dev) for t in seeded_inputs(12, 160, 192, 3))
    with torch.no_grad():
        want = model.forward_once(rgb, ir)[0].clone()
        for df in ((4, None), (5, 3), (6, 5)):
            model.depth_first = df
            got = model.forward_once(rgb, ir)[0]
            assert torch.equal(got, want), df
        model.overlap_streams = False
        assert torch.equal(model.forward_once(rgb, ir)[0], want)
        model.overlap_streams = True
        model.capture(12, 160, 192)
        for _ in range(2):
            got = model(rgb, ir)[0]
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        model.depth_first = None                 # the setter drops the graph: the next call walks layer by layer again
        assert not model._graphs
    model.release_graphs()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_cft_output_fusion_matches_the_three_launch_path(dev, dtype):
    """Model.cft_fusion_plan: the two Add2 layers behind every GPT block and the Add that sums them run as one kernel
    (cft_gpt_upsample_add2).  The plan finds the three groups of an x3 config; fp32 results are bit-identical to the unfused walk
    (same fp32 operations); in bf16 the Add is formed from unrounded sums, so the outputs differ from the unfused walk by roundings
    only and are no further from the oracle."""
    g, cfg, model, rgb, ir = load_case([p for p in GOLDEN if p.endswith("s_x3_320.pt")][0])
    model = model.to(dev).set_compute_dtype(dtype)
    plan = model.cft_fusion_plan()
    assert len(plan) == 3 and all(k is not None for (_, _, k) in plan.values())
    with torch.no_grad():
        model.fuse_cft_outputs = True
        pred_f, raw_f = model.forward_once(rgb.to(dev), ir.to(dev))
        model.overlap_streams = False
        pred_f1, _ = model.forward_once(rgb.to(dev), ir.to(dev))
        model.overlap_streams = True
        model.fuse_cft_outputs = False
        pred_u, raw_u = model.forward_once(rgb.to(dev), ir.to(dev))
        model.fuse_cft_outputs = True
    torch.cuda.synchronize()
    assert torch.equal(pred_f, pred_f1)                      # one lane or two: same kernels
    if dtype == torch.float32:
        assert torch.equal(pred_f, pred_u)
    else:
        want_raw = [r for r in g["raw"]]
        e_f, e_u = _sig_err([r.cpu() for r in raw_f], want_raw), _sig_err([r.cpu() for r in raw_u], want_raw)
        assert _sig_err([r.cpu() for r in raw_f], [r.cpu() for r in raw_u]) < 2e-2 and e_f <= 1.15 * e_u + 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_conv_c3_chain_is_bit_identical_to_the_layerwise_walk(dev, dtype):
    """Model.chain_plan: the stride-2 Conv in front of every backbone C3 is handed to that C3 un-run; where the pair is eligible
    (yolov5l: rows 1-2 / 6-7, 64 -> 128 channels, and rows 3-4 / 8-9, 128 -> 256) both run as one cft_conv2d_chain kernel; so do cv2 of Bottleneck j and cv1 of
    Bottleneck j + 1 inside the head's 256-channel C3s, which have no shortcuts.  Same arithmetic: the detections are
    bit-identical to the walk that runs every layer on its own, on one lane or two, and under HIP-graph replay."""
    from msod_amd import ops
    from msod_amd.utils.seeded import seeded_inputs
    cfg, model, sd = _seeded("cfg3", 3)
    model = model.to(dev).set_compute_dtype(dtype)
    assert model.chain_plan() == frozenset({1, 3, 6, 8, 13, 15})
    rgb, ir = seeded_inputs(2, 192, 256, 3)
    x, x2 = rgb.to(dev), ir.to(dev)
    log = []
    monkey_rows = ops.CHAIN_RES_MIN_ROWS
    ops.CHAIN_RES_MIN_ROWS = 0                       # (a 2-pair test: take the chained 3x3 + shortcut + 1x1 kernel regardless of the size heuristic)
    try:
        with torch.no_grad():
            ops.set_launch_log(log)
            try:
                model.overlap_streams = False
                pred_c, raw_c = model.forward_once(x, x2)
            finally:
                ops.set_launch_log(None)
            model.overlap_streams = True
            pred_c2, _ = model.forward_once(x, x2)
            model.chain_convs = False
            pred_u, raw_u = model.forward_once(x, x2)
            model.chain_convs = True
            pred_u8 = model.forward_once((x * 255).round().to(torch.uint8), (x2 * 255).round().to(torch.uint8))[0]      # uint8 images, chained
            model.chain_convs = False
            pred_u8_u = model.forward_once((x * 255).round().to(torch.uint8), (x2 * 255).round().to(torch.uint8))[0]
            model.chain_convs = True
            model.capture(2, 192, 256)
            pred_g = model(x, x2)[0].clone()
        torch.cuda.synchronize()
    finally:
        ops.CHAIN_RES_MIN_ROWS = monkey_rows
    # rows 1-2, 3-4 (RGB), 6-7, 8-9 (IR); and inside the two 256-channel C3s of the head (no shortcuts) cv2[j] + cv1[j + 1], j = 0, 1
    assert sum(1 for rec in log if rec[0].startswith("conv_focus")) == 2
    assert sum(1 for rec in log if rec[0].startswith("conv_chain_k3s2")) == 4 and sum(1 for rec in log if rec[0].startswith("conv_chain_k3s1")) == 4
    # rows 14 / 16: the 256-channel C3s WITH shortcuts (n = 9): cv2[j] (+ shortcut) and cv1[j + 1] as one launch, j = 0..7, per stream
    assert sum(1 for rec in log if rec[0].startswith("conv_chainres_k3s1_n256")) == 16
    assert sum(1 for rec in log if rec[0] == "conv_k1s1_n256_K256") <= 7       # (was 23: 18 of them were Bottleneck cv1 launches; 2 remain)
    assert torch.equal(pred_c, pred_u) and torch.equal(pred_c2, pred_u) and torch.equal(pred_g, pred_u)
    assert torch.equal(pred_u8, pred_u8_u)
    assert all(torch.equal(a, b) for a, b in zip(raw_c, raw_u))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_splitk_cft_blocks_match_the_one_launch_path_and_the_oracle(dev, dtype):
    """Model.splitk: the CFT blocks' out_proj / fc2 run as split-K GEMMs whose fp32 partial sums the next LayerNorm folds into the residual
    stream.  Same products, another fp32 summation order: the detections agree with the one-launch-per-GEMM walk to fp32 rounding level, are
    reproducible run to run (fixed order, no atomics), and sit as close to the oracle."""
    from msod_amd import ops
    from oracle.cft_oracle import OracleModel
    from msod_amd.utils.seeded import seeded_inputs
    cfg, model, sd = _seeded("cfg3", 4)
    rgb, ir = seeded_inputs(8, 128, 160, 4)           # 1024 token rows: every out_proj / fc2 of the three blocks is split
    want_pred, want_raw = OracleModel(cfg)(sd, rgb[:2], ir[:2])
    model = model.to(dev).set_compute_dtype(dtype)
    x, x2 = rgb.to(dev), ir.to(dev)
    log = []
    with torch.no_grad():
        ops.set_launch_log(log)
        try:
            pred_s, raw_s = model.forward_once(x, x2)
        finally:
            ops.set_launch_log(None)
        pred_s2, _ = model.forward_once(x, x2)
        model.splitk = False
        pred_1, raw_1 = model.forward_once(x, x2)
        model.splitk = True
    torch.cuda.synchronize()
    assert torch.equal(pred_s, pred_s2)                                      # reproducible
    rs, r1 = [r[:2].cpu() for r in raw_s], [r[:2].cpu() for r in raw_1]
    if dtype == torch.float32:
        assert max((a - b).abs().max().item() for a, b in zip(rs, r1)) <= 2e-4
        _check_fp32(pred_s[:2].cpu(), rs, want_pred, want_raw)
    else:
        assert _sig_err(rs, r1) <= 1.5e-2 and _sig_err(rs, want_raw) <= 1.15 * _sig_err(r1, want_raw) + 1e-3
    n_ln = sum(1 for rec in log if rec[0] == "cft_layernorm")
    assert n_ln == 3 * 17                                                    # same LayerNorm launches; 48 of them now also fold partial sums


def test_captured_graph_is_dropped_when_weights_change(dev):
    """a captured graph replays the packed weights of capture time; in-place weight updates,
    load_state_dict and .to()/.half() must not return detections of the old weights."""
    from msod_amd.utils.seeded import seeded_state_dict
    g, cfg, model, rgb, ir = load_case([p for p in GOLDEN if p.endswith("s_add_320.pt")][0])
    model = model.to(dev).set_compute_dtype(torch.float32)
    x, x2 = rgb.to(dev), ir.to(dev)
    with torch.no_grad():
        model.capture(rgb.shape[0], rgb.shape[2], rgb.shape[3])
        before = model(x, x2)[0].clone()
        assert len(model._graphs) == 1
        sd2 = seeded_state_dict(model.state_dict(), 77)
        model.load_state_dict(sd2)                         # invalidates explicitly
        assert len(model._graphs) == 0
        after = model(x, x2)[0].clone()
        from msod_amd.models.yolo_test import Model
        fresh = Model(cfg)
        fresh.load_state_dict(sd2)
        want = fresh.to(dev).set_compute_dtype(torch.float32)(x, x2)[0]
        assert not torch.equal(before, after) and torch.equal(after, want)
        model.capture(rgb.shape[0], rgb.shape[2], rgb.shape[3])
        next(model.parameters()).mul_(1.5)                 # in-place update: caught by the version fingerprint at replay
        third = model(x, x2)[0]
        assert len(model._graphs) == 0 and not torch.equal(third, after)
    torch.cuda.synchronize()


def test_pairs_are_independent_at_full_shape(dev):
    """Size-independent property at the BASELINE shape (yolov5l+CFTx3, 640x640): the forward of a pair
    does not depend on its batch-mates, so pair 5 of a batch of 8 equals the same pair run alone
    (different tile shapes / grid sizes, same arithmetic per pixel)."""
    from msod_amd.utils.seeded import seeded_inputs
    cfg, model, sd = _seeded("cfg3", 11)
    model = model.to(dev).fuse().set_compute_dtype(torch.bfloat16)
    rgb, ir = seeded_inputs(8, 640, 640, 11)
    with torch.no_grad():
        full, _ = model(rgb.to(dev), ir.to(dev))
        one, _ = model(rgb[5:6].to(dev), ir[5:6].to(dev))
    torch.cuda.synchronize()
    assert full.shape == (8, 25200, 8) and torch.isfinite(full).all()
    assert torch.equal(full[5:6], one)
    xy = full[..., :2]
    assert xy.min() > -64 and xy.max() < 640 + 64                  # decoded centres stay near the image
    assert (full[..., 4:] >= 0).all() and (full[..., 4:] <= 1).all()


# ------------------------------------------------------------------------- boundary: inputs the reference's callers hold
def test_uint8_pair_input_matches_oracle_on_normalised_floats(dev):
    """Caller-side pre-processing: the model accepts the uint8 RGB / IR views of the
    reference's [B,6,H,W] batch directly; compared with the ORACLE evaluated on `img.float()/255`
    (reference test.py:106-113), in fp32 (tight) and fp16 (1e-2)."""
    from oracle.cft_oracle import OracleModel
    g, cfg, model, rgb, ir = load_case([p for p in GOLDEN if p.endswith("s_x3_320.pt")][0])
    img6 = (torch.cat([rgb, ir], 1) * 255).round().to(torch.uint8)
    f = img6.float() / 255.0
    want_pred, want_raw = OracleModel(cfg)(model.state_dict(), f[:, :3], f[:, 3:])
    d6 = img6.to(dev)
    for dtype, check in ((torch.float32, _check_fp32), (torch.float16, _check_f16)):
        model = model.to(dev).set_compute_dtype(dtype)
        with torch.no_grad():
            p8, r8 = model(d6[:, :3], d6[:, 3:])
        torch.cuda.synchronize()
        check(p8.cpu(), [r.cpu() for r in r8], want_pred, want_raw)


def test_model_half_with_half_images_like_the_reference_callers(dev):
    """test.py:66-68 / detect_twostream.py:40-41: `model.half()` then `img.half()` inputs."""
    from oracle.cft_oracle import OracleModel
    g, cfg, model, rgb, ir = load_case([p for p in GOLDEN if p.endswith("s_1cft_256.pt")][0])
    want_pred, want_raw = OracleModel(cfg)(model.state_dict(), rgb, ir)
    model = model.to(dev).half()
    assert model.compute_dtype == torch.float16 and next(model.parameters()).dtype == torch.float16
    with torch.no_grad():
        pred, raw = model(rgb.to(dev).half(), ir.to(dev).half())
    torch.cuda.synchronize()
    assert pred.dtype == torch.float32                       # detections stay fp32 (Detect logits come out of the GEMM in fp32)
    _check_f16(pred.cpu(), [r.cpu() for r in raw], want_pred, want_raw)
    model.float()
    assert model.compute_dtype == torch.float32


def test_reference_pickled_checkpoint_runs_on_gpu(dev):
    """a checkpoint pickled by the REFERENCE's own classes (tests/golden/ref_ckpt_tiny.pt, written
    by make_golden.py `ckpt` in the build container exactly as train.py:850-860 does) goes through
    compat.attempt_load (= models/experimental.py:113-134) and its forward matches the reference's recorded
    outputs and the oracle; also the list / Ensemble form."""
    from msod_amd import compat
    from oracle.cft_oracle import OracleModel
    from msod_amd.utils.seeded import seeded_inputs
    ck = os.path.join(HERE, "golden", "ref_ckpt_tiny.pt")
    out = torch.load(os.path.join(HERE, "golden", "ref_ckpt_tiny_out.pt"), weights_only=False)
    model = compat.attempt_load(ck, map_location="cpu")
    assert type(model).__module__.startswith("msod_amd") and model.compute_dtype == torch.float32
    assert model.names == out["names"] and torch.equal(model.stride, out["stride"])
    rgb, ir = seeded_inputs(out["batch"], out["height"], out["width"], out["seed"])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    want_pred, want_raw = OracleModel(out["cfg"])(sd, rgb, ir)
    model = model.to(dev)
    with torch.no_grad():
        pred, raw = model(rgb.to(dev), ir.to(dev))
    pred, raw = pred.cpu(), [r.cpu() for r in raw]
    _check_fp32(pred, raw, want_pred, want_raw)
    _check_fp32(pred, raw, out["pred"], out["raw"])           # the reference's own forward of the same checkpoint
    with torch.no_grad():
        model.half()
        p16, r16 = model(rgb.to(dev).half(), ir.to(dev).half())
    _check_f16(p16.cpu(), [r.cpu() for r in r16], out["pred"], out["raw"])
    ens = compat.attempt_load([ck, ck], map_location="cpu").to(dev)
    assert type(ens).__name__ == "Ensemble" and len(ens) == 2 and ens.names == out["names"]
    with torch.no_grad():
        y, none = ens(rgb.to(dev), ir.to(dev))
    assert none is None and y.shape == (out["batch"], 2 * out["pred"].shape[1], out["pred"].shape[2])
    n = out["pred"].shape[1]
    assert torch.allclose(y[:, :n].cpu(), out["pred"], rtol=1e-3, atol=1e-3) and torch.equal(y[:, :n], y[:, n:])
    import sys
    for name in ("models", "models.common", "models.yolo_test"):
        sys.modules.pop(name, None)


@pytest.mark.parametrize("shape", [(1, 64, 96), (3, 96, 64), (5, 32, 32)], ids=["b1_rect", "b3_rect", "b5_min"])
def test_small_odd_shapes_match_oracle(dev, shape):
    """Batch 1 / odd batches, the smallest legal image (32x32: 1x1 P5 maps, adaptive pooling windows that
    repeat pixels, bilinear upsampling from 8x8 DOWN to 1x1) and non-square inputs, fp32 + fp16 vs the oracle."""
    from msod_amd.utils.seeded import seeded_inputs
    from oracle.cft_oracle import OracleModel
    b, h, w = shape
    cfg, model, sd = _seeded("yolov5s_fusion_transformerx3_vedai", 21)
    rgb, ir = seeded_inputs(b, h, w, 21)
    want_pred, want_raw = OracleModel(cfg)(sd, rgb, ir)
    pred, raw = _run(model, rgb, ir, dev, torch.float32)
    _check_fp32(pred, raw, want_pred, want_raw)
    pred, raw = _run(model, rgb, ir, dev, torch.float16)
    _check_f16(pred, raw, want_pred, want_raw)


def test_profile_flag_reports_per_layer_times(dev):
    """forward_once(profile=True) (reference models/yolo_test.py:252-260,270-271): one row per top-level layer with its
    time, the algorithmic GFLOP of its GEMM launches, parameter count and type; the outputs are those of a normal forward."""
    g, cfg, model, rgb, ir = load_case([p for p in GOLDEN if p.endswith("s_1cft_256.pt")][0])
    model = model.to(dev).set_compute_dtype(torch.float32)
    with torch.no_grad():
        want, _ = model(rgb.to(dev), ir.to(dev))
        got, _ = model(rgb.to(dev), ir.to(dev), profile=True)
    assert torch.equal(want, got)
    rows = model.profile_ms
    assert len(rows) == len(model.model) and [r[0] for r in rows] == list(range(len(rows)))
    assert all(r[2] > 0 for r in rows) and rows[0][1] == "Focus" and rows[-1][1] == "Detect"
    from oracle.cft_oracle import algorithmic_flops
    total = sum(r[3] for r in rows) * 1e9
    want_fl = algorithmic_flops(cfg, rgb.shape[2], rgb.shape[3])
    assert abs(total / rgb.shape[0] - want_fl["total"]) / want_fl["total"] < 0.02     # convs + linears + QK^T / AV


def test_bad_image_sizes_raise(dev):
    from msod_amd.models.configs import named_config
    from msod_amd.models.yolo_test import Model
    model = Model(named_config("cfg1")).to(dev)
    with pytest.raises(ValueError, match="multiples of the largest stride"):
        model(torch.zeros(1, 3, 100, 64, device=dev), torch.zeros(1, 3, 100, 64, device=dev))
    with pytest.raises(ValueError, match="equal shape"):
        model(torch.zeros(1, 3, 64, 64, device=dev), torch.zeros(2, 3, 64, 64, device=dev))


def test_yolov5x_four_cft_blocks_matches_oracle(dev):
    """Reference yaml `yolov5x_fusion_transformer_FLIR` (depth 1.33 / width 1.25, FOUR GPT blocks): channel counts
    80/160/320/640/1280 exercise K steps that straddle taps, N tails (160, 320 are not multiples of the 128/256
    tiles) and zero-padded attention heads (d=160 -> head width 20 -> 32).  fp32 / fp16 / bf16 vs the oracle."""
    from msod_amd.utils.seeded import seeded_inputs
    from oracle.cft_oracle import OracleModel
    cfg, model, sd = _seeded("yolov5x_fusion_transformer_FLIR", 31)
    rgb, ir = seeded_inputs(1, 128, 160, 31)
    want_pred, want_raw = OracleModel(cfg)(sd, rgb, ir)
    pred, raw = _run(model, rgb, ir, dev, torch.float32)
    _check_fp32(pred, raw, want_pred, want_raw)
    pred16, raw16 = _run(model, rgb, ir, dev, torch.float16)
    _check_f16(pred16, raw16, want_pred, want_raw)
    predb, rawb = _run(model, rgb, ir, dev, torch.bfloat16)
    _check_bf16(predb, rawb, want_pred, want_raw)


def test_modules_under_a_foreign_executor_with_torch_neighbours(dev):
    """The strict form of the boundary on a GPU (the CPU test
    test_reference_graph_file_builds_over_this_packages_common shows the reference's own graph file building over
    this package's models.common - the reference tree itself is not on the GPU box).  Here the HIP-backed modules are
    driven the way that file drives them: a plain graph walk (restating models/yolo_test.py:235-272), torch's own
    nn.Upsample between them and a torch Detect (the oracle's restatement of models/yolo_test.py:41-59) on their
    outputs; the modules compute in the precision of their parameters - fp32, and fp16 after .half()."""
    import torch.nn as nn
    from msod_amd.models.common import Focus, Upsample
    from msod_amd.utils.seeded import seeded_inputs
    from oracle import cft_oracle as O
    cfg, model, sd = _seeded("yolov5s_fusion_transformerx3_vedai", 61)
    rgb, ir = seeded_inputs(2, 96, 160, 61)
    want_pred, want_raw = O.OracleModel(cfg)(sd, rgb, ir)
    layers = []
    for m in model.model:
        if isinstance(m, Upsample):                      # what the reference's parse_model builds: torch's Upsample
            t = nn.Upsample(None, 2, "nearest")
            t.i, t.f = m.i, m.f
            m = t
        layers.append(m)
    for m in model.modules():
        if isinstance(m, Focus):
            m.compute_dtype = None                       # precision of the parameters, as under the reference's Model
    body = nn.Sequential(*layers[:-1]).to(dev)
    det = layers[-1]
    ag = O.sorted_anchors(cfg["anchors"])[1]

    def walk(x, x2):
        y = []
        for m in body:
            if m.f == -4:
                x = m(x2)
            else:
                if m.f != -1:
                    x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
                x = m(x)
            y.append(x if m.i in model.save else None)
        feats = [y[j] for j in det.f]
        torch.cuda.synchronize()
        return O.detect(sd, f"model.{det.i}.", [f.float().cpu().contiguous() for f in feats], cfg["nc"], ag)

    with torch.no_grad():
        pred, raw = walk(rgb.to(dev), ir.to(dev))
        _check_fp32(pred, raw, want_pred, want_raw)
        body.half()
        pred, raw = walk(rgb.to(dev).half(), ir.to(dev).half())
        _check_f16(pred, raw, want_pred, want_raw)
