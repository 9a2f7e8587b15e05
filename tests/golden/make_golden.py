"""Generate the golden vectors that pin ``oracle/cft_oracle.py`` to the reference.

Runs ONLY in the build container (needs /root/reference).  It imports the reference's own
``models.yolo_test.Model`` unmodified - with empty stand-in modules for the import-time-only
dependencies that are missing from this image (cv2, torchvision, seaborn) -
loads the deterministic synthetic weights of ``utils/seeded.py``, runs the reference
forward on CPU fp32 and stores inputs' seeds, outputs and per-layer statistics.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.pt

The cases cover: every fusion layout, s and l sizes, nc in {1,3,9}, square and rectangular
inputs, batch > 1, BN-unfused and ``.fuse()``d forwards, and the derived BASELINE configs.
"""
import hashlib
import logging
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

# name, config name, batch, H, W, fused, seed
CASES = [
    ("s_add_320", "cfg1", 1, 320, 320, False, 0),
    ("s_add_320_fused", "cfg1", 1, 320, 320, True, 0),
    ("s_1cft_256", "cfg2", 2, 256, 256, True, 1),
    ("s_x3_320", "yolov5s_fusion_transformerx3_vedai", 1, 320, 320, True, 2),
    ("s_x3_rect", "yolov5s_fusion_transformerx3_vedai", 2, 192, 320, False, 3),
    ("s_x4_256", "yolov5s_fusion_transformer_vedai", 1, 256, 256, True, 4),
    ("l_x3_flir_256", "cfg3", 1, 256, 256, True, 5),
    ("l_x3_llvip_192", "cfg4", 1, 192, 192, True, 6),
]


def install_reference():
    for name in ("cv2", "torchvision", "seaborn"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["cv2"].setNumThreads = lambda n: None
    sys.path.insert(0, REF)
    logging.disable(logging.CRITICAL)


def tap_stats(t):
    """Compact, order-sensitive fingerprint of a feature map: moments + 32 strided samples."""
    f = t.detach().float().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, 32).long()
    return {"shape": tuple(t.shape), "mean": f.mean().item(), "std": f.std().item(),
            "absmax": f.abs().max().item(), "samples": f[idx].clone()}


def main():
    install_reference()
    sys.path.insert(0, ROOT)
    import msod_amd  # noqa: F401  (package alias; pure-python parts only)
    from msod_amd.models.configs import named_config
    from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict
    from models.yolo_test import Model  # the reference

    torch.set_num_threads(os.cpu_count())
    for name, cfg_name, b, h, w, fused, seed in CASES:
        cfg = named_config(cfg_name)
        torch.manual_seed(0)
        model = Model(cfg).eval()
        model.load_state_dict(seeded_state_dict(model.state_dict(), seed))
        if fused:
            model.fuse()
        rgb, ir = seeded_inputs(b, h, w, seed)
        taps = {}
        hooks = []
        for i, m in enumerate(model.model):
            if i in model.save and type(m).__name__ not in ("Detect", "GPT"):
                hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: taps.__setitem__(i, tap_stats(out))))
        with torch.no_grad():
            pred, raw = model(rgb, ir)
        for hk in hooks:
            hk.remove()
        blob = {
            "case": dict(name=name, cfg=cfg_name, batch=b, height=h, width=w, fused=fused, seed=seed),
            "torch": torch.__version__,
            "pred": pred.clone(), "raw": [r.clone() for r in raw], "taps": taps,
            "sha_rgb": hashlib.sha256(rgb.numpy().tobytes()).hexdigest(),
        }
        path = os.path.join(HERE, name + ".pt")
        torch.save(blob, path)
        print(f"{name:18s} pred {tuple(pred.shape)} raw std {raw[0].std():.3f} -> {os.path.getsize(path)/1e3:.0f} kB")


def make_nms_golden():
    """Reference `utils.general.non_max_suppression` on golden predictions.  torchvision is absent here, so
    its one call (`torchvision.ops.nms`, utils/general.py:527) is bound to the restatement of its published
    algorithm in oracle/nms_oracle.py; every other line executed is the reference's."""
    install_reference()
    sys.path.insert(0, ROOT)
    from oracle.nms_oracle import greedy_nms
    sys.modules["torchvision"].ops = types.SimpleNamespace(nms=greedy_nms)
    from utils.general import non_max_suppression  # the reference
    cases = []
    for src, kw in [("s_x3_rect", dict(conf_thres=0.25, iou_thres=0.45)),
                    ("s_x3_rect", dict(conf_thres=0.05, iou_thres=0.6, multi_label=True)),
                    ("s_x3_rect", dict(conf_thres=0.3, iou_thres=0.45, agnostic=True)),
                    ("s_x3_rect", dict(conf_thres=0.25, iou_thres=0.45, classes=[1, 4, 7])),
                    ("l_x3_llvip_192", dict(conf_thres=0.25, iou_thres=0.45, multi_label=True)),   # nc = 1
                    ("l_x3_flir_256", dict(conf_thres=0.9, iou_thres=0.45))]:                      # few / no survivors
        pred = torch.load(os.path.join(HERE, src + ".pt"), weights_only=False)["pred"]
        out = non_max_suppression(pred.clone(), **kw)
        cases.append({"source": src, "kwargs": kw, "out": [o.clone() for o in out]})
        print("nms", src, kw, [tuple(o.shape) for o in out])
    torch.save(cases, os.path.join(HERE, "nms_cases.pt"))


def make_checkpoint():
    """A checkpoint exactly as the reference's train.py:850-860 writes one - the whole ``nn.Module`` pickled,
    ``.half()``, inside a dict with ``ema`` - for a deliberately tiny two-stream CFTx3 network (width 0.125:
    8..128 channels (hidden C3 widths down to 8, the smallest the 16-byte-granule kernels take), GPT widths 32/64/128, i.e. head widths 4/8/16 that the attention kernel zero-pads).
    ``ref_ckpt_tiny.pt`` holds nothing but what the reference's own classes pickle; ``ref_ckpt_tiny_out.pt`` holds
    the reference's outputs for it (after ``attempt_load``'s ``.float().fuse().eval()``,
    models/experimental.py:119) on seeded inputs.  The GPU box has no /root/reference: the -m gpu test
    un-pickles this file through ``compat.attempt_load``."""
    install_reference()
    sys.path.insert(0, ROOT)
    import copy
    import msod_amd  # noqa: F401
    from msod_amd.models.configs import named_config
    from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict
    from models.yolo_test import Model  # the reference
    cfg = copy.deepcopy(named_config("yolov5s_fusion_transformerx3_vedai"))
    cfg["width_multiple"], cfg["nc"] = 0.125, 3
    torch.manual_seed(0)
    model = Model(cfg).eval()
    model.load_state_dict(seeded_state_dict(model.state_dict(), 9))
    model.names = ["person", "car", "bicycle"]
    ck = {"epoch": 7, "best_fitness": 0.5, "training_results": None, "model": copy.deepcopy(model).half(),
          "ema": None, "optimizer": None, "wandb_id": None}
    path = os.path.join(HERE, "ref_ckpt_tiny.pt")
    torch.save(ck, path)
    loaded = torch.load(path, map_location="cpu", weights_only=False)["model"].float().fuse().eval()
    rgb, ir = seeded_inputs(2, 96, 128, 9)
    with torch.no_grad():
        pred, raw = loaded(rgb, ir)
    torch.save({"cfg": cfg, "seed": 9, "batch": 2, "height": 96, "width": 128, "pred": pred.clone(), "raw": [r.clone() for r in raw],
                "names": loaded.names, "stride": loaded.stride.clone()}, os.path.join(HERE, "ref_ckpt_tiny_out.pt"))
    print(f"ref_ckpt_tiny.pt {os.path.getsize(path)/1e3:.0f} kB, pred {tuple(pred.shape)}, raw std {raw[0].std():.3f}")


def make_train_golden():
    """The reference's own TRAINING-mode forward (model.train(): BatchNorm batch statistics + running-stat updates,
    Detect returning the raw list, models/yolo_test.py:59) with every nn.Dropout probability set to 0 (torch's dropout
    RNG stream is not reproducible elsewhere); pins oracle/cft_oracle.py's train=True path."""
    install_reference()
    sys.path.insert(0, ROOT)
    import msod_amd  # noqa: F401
    from msod_amd.models.configs import named_config
    from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict
    from models.yolo_test import Model  # the reference
    cfg_name, b, h, w, seed = "yolov5s_fusion_transformerx3_vedai", 2, 96, 128, 8
    cfg = named_config(cfg_name)
    torch.manual_seed(0)
    model = Model(cfg)
    model.load_state_dict(seeded_state_dict(model.state_dict(), seed))
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    rgb, ir = seeded_inputs(b, h, w, seed)
    with torch.no_grad():
        raws = model(rgb, ir)
    assert isinstance(raws, list) and len(raws) == 3
    stats = {k: v.clone() for k, v in model.state_dict().items() if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    path = os.path.join(HERE, "s_x3_train_96.pt")
    torch.save({"case": dict(name="s_x3_train_96", cfg=cfg_name, batch=b, height=h, width=w, seed=seed, train=True),
                "torch": torch.__version__, "raw": [r.clone() for r in raws], "stats": stats}, path)
    print(f"s_x3_train_96 raw {[tuple(r.shape) for r in raws]} std {raws[0].std():.3f}, {len(stats)} statistics -> {os.path.getsize(path)/1e3:.0f} kB")


def make_letterbox_golden():
    """The reference's own `letterbox` (utils/datasets.py:1698-1728) on seeded uint8 images; cv2 is absent, so its two
    calls (`cv2.resize`, `cv2.copyMakeBorder`) are bound to the restatements in oracle/letterbox_oracle.py; every other
    line executed - the geometry and the composition - is the reference's."""
    install_reference()
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import letterbox_oracle as LO
    cv2 = sys.modules["cv2"]
    cv2.resize, cv2.copyMakeBorder = LO.resize, LO.copyMakeBorder
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT = LO.INTER_LINEAR, LO.BORDER_CONSTANT
    from utils.datasets import letterbox  # the reference
    rng = np.random.RandomState(0)
    cases = []
    for (h, w), kw in [((120, 160), dict(new_shape=160)), ((128, 160), dict(new_shape=(160, 160), auto=False)),
                       ((256, 320), dict(new_shape=160, auto=False, scaleup=False)), ((75, 50), dict(new_shape=104, stride=8)),
                       ((50, 83), dict(new_shape=(96, 192), auto=False, scaleup=True)), ((64, 64), dict(new_shape=64)),
                       ((37, 51), dict(new_shape=96, scaleFill=True, auto=False)), ((90, 120), dict(new_shape=64, auto=True, stride=32))]:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        out, ratio, pad = letterbox(img, **kw)
        cases.append({"img": torch.from_numpy(img), "kwargs": kw, "out": torch.from_numpy(np.ascontiguousarray(out)),
                      "ratio": tuple(float(r) for r in ratio), "pad": tuple(float(p) for p in pad)})
        print("letterbox", (h, w), kw, "->", out.shape, ratio, pad)
    torch.save(cases, os.path.join(HERE, "letterbox_cases.pt"))


# name, config name, batch, H, W, fused, seed - weights from utils/seeded.default_init_state_dict (the reference constructor's
# distributions): the regime in which north_star's 1e-2 bf16 bound is asserted outright
DINIT_CASES = [
    ("dinit_l_x3_flir_256", "cfg3", 1, 256, 256, True, 0),
    ("dinit_s_x3_rect", "yolov5s_fusion_transformerx3_vedai", 2, 192, 320, False, 1),
    ("dinit_x_x3_256", "cfg5", 1, 256, 256, True, 2),
    ("dinit_s_x4_256", "yolov5s_fusion_transformer_vedai", 1, 256, 256, True, 3),
]


def make_lowp_golden():
    """The reference's OWN low-precision forward: its unmodified ``Model`` under ``torch.autocast("cpu", bfloat16)`` (what
    train.py:755 ``amp.autocast`` does on a GPU; CPU autocast only offers bf16) on the same seeded weights / inputs
    as the fp32 goldens.  Written to tests/golden/lowp_ref.pt: {case: raw_bf16 list}; and, for the DINIT_CASES (reference
    constructor weight distributions), the fp32 outputs too.  This pins the 16-bit bound to the reference instead of to a
    storage model of ours: the HIP bf16 path must be at least as close to the reference's fp32 output as the
    reference's own bf16 forward is."""
    install_reference()
    sys.path.insert(0, ROOT)
    import msod_amd  # noqa: F401
    from msod_amd.models.configs import named_config
    from msod_amd.utils.seeded import default_init_state_dict, seeded_inputs, seeded_state_dict
    from models.yolo_test import Model  # the reference
    torch.set_num_threads(os.cpu_count())
    out = {}
    for name, cfg_name, b, h, w, fused, seed in CASES + DINIT_CASES:
        dinit = name.startswith("dinit_")
        cfg = named_config(cfg_name)
        torch.manual_seed(0)
        model = Model(cfg).eval()
        sd = (default_init_state_dict if dinit else seeded_state_dict)(model.state_dict(), seed)
        model.load_state_dict(sd)
        if fused:
            model.fuse()
        rgb, ir = seeded_inputs(b, h, w, seed)
        with torch.no_grad():
            pred, raw = model(rgb, ir)
            pred, raw = pred.clone(), [r.clone() for r in raw]      # Detect mutates / reuses its list
            with torch.autocast("cpu", dtype=torch.bfloat16):
                _, raw16 = model(rgb, ir)
        assert all(r.dtype == torch.bfloat16 for r in raw16)
        store16 = [r.clone() for r in raw16]                         # kept in bf16 (lossless, half the bytes)
        raw16 = [r.float() for r in raw16]
        flat = lambda rs: torch.cat([r.reshape(-1) for r in rs])    # noqa: E731
        sig = (flat(raw16).sigmoid() - flat(raw).sigmoid()).abs()
        rec = {"case": dict(name=name, cfg=cfg_name, batch=b, height=h, width=w, fused=fused, seed=seed, dinit=dinit),
               "raw_bf16": store16, "sig_err_bf16": sig.max().item(), "torch": torch.__version__}
        if dinit:
            rec["raw"] = raw             # pred is a function of raw (Detect decode); the oracle supplies it
        else:       # must be the forward the fp32 golden holds
            g = torch.load(os.path.join(HERE, name + ".pt"), weights_only=False)
            assert all(torch.equal(a, c) for a, c in zip(raw, g["raw"])), name
        out[name] = rec
        print(f"{name:22s} reference bf16-autocast vs its fp32: max sigmoid-space err {sig.max().item():.3e}, "
              f"rms logit err {(flat(raw16) - flat(raw)).pow(2).mean().sqrt().item():.3e}, logit std {flat(raw).std().item():.3f}")
    path = os.path.join(HERE, "lowp_ref.pt")
    torch.save(out, path)
    print(f"lowp_ref.pt {os.path.getsize(path) / 1e3:.0f} kB")


# name, config name, batch, H, W, fused - the weights SURVEY.md section 8c / BASELINE.md section 2 literally prescribe
# (utils/seeded.survey_state_dict: torch.manual_seed(0) constructor weights + randomised BatchNorm / pos_emb)
SURVEY_CASES = [
    ("survey_l_x3_flir_256", "cfg3", 1, 256, 256, True),
    ("survey_l_x3_flir_unfused_192", "cfg3", 1, 192, 192, False),
    ("survey_s_1cft_256", "cfg2", 2, 256, 256, True),
]


def make_survey_golden():
    """SURVEY.md section 8c's golden-vector plan, literally: ``torch.manual_seed(0)``, the REFERENCE's own
    ``Model(cfg)`` constructor, BatchNorm statistics / affine and ``pos_emb`` randomised, ``torch.rand``-range inputs; the
    reference's fp32 forward and its bf16-autocast forward.  tests/golden/survey_ref.pt = {case: raw, raw_bf16, fingerprint of the
    state dict}: the fingerprint lets the GPU box check that this package's constructor reproduced the reference's weights."""
    install_reference()
    sys.path.insert(0, ROOT)
    import msod_amd  # noqa: F401
    from msod_amd.models.configs import named_config
    from msod_amd.utils.seeded import seeded_inputs, state_dict_fingerprint, survey_state_dict
    from models.yolo_test import Model  # the reference
    torch.set_num_threads(os.cpu_count())
    out = {}
    for name, cfg_name, b, h, w, fused in SURVEY_CASES:
        cfg = named_config(cfg_name)
        sd = survey_state_dict(lambda: Model(cfg), seed=0)        # the reference's constructor draws the weights
        torch.manual_seed(0)
        model = Model(cfg).eval()
        model.load_state_dict(sd)
        if fused:
            model.fuse()
        rgb, ir = seeded_inputs(b, h, w, 0)
        with torch.no_grad():
            pred, raw = model(rgb, ir)
            pred, raw = pred.clone(), [r.clone() for r in raw]
            with torch.autocast("cpu", dtype=torch.bfloat16):
                _, raw16 = model(rgb, ir)
        flat = lambda rs: torch.cat([r.float().reshape(-1) for r in rs])    # noqa: E731
        sig = (flat(raw16).sigmoid() - flat(raw).sigmoid()).abs()
        out[name] = {"case": dict(name=name, cfg=cfg_name, batch=b, height=h, width=w, fused=fused, seed=0),
                     "raw": raw, "raw_bf16": [r.clone() for r in raw16], "sig_err_bf16": sig.max().item(),
                     "fingerprint": state_dict_fingerprint(sd), "torch": str(torch.__version__)}
        print(f"{name:30s} logit std {flat(raw).std().item():.3f}; reference bf16-autocast vs its fp32: {sig.max().item():.3e}; sd {out[name]['fingerprint'][:16]}")
    path = os.path.join(HERE, "survey_ref.pt")
    torch.save(out, path)
    print(f"survey_ref.pt {os.path.getsize(path) / 1e3:.0f} kB")


def make_ladder_golden():
    """The gain ladder (utils/seeded.LADDER_GAINS), cfg3 at 256 x 256, fused, one pair: for every rung the REFERENCE's
    fp32 forward, its own bf16-autocast forward, and the input sensitivity - the change of its fp32 output when the image pair is
    replaced by another one (rms and max, in logit and in sigmoid space).  tests/golden/ladder_ref.pt = {gain: record}.  A rung's 16-bit
    bound is a statement about the kernels only where the sensitivity is well above the bound (the output depends on what the
    kernels compute); the test (tests/test_gpu_model.py) asserts 1e-2 outright on the rungs where the reference's own bf16 forward
    meets it and the sensitivity is >= 1e-2, and <= 1.3 x the reference's own bf16 error above."""
    install_reference()
    sys.path.insert(0, ROOT)
    import msod_amd  # noqa: F401
    from msod_amd.models.configs import named_config
    from msod_amd.utils.seeded import LADDER_GAINS, ladder_state_dict, seeded_inputs
    from models.yolo_test import Model  # the reference
    torch.set_num_threads(os.cpu_count())
    cfg_name, b, h, w, seed = "cfg3", 1, 256, 256, 5
    cfg = named_config(cfg_name)
    flat = lambda rs: torch.cat([r.float().reshape(-1) for r in rs])    # noqa: E731
    out = {}
    for gain in LADDER_GAINS:
        torch.manual_seed(0)
        model = Model(cfg).eval()
        model.load_state_dict(ladder_state_dict(model.state_dict(), gain, seed))
        model.fuse()
        rgb, ir = seeded_inputs(b, h, w, seed)
        rgb2, ir2 = seeded_inputs(b, h, w, seed + 100)
        with torch.no_grad():
            _, raw = model(rgb, ir)
            raw = [r.clone() for r in raw]
            _, raw2 = model(rgb2, ir2)
            raw2 = [r.clone() for r in raw2]
            with torch.autocast("cpu", dtype=torch.bfloat16):
                _, raw16 = model(rgb, ir)
        raw16 = [r.clone() for r in raw16]
        d = flat(raw2) - flat(raw)
        ds = flat(raw2).sigmoid() - flat(raw).sigmoid()
        e16 = (flat(raw16).sigmoid() - flat(raw).sigmoid()).abs()
        rec = {"case": dict(cfg=cfg_name, batch=b, height=h, width=w, fused=True, seed=seed, gain=gain), "torch": str(torch.__version__),
               "raw": raw, "raw_bf16": raw16, "sig_err_bf16": e16.max().item(),
               "logit_std": flat(raw).std().item(),
               "sens_logit_rms": d.pow(2).mean().sqrt().item(), "sens_logit_max": d.abs().max().item(),
               "sens_sig_rms": ds.pow(2).mean().sqrt().item(), "sens_sig_max": ds.abs().max().item()}
        out[gain] = rec
        print(f"gain {gain:4.2f}: logit std {rec['logit_std']:.3f}; input sensitivity logit rms {rec['sens_logit_rms']:.3e} max {rec['sens_logit_max']:.3e}, "
              f"sigmoid rms {rec['sens_sig_rms']:.3e} max {rec['sens_sig_max']:.3e}; reference bf16-autocast vs its fp32 (sigmoid space) {rec['sig_err_bf16']:.3e}")
    path = os.path.join(HERE, "ladder_ref.pt")
    torch.save(out, path)
    print(f"ladder_ref.pt {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ladder":
        make_ladder_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "lowp":
        make_lowp_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "survey":
        make_survey_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "letterbox":
        make_letterbox_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "train":
        make_train_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "nms":
        make_nms_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "ckpt":
        make_checkpoint()
    else:
        main()
        make_nms_golden()
        make_checkpoint()
        make_train_golden()
        make_letterbox_golden()
        make_lowp_golden()
        make_survey_golden()
