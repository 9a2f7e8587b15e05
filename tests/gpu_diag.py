"""One-shot GPU diagnostics (not a pytest file): end-to-end error of the HIP model against the CPU
oracle for several configs and the three precisions, plus what 16-bit STORAGE alone costs on the same network
(oracle/lowp_oracle.py) and how close the HIP 16-bit result is to that storage model.  Writes gpurun_out/diag.json."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd.models.configs import named_config  # noqa: E402
from msod_amd.models.yolo_test import Model  # noqa: E402
from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict  # noqa: E402
from oracle.cft_oracle import OracleModel  # noqa: E402
from oracle.lowp_oracle import LowpOracle  # noqa: E402


def metrics(pred, raw, wpred, wraw):
    pred = pred.float().cpu()
    raws = torch.cat([r.float().cpu().reshape(-1) for r in raw])
    wraws = torch.cat([r.reshape(-1) for r in wraw])
    sig = (raws.sigmoid() - wraws.sigmoid()).abs()
    return {
        "raw_max_abs": (raws - wraws).abs().max().item(),
        "raw_rms": (raws - wraws).pow(2).mean().sqrt().item(),
        "raw_std": wraws.std().item(),
        "sigmoid_max_abs": sig.max().item(),
        "pred_max_abs": (pred - wpred).abs().max().item(),
        "pred_max_rel": ((pred - wpred).abs() / (wpred.abs() + 1.0)).max().item(),
        "conf_max_abs": (pred[..., 4:] - wpred[..., 4:]).abs().max().item(),
    }


def main():
    out = []
    cases = [("cfg1", 1, 320, 320), ("cfg2", 2, 256, 256), ("yolov5s_fusion_transformerx3_vedai", 2, 192, 320),
             ("yolov5s_fusion_transformer_vedai", 1, 256, 256), ("cfg3", 1, 256, 256), ("cfg3", 1, 640, 640), ("cfg5", 1, 640, 640)]
    for name, b, h, w in cases:
        cfg = named_config(name)
        model = Model(cfg)
        sd = seeded_state_dict(model.state_dict(), seed=3)
        model.load_state_dict(sd)
        rgb, ir = seeded_inputs(b, h, w, seed=3)
        t0 = time.time()
        wpred, wraw = OracleModel(cfg)(sd, rgb, ir)
        t_or = time.time() - t0
        rec = {"case": name, "shape": [b, h, w], "oracle_s": t_or}
        lowp = {}
        for dt in (torch.bfloat16, torch.float16):      # what 16-bit STORAGE alone costs (oracle/lowp_oracle.py)
            lp, lr = LowpOracle(cfg, dt)(sd, rgb, ir)
            lowp[dt] = (lp, lr)
            rec[f"storage_model_{dt}_vs_fp32"] = metrics(lp, lr, wpred, wraw)
        model = model.cuda()
        for dtype in (torch.float32, torch.float16, torch.bfloat16):
            model.set_compute_dtype(dtype)
            try:
                with torch.no_grad():
                    pred, raw = model(rgb.cuda(), ir.cuda())
                torch.cuda.synchronize()
                rec[str(dtype)] = metrics(pred, raw, wpred, wraw)
                if dtype in lowp:
                    rec[f"{dtype}_vs_storage_model"] = metrics(pred, raw, *lowp[dtype])
            except Exception as e:  # keep going: one report per run
                rec[str(dtype)] = {"error": repr(e)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del model
        torch.cuda.empty_cache()
    # the reference's OWN bf16 forward (tests/golden/lowp_ref.pt, recorded in the build container) next to the HIP bf16 / fp16 forward
    from msod_amd.utils.seeded import default_init_state_dict
    lowp_ref = torch.load(os.path.join(ROOT, "tests", "golden", "lowp_ref.pt"), weights_only=False)
    flat = lambda rs: torch.cat([r.float().cpu().reshape(-1) for r in rs])     # noqa: E731
    for name, rec0 in lowp_ref.items():
        c = rec0["case"]
        cfg = named_config(c["cfg"])
        model = Model(cfg)
        model.load_state_dict((default_init_state_dict if c["dinit"] else seeded_state_dict)(model.state_dict(), c["seed"]))
        if c["fused"]:
            model.fuse()
        rgb, ir = seeded_inputs(c["batch"], c["height"], c["width"], c["seed"])
        want = flat(rec0["raw"]) if c["dinit"] else flat(torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)["raw"])
        ref16 = flat(rec0["raw_bf16"])
        rec = {"lowp_case": name, "weights": "reference-constructor distributions" if c["dinit"] else "seeded (lively)",
               "reference_bf16_autocast": {"sigmoid_max_abs": (ref16.sigmoid() - want.sigmoid()).abs().max().item(),
                                           "rms_over_std": ((ref16 - want).pow(2).mean().sqrt() / want.std()).item()}}
        model = model.cuda()
        for dtype in (torch.bfloat16, torch.float16, torch.float32):
            model.set_compute_dtype(dtype)
            with torch.no_grad():
                _, raw = model(rgb.cuda(), ir.cuda())
            got = flat(raw)
            rec[f"hip_{str(dtype).split('.')[-1]}"] = {"sigmoid_max_abs": (got.sigmoid() - want.sigmoid()).abs().max().item(),
                                                      "raw_max_abs": (got - want).abs().max().item(),
                                                      "rms_over_std": ((got - want).pow(2).mean().sqrt() / want.std()).item()}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del model
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
