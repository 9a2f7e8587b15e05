"""Timing of the batched NMS kernel on the bench workload (run on the GPU box; writes gpurun_out/nms_bench.json).

Predictions: one real forward of yolov5l + CFTx3 (seeded weights) at 640x640, 64 pairs -> [64, 25200, 8].  Seeded
weights give object confidences spread over (0, 1), i.e. far MORE candidates than a trained detector produces, so
these are upper bounds.  conf 0.25 = detect_twostream.py's default, 0.001 = test.py's mAP setting.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd.models.configs import named_config  # noqa: E402
from msod_amd.models.yolo_test import Model  # noqa: E402
from msod_amd.utils.general import batched_nms  # noqa: E402
from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    model = Model(named_config("cfg3"))
    model.load_state_dict(seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev).fuse().set_compute_dtype(torch.float16)
    rgb, ir = seeded_inputs(B, 640, 640, 0)
    with torch.no_grad():
        pred, _ = model(rgb.to(dev), ir.to(dev))
    torch.cuda.synchronize()
    out = {"batch": B, "rows": pred.shape[1], "cases": []}
    for conf, multi in ((0.25, False), (0.001, False), (0.001, True)):
        dets, counts = batched_nms(pred, conf, 0.45, multi_label=multi)
        torch.cuda.synchronize()
        cand = int((pred[..., 4] > conf).sum().item()) / B
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 5
        e0.record()
        for _ in range(iters):
            batched_nms(pred, conf, 0.45, multi_label=multi)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        rec = {"conf_thres": conf, "multi_label": multi, "candidate_rows_per_image": cand, "kept_per_image": float(counts.float().mean()),
               "ms_per_batch": round(ms, 3), "us_per_image": round(ms * 1e3 / B, 1)}
        print(json.dumps(rec), flush=True)
        out["cases"].append(rec)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "nms_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
