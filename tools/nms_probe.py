import json, os, sys, torch
ROOT = "/root/repo" if os.path.isdir("/root/repo") else os.getcwd()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ROOT))
import msod_amd
from msod_amd.models.configs import named_config
from msod_amd.models.yolo_test import Model
from msod_amd.utils.general import batched_nms
from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict
dev = torch.device("cuda:0"); B = 64
model = Model(named_config("cfg3")); model.load_state_dict(seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).fuse().set_compute_dtype(torch.float16)
rgb, ir = seeded_inputs(B, 640, 640, 0)
with torch.no_grad():
    pred, _ = model(rgb.to(dev), ir.to(dev))
torch.cuda.synchronize()
def t(conf, max_det, iters=5, **kw):
    dets, counts = batched_nms(pred, conf, 0.45, max_det=max_det, **kw); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): batched_nms(pred, conf, 0.45, max_det=max_det, **kw)
    e1.record(); torch.cuda.synchronize()
    cand = float((pred[..., 4] > conf).sum()) / B
    print(json.dumps({"conf": conf, "max_det": max_det, "cand_per_img": cand, "kept": float(counts.float().mean()), "ms": round(e0.elapsed_time(e1) / iters, 3)}), flush=True)
obj = pred[..., 4].flatten()
q = torch.quantile(obj[torch.randperm(obj.numel(), device=dev)[:1000000]].float(), torch.tensor([1 - 2000 / 25200, 1 - 500 / 25200], device=dev))
for conf, md in ((0.25, 300), (0.25, 1), (0.25, 30), (0.25, 100), (float(q[0]), 300), (float(q[0]), 1), (float(q[1]), 300), (1.5, 300)):
    t(conf, md)
