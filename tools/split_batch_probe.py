"""Probe: does running the 64-pair step as several independent sub-batch graphs on concurrent HIP streams beat one graph?
(Kernels of one sub-batch fill the tails / partial rounds of the other's.)    python tools/split_batch_probe.py [parts ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd.graph import CapturedForward  # noqa: E402
from msod_amd.models.configs import named_config  # noqa: E402
from msod_amd.models.yolo_test import Model  # noqa: E402
from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict  # noqa: E402


def main():
    parts_list = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
    dev = torch.device("cuda:0")
    B, S = 64, 640
    model = Model(named_config("cfg3"))
    model.load_state_dict(seeded_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).fuse().set_compute_dtype(torch.bfloat16)
    rgb, ir = seeded_inputs(B, S, S, seed=0)
    rgb, ir = rgb.to(dev), ir.to(dev)
    res = {}
    ref = None
    for parts in parts_list:
        b = B // parts
        caps = [CapturedForward(model, b, S, S) for _ in range(parts)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
        for i, c in enumerate(caps):
            c.rgb.copy_(rgb[i * b:(i + 1) * b]); c.ir.copy_(ir[i * b:(i + 1) * b])

        def step():
            cur = torch.cuda.current_stream(dev)
            if parts == 1:
                caps[0].graph.replay()
                return
            for s_, c in zip(streams, caps):
                s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    c.graph.replay()
            for s_ in streams:
                cur.wait_stream(s_)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / 20
        pred = torch.cat([c.pred for c in caps], 0).float()
        if ref is None:
            ref = pred.clone()
        same = bool(torch.equal(pred, ref))
        res[parts] = {"ms_per_step": round(ms, 3), "pairs_per_s": round(B / ms * 1e3, 1), "identical_to_one_graph": same}
        print(parts, res[parts], flush=True)
        del caps
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "split_batch_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
