"""Experiment: two forwards in flight - does a phase offset between them (one in its backbone while the other is in its CFT blocks /
head) or a stream priority change the steady-state rate?  (run on the GPU box)"""
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
    dev = torch.device("cuda:0")
    model = Model(named_config("cfg3"))
    model.load_state_dict(seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev).fuse().set_compute_dtype(torch.bfloat16)
    rgb, ir = seeded_inputs(64, 640, 640, 0)
    with torch.no_grad():
        caps = [CapturedForward(model, 64, 640, 640) for _ in range(2)]
    for c in caps:
        c.rgb.copy_(rgb.to(dev))
        c.ir.copy_(ir.to(dev))
    torch.cuda.synchronize()

    pool = [torch.cuda.Stream(device=dev) for _ in range(8)]

    def run(streams, steps=12):
        for w in range(2):
            with torch.cuda.stream(streams[w & 1]):
                caps[w & 1].graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            with torch.cuda.stream(streams[t & 1]):
                caps[t & 1].graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    out = []
    for i in range(8):
        row = []
        for j in range(8):
            row.append(round(run((pool[i], pool[j])), 2) if i != j else None)
        print(i, row, flush=True)
        out.append(row)
    default = torch.cuda.current_stream(dev)
    print("default+pool[j]", [round(run((default, pool[j])), 2) for j in range(8)], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stream_pair_matrix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
