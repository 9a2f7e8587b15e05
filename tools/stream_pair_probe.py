"""Two forwards in flight: ms per step for every ordered pair of 8 torch pool streams (and the default stream).  HIP maps streams onto
a few hardware queues (pool index mod 4 here) and a captured forward brings internal branch streams of its own: pairs on queues
(0,1), (1,0), (3,1) run at 18.5-18.7 ms, every other pair at 19.6-20.3 ms (profiles/r03_forwards_in_flight.txt).  This is why
distributed.ForwardPipeline.pick_streams() probes a few stream groups before the timed region.  (run on the GPU box)"""
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
    def run1(stream, steps=12):
        with torch.cuda.stream(stream):
            caps[0].graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            with torch.cuda.stream(stream):
                caps[0].graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    default = torch.cuda.current_stream(dev)
    print("one forward in flight: default", round(run1(default), 2), "pool[j]", [round(run1(pool[j]), 2) for j in range(8)], flush=True)
    print("default+pool[j]", [round(run((default, pool[j])), 2) for j in range(8)], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stream_pair_matrix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
