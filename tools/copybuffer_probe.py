"""Where do the `__amd_rocclr_copyBuffer` dispatches in the rocprof traces come from?  (VERDICT r1 hygiene item.)
Run one mode per process under `rocprofv3 --kernel-trace --stats`:  plain | events | streams | graph"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import msod_amd  # noqa: E402,F401
from msod_amd import ops  # noqa: E402

mode = sys.argv[1]
dev = torch.device("cuda:0")
a = ops.new_nhwc(2, 16, 16, 64, torch.bfloat16, dev)
b = ops.new_nhwc(2, 16, 16, 64, torch.bfloat16, dev)
a.zero_(); b.zero_()
out = ops.add(a, b)
torch.cuda.synchronize()
N = 200
if mode == "plain":
    for _ in range(N):
        ops.add(a, b, out=out)
elif mode == "events":
    evs = []
    for _ in range(N):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.add(a, b, out=out); e1.record()
        evs.append((e0, e1))
elif mode == "streams":
    s1 = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    for _ in range(N):
        s1.wait_stream(main)
        with torch.cuda.stream(s1):
            ops.add(a, b, out=out)
        main.wait_stream(s1)
        ops.add(a, b, out=out)
elif mode == "graph":
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.add(a, b, out=out)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.add(a, b, out=out)
    for _ in range(10):
        g.replay()
torch.cuda.synchronize()
print("done", mode)
