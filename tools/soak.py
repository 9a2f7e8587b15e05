"""Soak run of bench.py's default step mode: K single-stream captured forwards in flight at the benchmarked shape for N steps, every replay
compared bit for bit with the output of the first one (a missing wait or barrier in a hand-scheduled kernel would show up as a rare mismatch
under this co-scheduling).      python tools/soak.py [--steps 3000] [--in-flight 3] [--dtype bf16|f16] [--batch 64]"""
import argparse
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd import distributed as D  # noqa: E402
from msod_amd.graph import CapturedForward  # noqa: E402
from msod_amd.models.configs import named_config  # noqa: E402
from msod_amd.models.yolo_test import Model  # noqa: E402
from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--in-flight", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--check-every", type=int, default=150)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    cfg = named_config("cfg3")
    model = Model(cfg)
    model.load_state_dict(seeded_state_dict(model.state_dict(), 0))
    model = model.fuse().to(dev).set_compute_dtype(dtype)
    rgb, ir = seeded_inputs(args.batch, 640, 640, 0)
    x, x2 = rgb.to(dev), ir.to(dev)
    with torch.no_grad():
        model.overlap_streams = False
        caps = [CapturedForward(model, args.batch, 640, 640) for _ in range(args.in_flight)]
        for c in caps:
            c.rgb.copy_(x)
            c.ir.copy_(x2)
        caps[0].replay_static()
        torch.cuda.synchronize()
        pred0, raw0 = caps[0].pred.clone(), [r.clone() for r in caps[0].raw]
        streams = [torch.cuda.Stream(device=dev) for _ in caps]
        pipe = D.ForwardPipeline([(lambda c=c: c.replay_static()[0]) for c in caps], streams)
        bad, done, t0 = 0, 0, time.perf_counter()
        while done < args.steps:
            n = min(args.check_every, args.steps - done)
            for _ in range(n):
                pipe.step()
            torch.cuda.synchronize()
            done += n
            for i, c in enumerate(caps):
                same = torch.equal(c.pred, pred0) and all(torch.equal(a, b) for a, b in zip(c.raw, raw0))
                if not same:
                    bad += 1
                    print(f"MISMATCH after step {done}: graph {i}, max |diff| {(c.pred - pred0).abs().max().item():.3e}", flush=True)
        el = time.perf_counter() - t0
    print(f"soak: {done} steps of {args.batch} pairs, {args.in_flight} in flight, {args.dtype}: {bad} mismatching checks of {(done // args.check_every) * len(caps)}; "
          f"{args.batch * done / el:.0f} pairs/s incl. the checks; finite: {bool(torch.isfinite(pred0).all())}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
