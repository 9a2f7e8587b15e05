"""HIP-graph capture of the whole two-stream forward.

yolov5l + CFTx3 is ~600 kernel launches per forward; launched one by one from Python the host
becomes the bottleneck well before the GPU does.  ``CapturedForward`` records the launches
once (``torch.cuda.CUDAGraph`` is a hipGraph on ROCm; our kernels are enqueued on torch's current
stream, so stream capture sees them) and replays them with a single ``hipGraphLaunch``.
All intermediate buffers come from the graph's private memory pool and stay resident.
"""
import torch


class CapturedForward:
    def __init__(self, model, batch, height, width, warmup=2, input_dtype=torch.float32):
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("capture: model must be on the GPU")
        self.rgb = torch.zeros((batch, 3, height, width), dtype=input_dtype, device=dev)
        self.ir = torch.zeros_like(self.rgb)
        model.prepare()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                  # first launches set kernel attributes, pack weights
                model.forward_once(self.rgb, self.ir, until_detect=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.pred, self.raw = model.forward_once(self.rgb, self.ir, until_detect=True)
        self.weights_key = model.weights_key()     # Model.forward drops the graph when the weights change

    def replay(self, rgb, ir):
        self.rgb.copy_(rgb, non_blocking=True)
        self.ir.copy_(ir, non_blocking=True)
        self.graph.replay()
        return self.pred, self.raw

    def replay_static(self):
        """Replay on whatever is already in the static input buffers (benchmark loop)."""
        self.graph.replay()
        return self.pred, self.raw
