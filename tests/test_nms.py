"""NMS row: oracle vs the reference-generated golden (CPU), HIP kernel vs oracle (GPU)."""
import os

import pytest
import torch

import msod_amd  # noqa: F401
from oracle import nms_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = torch.load(os.path.join(HERE, "golden", "nms_cases.pt"), weights_only=False)


def _pred(src):
    return torch.load(os.path.join(HERE, "golden", src + ".pt"), weights_only=False)["pred"]


@pytest.mark.parametrize("i", range(len(CASES)))
def test_nms_oracle_matches_reference_golden(i):
    c = CASES[i]
    out = nms_oracle.non_max_suppression(_pred(c["source"]), **c["kwargs"])
    assert len(out) == len(c["out"])
    for a, b in zip(out, c["out"]):
        assert a.shape == b.shape and torch.equal(a, b)


def test_greedy_nms_basics():
    boxes = torch.tensor([[0., 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    assert nms_oracle.greedy_nms(boxes, scores, 0.5).tolist() == [3, 2]
    assert nms_oracle.greedy_nms(boxes, scores, 0.99).tolist() == [3, 0, 1, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(CASES)))
def test_hip_nms_matches_oracle(dev, i):
    """Same detections, same order; boxes/scores are copies of the inputs so they must be bit-equal."""
    from msod_amd.utils.general import non_max_suppression
    c = CASES[i]
    pred = _pred(c["source"])
    want = nms_oracle.non_max_suppression(pred, **c["kwargs"])
    got = non_max_suppression(pred.to(dev), **c["kwargs"])
    assert len(got) == len(want)
    for a, b in zip(got, want):
        a = a.cpu()
        assert a.shape == b.shape, (a.shape, b.shape)
        assert torch.allclose(a, b, rtol=0, atol=1e-4), (a - b).abs().max()


@pytest.mark.gpu
def test_hip_nms_edge_cases(dev):
    from msod_amd.utils.general import batched_nms, non_max_suppression
    # no candidate at all; exactly max_det identical boxes of different classes (class offset keeps them apart)
    empty = torch.zeros(2, 50, 8, device=dev)
    assert [tuple(o.shape) for o in non_max_suppression(empty)] == [(0, 6), (0, 6)]
    p = torch.zeros(1, 400, 405, device=dev)
    p[0, :, 0:2] = 100.0
    p[0, :, 2:4] = 20.0
    p[0, :, 4] = 0.9
    p[0, torch.arange(400), 5 + torch.arange(400)] = torch.linspace(0.99, 0.5, 400, device=dev)
    dets, counts = batched_nms(p, 0.25, 0.45)
    torch.cuda.synchronize()
    assert int(counts[0]) == 300                                  # max_det
    assert torch.equal(dets[0, :, 5].cpu(), torch.arange(300).float())   # descending confidence = ascending class here
    dets, counts = batched_nms(p, 0.25, 0.45, agnostic=True)
    assert int(counts[0]) == 1                                    # identical boxes collapse when class-agnostic
