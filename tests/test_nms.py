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


def _random_pred(rows, nc, seed, spread=600.0):
    g = torch.Generator().manual_seed(seed)
    p = torch.zeros(1, rows, nc + 5)
    p[0, :, 0:2] = torch.rand(rows, 2, generator=g) * spread
    p[0, :, 2:4] = torch.rand(rows, 2, generator=g) * 40 + 8
    p[0, :, 4] = torch.rand(rows, generator=g) * 0.5 + 0.5
    p[0, :, 5:] = torch.rand(rows, nc, generator=g)
    return p


def test_nms_oracle_max_nms_and_labels_branches():
    """The oracle's restatement of the two reference branches the golden cases do not reach: the max_nms
    pre-truncation (utils/general.py:515-516) and the autolabelling rows (:480-487)."""
    p = _random_pred(300, 3, 1)
    full = nms_oracle.non_max_suppression(p, 0.05, 0.45, multi_label=True)[0]
    cut = nms_oracle.non_max_suppression(p, 0.05, 0.45, multi_label=True, max_nms=200)[0]
    assert cut.shape[0] <= full.shape[0] and cut[:, 4].min() >= full[:, 4].min()
    kth = torch.sort((p[0, :, 5:] * p[0, :, 4:5]).reshape(-1), descending=True)[0][199]
    assert cut[:, 4].min() >= kth
    lab = [torch.tensor([[2.0, 300.0, 300.0, 50.0, 60.0]])]
    out = nms_oracle.non_max_suppression(p, 0.25, 0.45, labels=lab)[0]
    assert out[0, 4] == 1.0 and out[0, 5] == 2.0 and torch.allclose(out[0, :4], torch.tensor([275.0, 270.0, 325.0, 330.0]))


@pytest.mark.gpu
def test_hip_nms_class_filter_is_exact_beyond_64_classes(dev):
    """`classes=` must filter exactly for any class id (80-class heads), reference :505-506."""
    from msod_amd.utils.general import non_max_suppression
    p = _random_pred(500, 80, 2)
    for classes in ([70], [3], [3, 64, 79], []):
        want = nms_oracle.non_max_suppression(p, 0.25, 0.45, classes=classes)[0]
        got = non_max_suppression(p.to(dev), 0.25, 0.45, classes=classes)[0].cpu()
        assert got.shape == want.shape and torch.allclose(got, want, rtol=0, atol=1e-4)
        assert all(int(c) in classes for c in got[:, 5].tolist())


@pytest.mark.gpu
def test_hip_nms_max_nms_truncation_and_labels(dev):
    from msod_amd.utils.general import batched_nms, non_max_suppression
    p = _random_pred(2000, 3, 3, spread=4000.0)            # spread out: little suppression, many survivors
    want = nms_oracle.non_max_suppression(p, 0.05, 0.45, multi_label=True, max_nms=1000)[0]
    dets, counts = batched_nms(p.to(dev), 0.05, 0.45, multi_label=True, max_nms=1000)
    got = dets[0, :int(counts[0])].cpu()
    assert got.shape == want.shape and torch.allclose(got, want, rtol=0, atol=1e-4)
    # the truncation really bites here: without it lower-confidence boxes survive
    dets2, counts2 = batched_nms(p.to(dev), 0.05, 0.45, multi_label=True, max_nms=0)
    assert int(counts2[0]) == 300 and int(counts[0]) == 300
    lab = [torch.tensor([[2.0, 300.0, 300.0, 50.0, 60.0], [0.0, 900.0, 100.0, 30.0, 30.0]])]
    p1 = _random_pred(300, 3, 4)
    want = nms_oracle.non_max_suppression(p1, 0.25, 0.45, labels=lab)[0]
    got = non_max_suppression(p1.to(dev), 0.25, 0.45, labels=lab)[0].cpu()
    assert got.shape == want.shape and torch.allclose(got[2:], want[2:], rtol=0, atol=1e-4)
    assert sorted(got[:2, 5].tolist()) == [0.0, 2.0] and (got[:2, 4] == 1.0).all()      # the two labels tie at conf 1.0


def _clustered_pred(n_clusters, per_cluster, nc=2, tie_scores=False):
    """Clusters of near-identical boxes (one survivor each); cluster c's scores sit in a band below cluster c-1's, so
    the survivors are spread over the whole score range: the kernel must walk several LDS chunks to find 300 of them."""
    g = torch.Generator().manual_seed(7)
    rows = n_clusters * per_cluster
    p = torch.zeros(1, rows, nc + 5)
    c = torch.arange(n_clusters).repeat_interleave(per_cluster)
    p[0, :, 0] = 40.0 + (c % 40) * 60.0 + torch.rand(rows, generator=g) * 2
    p[0, :, 1] = 40.0 + (c // 40) * 60.0 + torch.rand(rows, generator=g) * 2
    p[0, :, 2:4] = 30.0
    p[0, :, 4] = 0.9 if tie_scores else (0.99 - c.float() * 0.002 - torch.rand(rows, generator=g) * 0.0015)
    p[0, :, 5] = 1.0
    p[0, :, 6] = 0.1
    perm = torch.randperm(rows, generator=g)
    return p[:, perm].contiguous()


@pytest.mark.gpu
def test_hip_nms_walks_several_lds_chunks(dev):
    """> 2048 candidates and fewer than max_det survivors in the first chunk: later chunks are thinned by the boxes
    kept so far and continue the greedy order exactly."""
    from msod_amd.utils.general import batched_nms
    p = _clustered_pred(320, 20)                         # 6400 candidates, 320 clusters -> 300 kept, spread over all chunks
    want = nms_oracle.non_max_suppression(p, 0.25, 0.45)[0]
    dets, counts = batched_nms(p.to(dev), 0.25, 0.45)
    got = dets[0, :int(counts[0])].cpu()
    assert want.shape[0] == 300 and got.shape == want.shape and torch.equal(got, want)
    p2 = _clustered_pred(120, 40)                        # 4800 candidates, only 120 survivors: every chunk is visited
    want = nms_oracle.non_max_suppression(p2, 0.25, 0.45)[0]
    dets, counts = batched_nms(p2.to(dev), 0.25, 0.45)
    got = dets[0, :int(counts[0])].cpu()
    assert want.shape[0] == 120 and got.shape == want.shape and torch.equal(got, want)


@pytest.mark.gpu
def test_hip_nms_score_ties_overflowing_a_chunk(dev):
    """5000 candidates with IDENTICAL confidence: the chunk threshold ties with all of them, the LDS arrays cannot hold
    the chunk and the kernel falls back to rounds over global memory; ties resolve by row order like the oracle's
    stable sort."""
    from msod_amd.utils.general import batched_nms
    p = _clustered_pred(250, 20, tie_scores=True)
    want = nms_oracle.non_max_suppression(p, 0.25, 0.45)[0]
    dets, counts = batched_nms(p.to(dev), 0.25, 0.45)
    got = dets[0, :int(counts[0])].cpu()
    assert want.shape[0] == 250 and got.shape == want.shape and torch.equal(got, want)


@pytest.mark.gpu
def test_hip_nms_first_chunk_settles_a_crowded_image_and_batches_mix(dev):
    """Round-3 path (filter -> select + sort -> suppression bit matrix -> one-wave scan): 6000 candidates of which the best
    2048 already yield max_det survivors (no continuation), in one batch with a sparse image (< 2048 candidates: the chunk is
    everything) and an empty one; every image bit-equal to the oracle."""
    from msod_amd.utils.general import batched_nms
    crowded = _random_pred(6000, 3, 11, spread=3000.0)
    sparse = _random_pred(6000, 3, 12, spread=500.0)
    sparse[0, 200:, 4] = 0.0                                   # 200 candidates: fewer than max_det survivors
    empty = torch.zeros_like(crowded)
    batch = torch.cat([crowded, sparse, empty, crowded.flip(1)], 0)
    dets, counts = batched_nms(batch.to(dev), 0.25, 0.45)
    torch.cuda.synchronize()
    for b in range(4):
        want = nms_oracle.non_max_suppression(batch[b:b + 1], 0.25, 0.45)[0]
        got = dets[b, :int(counts[b])].cpu()
        assert got.shape == want.shape and torch.equal(got, want), b
    assert int(counts[0]) == 300 and 0 < int(counts[1]) < 300 and int(counts[2]) == 0
    assert torch.equal(dets[0, :300].cpu()[:, 4], dets[3, :300].cpu()[:, 4])      # row order of the input does not matter
