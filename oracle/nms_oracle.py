"""CPU oracle for the NMS row.  TEST INFRASTRUCTURE ONLY.

Restates `non_max_suppression` of the reference (utils/general.py:455-543) in plain torch.  The reference
delegates the suppression itself to `torchvision.ops.nms` (:527), a third-party dependency that is absent
from this image and unpinned by the reference (`torchvision>=0.8.1`, requirements.txt:11).  Its published
algorithm is restated in `greedy_nms`: visit boxes in order of decreasing score, keep a box iff its IoU with
every already-kept box is <= iou_threshold, return the kept indices in that order.

Pinning: `tests/golden/nms_*.pt` are produced by the REFERENCE's own `non_max_suppression` executed in the
build container with `torchvision.ops.nms` bound to `greedy_nms` (tests/golden/make_golden.py), i.e. everything
but the third-party call is the reference's code; the third-party call itself is "parity unpinned".
"""
import torch


def box_iou_1_to_n(box, boxes):
    iw = (torch.min(box[2], boxes[:, 2]) - torch.max(box[0], boxes[:, 0])).clamp(min=0)
    ih = (torch.min(box[3], boxes[:, 3]) - torch.max(box[1], boxes[:, 1])).clamp(min=0)
    inter = iw * ih
    a1 = (box[2] - box[0]) * (box[3] - box[1])
    a2 = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    return inter / (a1 + a2 - inter)


def greedy_nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms semantics."""
    order = torch.argsort(scores, descending=True, stable=True)
    alive = torch.ones(boxes.shape[0], dtype=torch.bool)
    keep = []
    for idx in order.tolist():
        if not alive[idx]:
            continue
        keep.append(idx)
        iou = box_iou_1_to_n(boxes[idx], boxes)
        alive &= ~(iou > iou_threshold)
        alive[idx] = False
    return torch.tensor(keep, dtype=torch.long)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_nms=30000):
    """reference utils/general.py:455-543 (merge-NMS branch, off in the reference, omitted).  ``max_nms`` is the
    reference's constant (:469), a parameter here only so that tests can reach the truncation with small inputs."""
    nc = prediction.shape[2] - 5
    xc = prediction[..., 4] > conf_thres
    max_wh, max_det = 4096, 300
    multi_label = multi_label and nc > 1
    output = [torch.zeros((0, 6))] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        x = x[xc[xi]].clone()
        if labels and len(labels[xi]):       # :480-487
            l = torch.as_tensor(labels[xi], dtype=torch.float32)
            v = torch.zeros((len(l), nc + 5))
            v[:, :4] = l[:, 1:5]
            v[:, 4] = 1.0
            v[range(len(l)), l[:, 0].long() + 5] = 1.0
            x = torch.cat((x, v), 0)
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]
        box = torch.stack((x[:, 0] - x[:, 2] / 2, x[:, 1] - x[:, 3] / 2, x[:, 0] + x[:, 2] / 2, x[:, 1] + x[:, 3] / 2), 1)
        if multi_label:
            i, j = (x[:, 5:] > conf_thres).nonzero(as_tuple=False).T
            x = torch.cat((box[i], x[i, j + 5, None], j[:, None].float()), 1)
        else:
            conf, j = x[:, 5:].max(1, keepdim=True)
            x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        if classes is not None:
            x = x[(x[:, 5:6] == torch.tensor(classes)).any(1)]
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:
            x = x[x[:, 4].argsort(descending=True, stable=True)[:max_nms]]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        i = greedy_nms(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        output[xi] = x[i]
    return output
