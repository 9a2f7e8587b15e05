"""Post-processing that sits directly behind the forward in every reference caller
(`test.py:129`, `detect_twostream.py:86`): `non_max_suppression` of the reference's
`utils/general.py:455-543`, as ONE batched HIP kernel instead of a Python loop over images around
`torchvision.ops.nms`."""
import torch

from .. import _lib
from ..ops import _require_cuda, _stream

MAX_WH = 4096  # class offset in pixels (reference utils/general.py:467)
MAX_NMS = 30000  # boxes that enter the suppression at most (reference utils/general.py:469)


def _class_table(classes, nc, device):
    """uint8 [nc] allow-table for the kernel (exact for any class id, reference :505-506), None = all classes."""
    if classes is None:
        return None
    t = torch.zeros((nc,), dtype=torch.uint8)
    for c in classes:
        if 0 <= int(c) < nc:
            t[int(c)] = 1
    return t.to(device)


def batched_nms(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300,
                max_nms=MAX_NMS):
    """prediction [B, rows, nc+5] (fp32, on the GPU) -> (dets [B, max_det, 6] = (x1,y1,x2,y2,conf,cls),
    counts [B] int32).  No host synchronisation (unless ``classes`` is given as a Python list: one small H2D copy):
    fit for HIP-graph capture and for all-gathering the <= max_det survivors instead of all rows."""
    _require_cuda(prediction, "batched_nms")
    if prediction.dtype != torch.float32 or not prediction.is_contiguous():
        prediction = prediction.float().contiguous()
    B, rows, no = prediction.shape
    nc = no - 5
    multi_label = bool(multi_label) and nc > 1            # reference :472
    cap = rows * (nc if multi_label else 1)
    scratch = torch.empty((B * ((cap + 3) // 4 * 4) * 32,), dtype=torch.uint8, device=prediction.device)
    dets = torch.empty((B, max_det, 6), dtype=torch.float32, device=prediction.device)     # cft_nms zeroes the unused rows itself
    counts = torch.empty((B,), dtype=torch.int32, device=prediction.device)                 # and always writes every count
    if isinstance(classes, torch.Tensor) and classes.dtype == torch.uint8 and classes.numel() == nc and classes.is_cuda \
            and classes.device == prediction.device and classes.is_contiguous():
        allow = classes                       # a ready-made [nc] allow-table on the device (the kernel indexes it by class id)
    else:                                     # anything else is a collection of class ids (reference :505-506)
        allow = _class_table(classes.tolist() if isinstance(classes, torch.Tensor) else classes, nc, prediction.device)
    st = _lib.load().cft_nms(prediction.data_ptr(), B, rows, no, float(conf_thres), float(iou_thres), int(bool(agnostic)),
                             int(multi_label), allow.data_ptr() if allow is not None else None, int(max_det), int(max_nms),
                             scratch.data_ptr(), scratch.numel(), dets.data_ptr(), counts.data_ptr(), _stream())
    _lib.check(st, "cft_nms")
    return dets, counts


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=()):
    """Same signature and return value as the reference: a list with one (n,6) tensor [xyxy, conf, cls] per
    image, sorted by descending confidence, n <= 300."""
    if labels and any(len(l) for l in labels):
        # autolabelling (reference :480-487): a-priori labels [cls, x, y, w, h] join the image's candidates as rows
        # with obj = 1 and a one-hot class; images with fewer labels get obj = 0 padding rows (filtered in phase 1)
        B, rows, no = prediction.shape
        L = max(len(l) for l in labels)
        extra = torch.zeros((B, L, no), dtype=torch.float32, device=prediction.device)
        for xi, l in enumerate(labels):
            if len(l):
                l = torch.as_tensor(l, dtype=torch.float32, device=prediction.device)
                extra[xi, :len(l), :4] = l[:, 1:5]
                extra[xi, :len(l), 4] = 1.0
                extra[xi, torch.arange(len(l), device=prediction.device), l[:, 0].long() + 5] = 1.0
        prediction = torch.cat((prediction.float(), extra), 1)
    dets, counts = batched_nms(prediction, conf_thres, iou_thres, classes, agnostic, multi_label)
    counts = counts.tolist()
    return [dets[i, :n] for i, n in enumerate(counts)]


def xywh2xyxy(x):
    """[x, y, w, h] -> [x1, y1, x2, y2] (reference utils/general.py:386-393); tiny, used by callers on results."""
    y = x.clone()
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y
