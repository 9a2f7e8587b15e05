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
                max_nms=MAX_NMS, class_table=None):
    """prediction [B, rows, nc+5] (fp32, on the GPU) -> (dets [B, max_det, 6] = (x1,y1,x2,y2,conf,cls),
    counts [B] int32).  ``classes`` is ALWAYS a collection of class ids, as in the reference (:505-506; a list or a tensor
    on any device: one small H2D copy); ``class_table`` instead hands over a ready-made uint8 [nc] allow-table on the
    prediction's device (1 = keep the class) - no host work at all: fit for HIP-graph capture and for all-gathering the
    <= max_det survivors instead of all rows."""
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
    if class_table is not None:
        if classes is not None:
            raise ValueError("batched_nms: give either classes (ids) or class_table (allow-table), not both")
        if not (isinstance(class_table, torch.Tensor) and class_table.dtype == torch.uint8 and class_table.numel() == nc
                and class_table.device == prediction.device and class_table.is_contiguous()):
            raise ValueError(f"batched_nms: class_table must be a contiguous uint8 [{nc}] tensor on {prediction.device}")
        allow = class_table                   # the kernel indexes it by class id
    else:
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


def clip_coords(boxes, img_shape):
    """Clip xyxy boxes to the image (height, width), in place (reference utils/general.py:369-374)."""
    boxes[:, 0].clamp_(0, img_shape[1])
    boxes[:, 1].clamp_(0, img_shape[0])
    boxes[:, 2].clamp_(0, img_shape[1])
    boxes[:, 3].clamp_(0, img_shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """Map xyxy boxes from the letterboxed shape back to the original image, in place (reference utils/general.py:353-366)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords
