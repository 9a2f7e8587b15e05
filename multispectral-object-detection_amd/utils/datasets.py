"""Input pre-processing that sits directly in front of the forward in every reference caller:
``letterbox`` of the reference's ``utils/datasets.py:1698-1728`` - resize to a stride-friendly shape keeping the
aspect ratio, pad with grey - on the device, one kernel per image (``cft_letterbox_u8``), and the pair packer that
builds the uint8 ``[B,6,H,W]`` batch the model consumes (``utils/datasets.py:1274-1281``: BGR->RGB, HWC->CHW)."""
import numpy as np
import torch

from .. import _lib
from ..ops import _require_cuda, _stream


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scaleFill=False, scaleup=True, stride=32):
    """The arithmetic of reference utils/datasets.py:1700-1726, verbatim in meaning: returns
    (new_unpad (w, h), ratio (w, h), (dw, dh) per side, (top, bottom, left, right))."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad, ratio, (dw, dh), (top, bottom, left, right)


def letterbox(img, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32, out=None,
              chw_rgb=False):
    """Same signature and return value as the reference's ``letterbox``: ``img`` is an HWC uint8 image (cv2 order, BGR)
    as a CUDA tensor; returns ``(img, ratio, (dw, dh))`` with ``img`` the letterboxed HWC uint8 CUDA tensor.
    ``chw_rgb=True`` (or ``out=`` a [3,H,W] view) writes the CHW RGB plane instead - the layout the callers build next
    (``img[:, :, ::-1].transpose(2, 0, 1)``), e.g. straight into a slice of the [B,6,H,W] pair batch."""
    _require_cuda(img, "letterbox")
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or img.stride(2) != 1 or img.stride(1) != 3:
        raise ValueError("letterbox: expected an HWC uint8 image with contiguous pixels")
    sh, sw = int(img.shape[0]), int(img.shape[1])
    new_unpad, ratio, (dw, dh), (top, bottom, left, right) = letterbox_geometry((sh, sw), new_shape, auto, scaleFill, scaleup, stride)
    rw, rh = new_unpad
    H, W = rh + top + bottom, rw + left + right
    if out is None:
        out = torch.empty((3, H, W) if chw_rgb else (H, W, 3), dtype=torch.uint8, device=img.device)
    if tuple(out.shape) == (3, H, W):
        sy, sx, sc, flip = out.stride(1), out.stride(2), out.stride(0), 1
    elif tuple(out.shape) == (H, W, 3):
        sy, sx, sc, flip = out.stride(0), out.stride(1), out.stride(2), 0
    else:
        raise ValueError(f"letterbox: out has shape {tuple(out.shape)}, expected {(H, W, 3)} or {(3, H, W)}")
    st = _lib.load().cft_letterbox_u8(img.data_ptr(), sh, sw, img.stride(0), out.data_ptr(), H, W, sy, sx, sc, flip,
                                      rh, rw, top, left, int(color[0]), int(color[1]), int(color[2]), _stream())
    _lib.check(st, "cft_letterbox_u8")
    return out, ratio, (dw, dh)


def letterbox_pair(img_rgb, img_ir, new_shape=640, stride=32, auto=False, scaleup=False, out=None):
    """One RGB + one IR image (HWC BGR uint8, the pair of utils/datasets.py:1206-1207) -> the uint8 [6,H,W] block of the
    batch (RGB plane 0-2, IR plane 3-5), both letterboxed to the same shape, BGR->RGB and HWC->CHW fused into the kernel."""
    new_unpad, _, _, (top, bottom, left, right) = letterbox_geometry(tuple(img_rgb.shape[:2]), new_shape, auto, False, scaleup, stride)
    H, W = new_unpad[1] + top + bottom, new_unpad[0] + left + right
    if out is None:
        out = torch.empty((6, H, W), dtype=torch.uint8, device=img_rgb.device)
    _, ratio, pad = letterbox(img_rgb, new_shape, auto=auto, scaleup=scaleup, stride=stride, out=out[:3])
    letterbox(img_ir, new_shape, auto=auto, scaleup=scaleup, stride=stride, out=out[3:])
    return out, ratio, pad
