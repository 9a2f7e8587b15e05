"""CPU oracle for the letterbox row.  TEST INFRASTRUCTURE ONLY.

The reference's ``letterbox`` (utils/datasets.py:1698-1728) is its own arithmetic around two calls into OpenCV,
``cv2.resize(img, new_unpad, interpolation=cv2.INTER_LINEAR)`` and ``cv2.copyMakeBorder(..., BORDER_CONSTANT)``.  cv2
is a third-party dependency that is absent from this image and unpinned by the reference (``opencv-python>=4.1.2``,
requirements.txt); its published 8-bit bilinear algorithm (modules/imgproc/src/resize.cpp: fixed-point coefficients with
INTER_RESIZE_COEF_BITS = 11, ``HResizeLinear`` in int, ``VResizeLinear`` as
``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``) is restated here in numpy.

Pinning: ``tests/golden/letterbox_cases.pt`` is produced by the REFERENCE's own ``letterbox`` executed in the build
container with ``cv2.resize`` / ``cv2.copyMakeBorder`` bound to the two functions below (tests/golden/make_golden.py
``letterbox``), i.e. the geometry and the composition are the reference's code; the two third-party calls themselves are
"parity unpinned" (as with torchvision.ops.nms in nms_oracle.py)."""
import numpy as np

INTER_LINEAR = 1
BORDER_CONSTANT = 0


def _coeffs(dst, src):
    inv = float(dst) / float(src)
    scale = 1.0 / inv
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def resize(img, dsize, interpolation=INTER_LINEAR):
    """cv2.resize for uint8 HWC images, INTER_LINEAR.  dsize = (width, height)."""
    assert interpolation == INTER_LINEAR and img.dtype == np.uint8
    sh, sw = img.shape[:2]
    dw, dh = dsize
    sx, fx = _coeffs(dw, sw)
    lo = sx < 0
    fx[lo] = 0.0
    sx[lo] = 0
    hi = sx >= sw - 1
    fx[hi] = 0.0
    sx[hi] = sw - 1
    sx1 = np.minimum(sx + 1, sw - 1)
    a0 = np.rint((np.float32(1.0) - fx) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048.0)).astype(np.int64)
    sy, fy = _coeffs(dh, sh)
    sy0 = np.clip(sy, 0, sh - 1)
    sy1 = np.clip(sy + 1, 0, sh - 1)
    b0 = np.rint((np.float32(1.0) - fy) * np.float32(2048.0)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048.0)).astype(np.int64)
    src = img.astype(np.int64)
    h = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]        # [sh, dw, c], scale 2^11
    r0, r1 = h[sy0], h[sy1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def copyMakeBorder(img, top, bottom, left, right, borderType=BORDER_CONSTANT, value=(0, 0, 0)):
    assert borderType == BORDER_CONSTANT
    h, w = img.shape[:2]
    out = np.empty((h + top + bottom, w + left + right, img.shape[2]), dtype=img.dtype)
    out[...] = np.asarray(value, dtype=img.dtype)[None, None, :img.shape[2]]
    out[top:top + h, left:left + w] = img
    return out
