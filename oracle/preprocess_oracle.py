"""CPU ORACLE (test infrastructure, NOT product code) — image preprocessing.

Restates `/root/reference/vit_colmap/features/vit_extractor.py:117-132`:
BGR->RGB (cv2.cvtColor), floor to multiples of 14, `cv2.resize(..., INTER_LINEAR)` if the size
changed, `ToTensor` (/255) and `Normalize(ImageNet mean, std)`.

PARITY UNPINNED for the resize: OpenCV is not installed in this image, so `cv2.resize` cannot be
run; `resize_linear_u8` restates OpenCV's published 8-bit bilinear algorithm from memory
(half-pixel centres; float coefficients rounded to 11-bit fixed point with round-half-even;
int32 horizontal pass; vertical pass ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2).
The HIP kernel is tested bit-exact against THIS function.
"""
import numpy as np


def _coefs(dst: int, src: int):
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0
    s[hi] = src - 1
    s1 = np.minimum(s + 1, src - 1)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, s1, a0, a1


def resize_linear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 (h, w, c) -> uint8 (out_h, out_w, c), OpenCV INTER_LINEAR semantics [recalled]."""
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()
    x0, x1, ax0, ax1 = _coefs(out_w, w)
    y0, y1, ay0, ay1 = _coefs(out_h, h)
    src = img.astype(np.int64)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]     # (h, out_w, c)
    r0, r1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def preprocess(image_bgr: np.ndarray, patch: int = 14):
    """-> (float32 (3, h', w') normalised RGB, resized uint8 BGR frame)."""
    h, w = image_bgr.shape[:2]
    oh, ow = (h // patch) * patch, (w // patch) * patch
    resized = resize_linear_u8(image_bgr, oh, ow)
    rgb = resized[:, :, ::-1].astype(np.float32) / np.float32(255.0)
    x = ((rgb - MEAN) / STD).astype(np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1)), resized


def patchify(x: np.ndarray, patch: int = 14) -> np.ndarray:
    """(3, H, W) -> (Hp*Wp, 3*patch*patch) with element order (c, dy, dx)."""
    c, H, W = x.shape
    hp, wp = H // patch, W // patch
    return np.ascontiguousarray(x.reshape(c, hp, patch, wp, patch).transpose(1, 3, 0, 2, 4).reshape(hp * wp, -1))
