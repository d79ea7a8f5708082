"""Image file I/O with OpenCV's conventions (BGR uint8, None on failure).

The reference uses cv2.imread / cv2.imwrite (vit_extractor.py:698,733; dummy_extractor.py:54,58).
OpenCV is used when importable; otherwise Pillow decodes, applies the EXIF orientation (cv2.imread's
IMREAD_COLOR default does, Pillow's decoder does not) and the channels are swapped.
"""
from pathlib import Path

import numpy as np

try:  # pragma: no cover - depends on the image
    import cv2 as _cv2
except Exception:  # noqa: BLE001
    _cv2 = None


def imread(path) -> "np.ndarray | None":
    if _cv2 is not None:
        return _cv2.imread(str(path))
    try:
        from PIL import Image, ImageOps

        with Image.open(str(path)) as im:
            rgb = np.asarray(ImageOps.exif_transpose(im).convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])
    except Exception:  # noqa: BLE001 - cv2.imread returns None for anything unreadable
        return None


def imwrite(path, bgr: np.ndarray) -> bool:
    if _cv2 is not None:
        return bool(_cv2.imwrite(str(path), bgr))
    from PIL import Image

    arr = np.asarray(bgr)
    if arr.ndim == 3:
        arr = arr[:, :, ::-1]
    Image.fromarray(np.ascontiguousarray(arr)).save(str(Path(path)))
    return True
