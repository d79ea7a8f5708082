"""Torch front end of csrc/preprocess.hip (reference vit_extractor.py:117-132, batched)."""
import torch

from .. import _lib

PATCH = 14


def preprocess(images_bgr: torch.Tensor, out_dtype=torch.bfloat16, layout: str = "patches", want_resized=False):
    """images_bgr uint8 (B, h, w, 3) on the GPU -> model input.
    layout "patches": (B, Hp*Wp, 588); "patches_pad": (B, Hp*Wp, 640) zero padded; "nchw": (B, 3, h', w') with h' = floor(h/14)*14."""
    if not images_bgr.is_cuda:
        raise _lib.HipLibraryError("images must live on the GPU (no CPU fallback)")
    assert images_bgr.dtype == torch.uint8 and images_bgr.dim() == 4 and images_bgr.shape[3] == 3
    assert images_bgr.is_contiguous()
    lib = _lib.load()
    B, h, w, _ = images_bgr.shape
    oh, ow = (h // PATCH) * PATCH, (w // PATCH) * PATCH
    code = {torch.float32: 0, torch.bfloat16: 1}[out_dtype]
    if layout == "patches":
        out = torch.empty((B, (oh // PATCH) * (ow // PATCH), 3 * PATCH * PATCH), dtype=out_dtype, device=images_bgr.device)
        lay = 1
    elif layout == "patches_pad":   # rows padded to 640 elements (zeros): the A operand of vc_patch_embed_bf16
        out = torch.empty((B, (oh // PATCH) * (ow // PATCH), 640), dtype=out_dtype, device=images_bgr.device)
        lay = 2
    elif layout == "nchw":
        out = torch.empty((B, 3, oh, ow), dtype=out_dtype, device=images_bgr.device)
        lay = 0
    else:
        raise ValueError(layout)
    dbg = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=images_bgr.device) if want_resized else None
    _lib.check(lib.vc_preprocess_u8(_lib.ptr(images_bgr), B, h, w, oh, ow, code, lay, _lib.ptr(out), _lib.ptr(dbg),
                                    _lib.stream_ptr()), "vc_preprocess_u8")
    return (out, dbg) if want_resized else out
