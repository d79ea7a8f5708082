"""Torch-tensor front end of the selection / descriptor kernels (vit_colmap_amd/csrc/select.hip).

Replaces `ViTExtractor._dense_to_sparse` and helpers (reference
vit_colmap/features/vit_extractor.py:168-653), batched over images.  Tensors are plumbing for
device memory; the arithmetic runs in the C-ABI library and there is no CPU fallback.
"""
import torch

from .. import _lib

METHODS = {"harris": 0, "dog": 1, "combined": 2}


def _dtype_code(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise _lib.HipLibraryError(f"tokens must be float32 or bfloat16, got {t.dtype}")


def _check_tokens(tokens, H, W):
    if not tokens.is_cuda:
        raise _lib.HipLibraryError("tokens must live on the GPU (no CPU fallback)")
    assert tokens.dim() == 3 and tokens.is_contiguous() and tokens.shape[1] == H * W, tokens.shape


def structure_tensor(tokens: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """tokens (B, H*W, C) -> (B, 4, H*W) float32: mean gx^2, gy^2, gx*gy and the channel mean."""
    _check_tokens(tokens, H, W)
    lib = _lib.load()
    B, _, C = tokens.shape
    st = torch.empty((B, 4, H * W), dtype=torch.float32, device=tokens.device)
    _lib.check(lib.vc_structure_tensor(_lib.ptr(tokens), _dtype_code(tokens), B, H, W, C, _lib.ptr(st),
                                       _lib.stream_ptr()), "vc_structure_tensor")
    return st


def score_map(st: torch.Tensor, H: int, W: int, method: str = "harris") -> torch.Tensor:
    """(B, 4, H*W) -> (B, H, W) float32 score in [0, 1]."""
    if method not in METHODS:
        raise ValueError(f"Unknown detection method: {method}")  # reference vit_extractor.py:279
    lib = _lib.load()
    B = st.shape[0]
    score = torch.empty((B, H, W), dtype=torch.float32, device=st.device)
    _lib.check(lib.vc_score_map(_lib.ptr(st), B, H, W, METHODS[method], _lib.ptr(score), _lib.stream_ptr()),
               "vc_score_map")
    return score


def select_keypoints(score: torch.Tensor, target: int, bin_size: int = 16, nms_radius: float = 1.5,
                     kmax=None, debug_candidates: bool = False):
    """(B, H, W) score -> (yx int32 (B, kmax, 2), score (B, kmax), count int32 (B,)[, candidates])."""
    assert score.is_cuda and score.dtype == torch.float32 and score.is_contiguous() and score.dim() == 3
    lib = _lib.load()
    B, H, W = score.shape
    if kmax is None:
        kmax = target
    dev = score.device
    yx = torch.zeros((B, kmax, 2), dtype=torch.int32, device=dev)
    sc = torch.zeros((B, kmax), dtype=torch.float32, device=dev)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    dbg = (None, None, None)
    if debug_candidates:
        dbg = (torch.zeros_like(yx), torch.zeros_like(sc), torch.zeros_like(cnt))
    _lib.check(lib.vc_select_keypoints(_lib.ptr(score), B, H, W, target, bin_size, nms_radius, kmax, _lib.ptr(yx),
                                       _lib.ptr(sc), _lib.ptr(cnt), _lib.ptr(dbg[0]), _lib.ptr(dbg[1]),
                                       _lib.ptr(dbg[2]), _lib.stream_ptr()), "vc_select_keypoints")
    return (yx, sc, cnt, dbg) if debug_candidates else (yx, sc, cnt)


def describe(tokens: torch.Tensor, H: int, W: int, yx: torch.Tensor, count: torch.Tensor, resized_wh,
             original_wh, projection=None, want_f32: bool = False):
    """-> keypoints float32 (B, kmax, 2) [x, y], descriptors uint8 (B, kmax, D)[, float32 descriptors]."""
    _check_tokens(tokens, H, W)
    lib = _lib.load()
    B, _, C = tokens.shape
    kmax = yx.shape[1]
    dd = 0
    if projection is not None:
        assert projection.is_cuda and projection.dtype == torch.float32 and projection.is_contiguous()
        assert projection.shape[0] == C
        dd = projection.shape[1]
    D = dd if projection is not None else C
    dev = tokens.device
    kp = torch.empty((B, kmax, 2), dtype=torch.float32, device=dev)
    u8 = torch.empty((B, kmax, D), dtype=torch.uint8, device=dev)
    f32 = torch.empty((B, kmax, D), dtype=torch.float32, device=dev) if want_f32 else None
    _lib.check(lib.vc_describe(_lib.ptr(tokens), _dtype_code(tokens), B, H, W, C, _lib.ptr(yx), _lib.ptr(count), kmax,
                               _lib.ptr(projection), dd, int(resized_wh[0]), int(resized_wh[1]),
                               int(original_wh[0]), int(original_wh[1]), _lib.ptr(kp), _lib.ptr(f32), _lib.ptr(u8),
                               _lib.stream_ptr()), "vc_describe")
    return (kp, u8, f32) if want_f32 else (kp, u8)


def quantize_u8(x: torch.Tensor) -> torch.Tensor:
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    lib = _lib.load()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _lib.check(lib.vc_quantize_u8(_lib.ptr(x), _lib.ptr(out), x.numel(), _lib.stream_ptr()), "vc_quantize_u8")
    return out


def dense_to_sparse(tokens, H, W, original_wh, resized_wh, num_keypoints, method="harris", projection=None,
                    bin_size=16, nms_radius=1.5, want_f32=False):
    """The whole post-ViT path for a batch (reference `_dense_to_sparse`, vit_extractor.py:168-252).
    Returns dict(keypoints, desc_u8, count[, desc_f32], yx, score)."""
    st = structure_tensor(tokens, H, W)
    score = score_map(st, H, W, method)
    yx, sc, cnt = select_keypoints(score, num_keypoints, bin_size, nms_radius)
    out = describe(tokens, H, W, yx, cnt, resized_wh, original_wh, projection, want_f32)
    res = dict(keypoints=out[0], desc_u8=out[1], count=cnt, yx=yx, scores=sc, score=score)
    if want_f32:
        res["desc_f32"] = out[2]
    return res
