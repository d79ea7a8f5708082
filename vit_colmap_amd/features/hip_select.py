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
    yx = torch.empty((B, kmax, 2), dtype=torch.int32, device=dev)      # (the kernel zeroes the slots behind the kept points)
    sc = torch.empty((B, kmax), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
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


def describe_at(tokens: torch.Tensor, H: int, W: int, keypoints_xy: torch.Tensor, count: torch.Tensor, feature_wh, original_wh,
                projection=None, rootsift: bool = True, want_f32: bool = False):
    """Descriptors at given sub-pixel keypoints (reference hybrid_extractor.py:224-294): tokens (B, H*W, C),
    keypoints_xy float32 (B, kmax, 2) in original-image pixels, count int32 (B,) -> uint8 (B, kmax, D)[, float32]."""
    _check_tokens(tokens, H, W)
    assert keypoints_xy.is_cuda and keypoints_xy.dtype == torch.float32 and keypoints_xy.is_contiguous() and keypoints_xy.shape[-1] == 2
    lib = _lib.load()
    B, _, C = tokens.shape
    kmax = keypoints_xy.shape[1]
    dd = 0
    if projection is not None:
        assert projection.is_cuda and projection.dtype == torch.float32 and projection.is_contiguous() and projection.shape[0] == C
        dd = projection.shape[1]
    D = dd if projection is not None else C
    u8 = torch.empty((B, kmax, D), dtype=torch.uint8, device=tokens.device)
    f32 = torch.empty((B, kmax, D), dtype=torch.float32, device=tokens.device) if want_f32 else None
    _lib.check(lib.vc_describe_at(_lib.ptr(tokens), _dtype_code(tokens), B, H, W, C, _lib.ptr(keypoints_xy), _lib.ptr(count), kmax,
                                  _lib.ptr(projection), dd, int(feature_wh[0]), int(feature_wh[1]), int(original_wh[0]),
                                  int(original_wh[1]), 1 if rootsift else 0, _lib.ptr(f32), _lib.ptr(u8), _lib.stream_ptr()),
               "vc_describe_at")
    return (u8, f32) if want_f32 else u8


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


def heatmap_workspace_bytes(n_images: int, H: int, W: int, kmax: int) -> int:
    return _lib.load().vc_heatmap_workspace_bytes(n_images, H, W, kmax)


def heatmap_keypoints(kp_map: torch.Tensor, desc_map: torch.Tensor, num_keypoints: int, score_threshold: float,
                      nms_radius: int, original_wh, resized_wh):
    """Dense head outputs -> keypoints / descriptors (csrc/heatmap.hip; replaces the post-model part of the reference's
    TrainableViTExtractor._run_inference, trainable_vit_extractor.py:170-267), batched.

    kp_map (B, 4, H, W) float32 (logit, dx, dy, orientation); desc_map (B, D, H, W) float32 in any layout whose rows are
    dense (contiguous or channels-last).  Returns dict: keypoints (B, K, 6) float32, desc_u8 (B, K, D) uint8 (rows beyond
    count are zero), count (B,) int32."""
    if not (kp_map.is_cuda and desc_map.is_cuda):
        raise _lib.HipLibraryError("head outputs must live on the GPU (no CPU fallback)")
    assert kp_map.dtype == torch.float32 and desc_map.dtype == torch.float32 and kp_map.dim() == 4 and kp_map.shape[1] == 4
    kp_map = kp_map.contiguous()
    B, _, H, W = kp_map.shape
    D = desc_map.shape[1]
    assert tuple(desc_map.shape) == (B, D, H, W), (desc_map.shape, kp_map.shape)
    if desc_map.stride(2) != W * desc_map.stride(3):
        desc_map = desc_map.contiguous()
    lib = _lib.load()
    dev = kp_map.device
    K = int(num_keypoints)
    ws = torch.empty(max(lib.vc_heatmap_workspace_bytes(B, H, W, K), 16), dtype=torch.uint8, device=dev)
    kps = torch.empty((B, K, 6), dtype=torch.float32, device=dev)
    du8 = torch.empty((B, K, D), dtype=torch.uint8, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    w_o, h_o = original_wh
    w_n, h_n = resized_wh
    _lib.check(lib.vc_heatmap_keypoints(_lib.ptr(kp_map), _lib.ptr(desc_map), desc_map.stride(0), desc_map.stride(1),
                                        desc_map.stride(3), B, H, W, D, int(nms_radius), float(score_threshold), K,
                                        float(w_o / w_n), float(h_o / h_n), float(w_o - 1), float(h_o - 1), _lib.ptr(ws),
                                        _lib.ptr(kps), _lib.ptr(du8), _lib.ptr(cnt), _lib.stream_ptr()), "vc_heatmap_keypoints")
    return {"keypoints": kps, "desc_u8": du8, "count": cnt}
