"""Torch-tensor front end of the HIP matcher kernels (vit_colmap_amd/csrc/matcher.hip).

Tensors are plumbing for device memory and streams; all arithmetic runs in the C-ABI library.
Replaces the per-pair work of `pycolmap.match_exhaustive`
(reference vit_colmap/pipeline/run_pipeline.py:351-363).
"""
import numpy as np
import torch

from .. import _lib


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.HipLibraryError("matcher tensors must live on the GPU (no CPU fallback)")


def exhaustive_pairs(n_images: int, device=None) -> torch.Tensor:
    """All unordered pairs (a < b), row-major — COLMAP's exhaustive pairing order."""
    a, b = np.triu_indices(n_images, k=1)
    t = torch.from_numpy(np.stack([a, b], axis=1).astype(np.int32))
    return t.to(device) if device is not None else t


def prepare_descriptors(desc: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """desc uint8 [n_images, n_max, D] + counts int32 [n_images] -> MFMA-ready buffer (uint8 1-D)."""
    _need_cuda(desc, counts)
    assert desc.dtype == torch.uint8 and desc.dim() == 3 and desc.is_contiguous()
    assert counts.dtype == torch.int32 and counts.numel() == desc.shape[0]
    lib = _lib.load()
    n_images, n_max, d = desc.shape
    nbytes = lib.vc_prepared_bytes(n_images, n_max, d)
    if nbytes == 0:
        raise _lib.HipLibraryError(f"unsupported descriptor block shape {tuple(desc.shape)}")
    prepared = torch.empty(nbytes, dtype=torch.uint8, device=desc.device)
    _lib.check(lib.vc_prepare_descriptors(_lib.ptr(desc), _lib.ptr(counts), n_images, n_max, d,
                                          _lib.ptr(prepared), _lib.stream_ptr()), "vc_prepare_descriptors")
    return prepared


def match_pairs(prepared, counts, n_images, n_max, d, pairs, max_ratio=0.8, max_distance=0.7,
                cross_check=True, out_matches=None, out_counts=None):
    """-> (matches int32-viewed-uint32 [P, n_max, 2], match counts int32 [P])."""
    _need_cuda(prepared, counts, pairs)
    assert pairs.dtype == torch.int32 and pairs.is_contiguous() and pairs.shape[-1] == 2
    lib = _lib.load()
    P = pairs.shape[0]
    if out_matches is None:
        out_matches = torch.empty((P, n_max, 2), dtype=torch.int32, device=prepared.device)
    if out_counts is None:
        out_counts = torch.empty((P,), dtype=torch.int32, device=prepared.device)
    _lib.check(lib.vc_match_pairs_u8(_lib.ptr(prepared), _lib.ptr(counts), n_images, n_max, d,
                                     _lib.ptr(pairs), P, max_ratio, max_distance, int(cross_check),
                                     _lib.ptr(out_matches), _lib.ptr(out_counts), _lib.stream_ptr()),
               "vc_match_pairs_u8")
    return out_matches, out_counts


def knn_top2(d1: torch.Tensor, d2: torch.Tensor):
    """One-way search: per row of d1 -> (idx, best, second) against the rows of d2."""
    _need_cuda(d1, d2)
    assert d1.dtype == torch.uint8 and d2.dtype == torch.uint8 and d1.is_contiguous() and d2.is_contiguous()
    lib = _lib.load()
    n1, n2 = d1.shape[0], d2.shape[0]
    d = d1.shape[1]
    dev = d1.device
    idx = torch.full((n1,), -1, dtype=torch.int32, device=dev)
    best = torch.zeros((n1,), dtype=torch.int32, device=dev)
    second = torch.zeros((n1,), dtype=torch.int32, device=dev)
    if n1 == 0:
        return idx, best, second
    nbytes = lib.vc_knn_workspace_bytes(n1, n2, d)
    if nbytes == 0:
        raise _lib.HipLibraryError(f"unsupported shape n1={n1} n2={n2} d={d}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.vc_knn_top2_u8(_lib.ptr(d1), n1, _lib.ptr(d2), n2, d, _lib.ptr(idx), _lib.ptr(best),
                                  _lib.ptr(second), _lib.ptr(ws), nbytes, _lib.stream_ptr()), "vc_knn_top2_u8")
    return idx, best, second


def mutual_ratio(r12, r21, n1, n2, max_ratio=0.8, max_distance=0.7, cross_check=True):
    """(idx, best, second) triples for both directions -> (pairs int32 [n1, 2], count int32 [1])."""
    lib = _lib.load()
    dev = r12[0].device
    out = torch.empty((max(n1, 1), 2), dtype=torch.int32, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    a = [_lib.ptr(t) for t in r12]
    b = [_lib.ptr(t) for t in r21] if r21 is not None else [None, None, None]
    _lib.check(lib.vc_mutual_ratio(a[0], a[1], a[2], n1, b[0], b[1], b[2], n2, max_ratio, max_distance,
                                   int(cross_check), _lib.ptr(out), _lib.ptr(cnt), _lib.stream_ptr()),
               "vc_mutual_ratio")
    return out, cnt


def theta_table(n: int, device="cuda", evaluate: bool = False) -> torch.Tensor:
    """theta(s) for s in [0, n): the table the pair kernel reads, or (evaluate=True) the device's own evaluation."""
    lib = _lib.load()
    out = torch.empty((n,), dtype=torch.float32, device=device)
    fn, name = (lib.vc_theta_eval, "vc_theta_eval") if evaluate else (lib.vc_theta_table, "vc_theta_table")
    _lib.check(fn(_lib.ptr(out), n, _lib.stream_ptr()), name)
    return out
