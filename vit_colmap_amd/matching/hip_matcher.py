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


def match_pairs_blocked(desc: torch.Tensor, counts: torch.Tensor, pairs, max_ratio=0.8, max_distance=0.7,
                        cross_check=True, block_rows: int = _lib.VC_MAX_KEYPOINTS):
    """Image pairs whose blocks hold more rows than one kernel block (VC_MAX_KEYPOINTS) — e.g. the reference's
    `trainable_vit` pipeline asks for 20 480 keypoints (run_pipeline.py:328-333).

    Each image is cut into sub-blocks of `block_rows` rows; every (sub-block of a, sub-block of b) runs through the
    one-way top-2 search in both directions (vc_knn_top2_u8) and the per-row results are merged over the
    sub-blocks of the other image in ascending order — best and second-best are associative: a later sub-block
    takes over only with a strictly larger similarity (so the lowest column index still wins ties) and the
    displaced or the non-winning best becomes a runner-up candidate — then vc_mutual_ratio applies the angle /
    ratio tests and the cross check to the merged arrays.  Bit-identical to the single-block path by construction
    (tests/test_matcher_gpu.py compares both with the C oracle).

    desc uint8 (n_images, n_max, D) on the GPU, counts int32 (n_images,), pairs int (P, 2) on the host.
    Returns a list of P uint32 (M, 2) arrays."""
    _need_cuda(desc, counts)
    cnt = counts.cpu().numpy()
    out = []
    for a, b in np.asarray(pairs).reshape(-1, 2):
        n1, n2 = int(cnt[a]), int(cnt[b])
        if n1 == 0 or n2 == 0:
            out.append(np.zeros((0, 2), np.uint32))
            continue
        da, db = desc[a, :n1], desc[b, :n2]
        dev = desc.device
        r12 = [torch.full((n1,), -1, dtype=torch.int32, device=dev), torch.zeros(n1, dtype=torch.int32, device=dev),
               torch.zeros(n1, dtype=torch.int32, device=dev)]
        r21 = [torch.full((n2,), -1, dtype=torch.int32, device=dev), torch.zeros(n2, dtype=torch.int32, device=dev),
               torch.zeros(n2, dtype=torch.int32, device=dev)]

        def merge(acc, lo, hi, idx, best, second, offset):
            # rows lo..hi of the accumulated (idx, best, second) absorb one more sub-block of the other image
            ai, ab, asec = acc[0][lo:hi], acc[1][lo:hi], acc[2][lo:hi]
            gt = best > ab
            new_sec = torch.where(gt, torch.maximum(ab, second), torch.maximum(asec, best))
            acc[0][lo:hi] = torch.where(gt, idx + offset, ai)
            acc[2][lo:hi] = new_sec
            acc[1][lo:hi] = torch.maximum(ab, best)

        for i0 in range(0, n1, block_rows):
            i1 = min(i0 + block_rows, n1)
            for j0 in range(0, n2, block_rows):      # ascending: the earlier sub-block keeps ties
                j1 = min(j0 + block_rows, n2)
                blk_a, blk_b = da[i0:i1].contiguous(), db[j0:j1].contiguous()
                merge(r12, i0, i1, *knn_top2(blk_a, blk_b), j0)
        for j0 in range(0, n2, block_rows):
            j1 = min(j0 + block_rows, n2)
            for i0 in range(0, n1, block_rows):
                i1 = min(i0 + block_rows, n1)
                merge(r21, j0, j1, *knn_top2(db[j0:j1].contiguous(), da[i0:i1].contiguous()), i0)
        m, c = mutual_ratio(r12, r21, n1, n2, max_ratio, max_distance, cross_check)
        k = int(c.item())
        out.append(m[:k].cpu().numpy().view(np.uint32).copy())
    return out
