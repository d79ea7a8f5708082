"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).

The reference is single-process; this module is the build's own design (§8e):
  * images: contiguous blocks per rank in sorted-filename order, so image ids stay deterministic;
  * descriptors: ONE all-gather of the per-rank uint8 blocks (padded to n_max rows) and counts —
    the only collective on the data path; on xGMI's point-to-point links an all-gather of
    n_local*n_max*D bytes per rank (9.8 MB at 50 x 512 x 384) is far below a millisecond;
  * pairs: the exhaustive pair list is dealt round-robin (pair p -> rank p % world), so every
    rank matches the same number of pairs (+-1) with no further communication;
  * results: match lists are gathered to rank 0, the single SQLite writer.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int):
    """Contiguous block [lo, hi) of n items for `rank`: ceil(n / world) per rank, last ones shorter."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def pairs_for_rank(n_images: int, rank: int, world: int) -> np.ndarray:
    """Rows p of the exhaustive pair list (a < b, row-major) with p % world == rank; int32 (P_r, 2)."""
    a, b = np.triu_indices(n_images, k=1)
    sel = np.arange(rank, len(a), world)
    return np.stack([a[sel], b[sel]], axis=1).astype(np.int32)


def pair_index(n_images: int, a, b):
    """Position of pair (a < b) in the row-major exhaustive list."""
    a = np.asarray(a, np.int64)
    b = np.asarray(b, np.int64)
    return a * (2 * n_images - a - 1) // 2 + (b - a - 1)


def all_gather_descriptors(desc: torch.Tensor, counts: torch.Tensor):
    """desc uint8 (n_local, n_max, D), counts int32 (n_local,) -> the same for all ranks' images,
    concatenated in rank order.  Every rank must pass the same n_local (pad with count 0)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return desc, counts
    world = dist.get_world_size()
    all_desc = torch.empty((world * desc.shape[0],) + tuple(desc.shape[1:]), dtype=desc.dtype, device=desc.device)
    all_counts = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(all_desc, desc.contiguous())
    dist.all_gather_into_tensor(all_counts, counts.contiguous())
    return all_desc, all_counts


def gather_match_lists(pairs: np.ndarray, counts: np.ndarray, matches: np.ndarray, dst: int = 0):
    """Variable-length gather of (pair, match list) results to `dst` (host arrays; small).
    Returns on dst a list of (pairs, counts, matches) per rank, elsewhere None."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [(pairs, counts, matches)]
    payload = (pairs, counts, [matches[p, : counts[p]].copy() for p in range(len(pairs))])
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(payload, out, dst=dst)
    if out is None:
        return None
    res = []
    for prs, cnt, lists in out:
        m = np.zeros((len(prs), max([len(x) for x in lists] + [1]), 2), np.uint32)
        for p, x in enumerate(lists):
            m[p, : len(x)] = x
        res.append((prs, cnt, m))
    return res
