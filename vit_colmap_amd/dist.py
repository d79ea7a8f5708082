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
Product entries built on it: `matching.match_exhaustive(..., distributed=True)` (database in, database out) and
`pipeline.distributed.run_sharded` (what `Pipeline.run` calls when a process group with more than one rank exists).
"""
import numpy as np
import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world():
    return (dist.get_rank(), dist.get_world_size()) if is_distributed() else (0, 1)


def comm_device(default="cuda"):
    """Device collectives' tensors must live on: the GPU under RCCL ("nccl"), the host under gloo."""
    if is_distributed() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return torch.device(default)


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


def all_gather_descriptors(desc: torch.Tensor, counts: torch.Tensor, force_collective: bool = False):
    """desc uint8 (n_local, n_max, D), counts int32 (n_local,) -> the same for all ranks' images,
    concatenated in rank order.  Every rank must pass the same n_local (pad with count 0).
    force_collective: run the collective even in a one-rank group (the single-GPU RCCL smoke test: the same
    all_gather_into_tensor calls on device tensors as a multi-GPU run, world size 1)."""
    if not is_distributed() and not (force_collective and dist.is_available() and dist.is_initialized()):
        return desc, counts
    world = dist.get_world_size()
    all_desc = torch.empty((world * desc.shape[0],) + tuple(desc.shape[1:]), dtype=desc.dtype, device=desc.device)
    all_counts = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(all_desc, desc.contiguous())
    dist.all_gather_into_tensor(all_counts, counts.contiguous())
    return all_desc, all_counts


def all_gather_rows(t: torch.Tensor) -> torch.Tensor:
    """Any (n_local, ...) tensor -> (world * n_local, ...) in rank order (same n_local on every rank)."""
    if not is_distributed():
        return t
    out = torch.empty((dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


def max_over_ranks(*values):
    """Element-wise maximum of small non-negative integers over all ranks (block shapes that must agree)."""
    if not is_distributed():
        return tuple(int(v) for v in values)
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return tuple(int(v) for v in t.tolist())


def broadcast_array(arr, src: int = 0, device=None):
    """numpy array on `src` (anything elsewhere) -> the same array on every rank."""
    if not is_distributed():
        return arr
    meta = [None]
    if dist.get_rank() == src:
        arr = np.ascontiguousarray(arr)
        meta = [(arr.shape, arr.dtype.str)]
    dist.broadcast_object_list(meta, src=src)
    shape, dtype = meta[0]
    dev = comm_device(device or "cuda")
    if dist.get_rank() == src:
        t = torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(dev)
    else:
        t = torch.empty(int(np.prod(shape)) * np.dtype(dtype).itemsize, dtype=torch.uint8, device=dev)
    if t.numel():
        dist.broadcast(t, src=src)
    return t.cpu().numpy().view(np.dtype(dtype)).reshape(shape)


def gather_pair_lists(pairs: np.ndarray, lists, dst: int = 0):
    """Variable-length gather of (pair, match list) results to `dst` (host arrays; small).
    Returns on dst a dict {(a, b): uint32 (M, 2)} over all ranks' pairs, elsewhere None."""
    if not is_distributed():
        return {(int(a), int(b)): m for (a, b), m in zip(pairs, lists)}
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object((np.asarray(pairs), [np.asarray(m) for m in lists]), out, dst=dst)
    if out is None:
        return None
    merged = {}
    for prs, ls in out:
        for (a, b), m in zip(prs, ls):
            merged[(int(a), int(b))] = m
    return merged


def gather_pair_results(pairs: np.ndarray, results, dst: int = 0):
    """Variable-size gather of one picklable result per pair (two-view geometries: inlier lists + matrices; small) to
    `dst`.  Returns on dst a dict {(a, b): result} over all ranks' pairs, elsewhere None."""
    if not is_distributed():
        return {(int(a), int(b)): r for (a, b), r in zip(pairs, results)}
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object((np.asarray(pairs), list(results)), out, dst=dst)
    if out is None:
        return None
    merged = {}
    for prs, rs in out:
        for (a, b), r in zip(prs, rs):
            merged[(int(a), int(b))] = r
    return merged


def raise_if_any_failed(local_error=None, what: str = ""):
    """Collective error hand-shake: every rank calls it at the same point with its own exception (or None).  If any rank
    failed, EVERY rank raises — the failing one its own exception, the others a RuntimeError naming the rank — instead of
    the healthy ranks waiting in the next collective until the process-group timeout (e.g. rank 0, the only SQLite
    writer, hitting an IntegrityError while the others sit in a broadcast)."""
    if not is_distributed():
        if local_error is not None:
            raise local_error
        return
    flag = torch.tensor([0 if local_error is None else 1 + dist.get_rank()], dtype=torch.int64, device=comm_device())
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if local_error is not None:
        raise local_error
    if int(flag.item()):
        raise RuntimeError(f"rank {int(flag.item()) - 1} failed{(' in ' + what) if what else ''}; this rank stops with it")


def broadcast_object(obj, src: int = 0):
    """Small picklable object on `src` -> every rank."""
    if not is_distributed():
        return obj
    box = [obj if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def gather_match_lists(pairs: np.ndarray, counts: np.ndarray, matches: np.ndarray, dst: int = 0):
    """Padded-array form of gather_pair_lists: returns on dst a list of (pairs, counts, matches) per rank."""
    if not is_distributed():
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
