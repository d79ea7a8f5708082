"""Exhaustive matching over a COLMAP database — the DB-in / DB-out contract of
`pycolmap.match_exhaustive(database_path=..., matching_options=... | sift_options=...)` that the
reference calls at vit_colmap/pipeline/run_pipeline.py:351-363, on the HIP matcher.

Reads every image's uint8 descriptors, matches all unordered pairs (a < b in image-id order) on
the GPU, and writes one `matches` row per pair (also when it is empty, as COLMAP does [recalled]).
With `verify=True` (default) the match lists are then geometrically verified (matching/two_view.py) and
`two_view_geometries` rows written, as `match_exhaustive` does inside COLMAP.

Multi-GPU (`distributed=True`, or automatically when a torch.distributed group with more than one rank exists):
rank 0 reads the database and broadcasts the descriptor blocks, the pair list is dealt round-robin over the
ranks, every rank matches its share on its own GPU, the lists are gathered to rank 0 and rank 0 alone writes
(vit_colmap_amd/dist.py).
"""
import logging
import time

import numpy as np
import torch

from .. import _lib
from .. import dist as vd
from ..database.colmap_db import SqliteColmapDatabase
from .hip_matcher import exhaustive_pairs, match_pairs, prepare_descriptors

logger = logging.getLogger(__name__)


def _sift_options(matching_options, sift_options):
    opts = matching_options if matching_options is not None else sift_options
    if opts is None:
        from ..utils.config import MatchingConfig

        opts = MatchingConfig().to_matching_options()
    return getattr(opts, "sift", opts)  # FeatureMatchingOptions(.sift) or SiftMatchingOptions


def load_descriptor_blocks(db: SqliteColmapDatabase):
    """-> image ids (ascending), uint8 [n_images][n_max][D] (zero padded), counts int32."""
    images = db.read_all_images()
    ids = [im.image_id for im in images]
    descs = [db.read_descriptors(i) for i in ids]
    dims = {d.shape[1] for d in descs if d is not None and d.shape[0] > 0}
    if len(dims) > 1:
        raise ValueError(f"images have descriptors of different dimensions: {sorted(dims)}")
    D = dims.pop() if dims else 0
    counts = np.array([0 if d is None else d.shape[0] for d in descs], np.int32)
    n_max = int(counts.max()) if len(counts) else 0
    block = np.zeros((len(ids), max(n_max, 1), max(D, 1)), np.uint8)
    for k, d in enumerate(descs):
        if d is not None and d.shape[0] > 0:
            block[k, : d.shape[0]] = d
    return ids, block, counts, D


def hip_match_blocks(block, counts, pairs, max_ratio=0.8, max_distance=0.7, cross_check=True, device="cuda",
                     pair_chunk: int = 16384):
    """uint8 blocks [n][n_max][D] + counts (host or device) and pairs int32 (P, 2) (host) -> list of P uint32 (M, 2)
    match lists, on the HIP matcher.  Blocks with more rows than one kernel block holds (VC_MAX_KEYPOINTS) are
    matched in row / column sub-blocks whose top-2 results are merged (hip_matcher.match_pairs_blocked)."""
    if not torch.cuda.is_available():
        raise _lib.HipLibraryError("the matcher is HIP-only (no CPU fallback): no GPU visible")
    d_desc = block if torch.is_tensor(block) else torch.from_numpy(np.ascontiguousarray(block))
    d_counts = counts if torch.is_tensor(counts) else torch.from_numpy(np.ascontiguousarray(counts, np.int32))
    d_desc, d_counts = d_desc.to(device), d_counts.to(device)
    n, n_max, D = d_desc.shape
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    if D > _lib.VC_MAX_DESC_DIM:
        raise _lib.HipLibraryError(f"descriptors of {D} bytes exceed the kernels' limit ({_lib.VC_MAX_DESC_DIM})")
    if n_max > _lib.VC_MAX_KEYPOINTS:
        from .hip_matcher import match_pairs_blocked

        return match_pairs_blocked(d_desc, d_counts, pairs, max_ratio, max_distance, cross_check)
    prepared = prepare_descriptors(d_desc, d_counts)
    out = []
    for s in range(0, len(pairs), pair_chunk):
        chunk = torch.from_numpy(pairs[s:s + pair_chunk]).to(device)
        m, c = match_pairs(prepared, d_counts, n, n_max, D, chunk, max_ratio, max_distance, cross_check)
        c_np = c.cpu().numpy()
        if (c_np < 0).any():   # VC_COUNT_SELFCHECK_FAILED: the kernel's cursor check (include/vitcolmap_hip.h) — never a result
            bad = np.nonzero(c_np < 0)[0][:8] + s
            raise _lib.HipLibraryError(f"vc_match_pairs_u8: consistency check failed for pairs {bad.tolist()}")
        m_np = m.cpu().numpy().view(np.uint32)
        out.extend(m_np[p, : c_np[p]].copy() for p in range(len(c_np)))
    return out


def _pack_keypoints(kps: dict, n: int):
    """{index: (N, >= 2)} -> float32 (n, max N, 2) zero padded + int32 counts: the form that travels between ranks."""
    cnt = np.array([len(kps[k]) for k in range(n)], np.int32)
    out = np.zeros((n, max(int(cnt.max()) if n else 0, 1), 2), np.float32)
    for k in range(n):
        out[k, : cnt[k]] = kps[k][:, :2]
    return out, cnt


def _unpack_keypoints(arr, cnt):
    return {k: arr[k, : cnt[k]] for k in range(len(cnt))}


def match_exhaustive(database_path: str, matching_options=None, sift_options=None, device="cuda",
                     pair_chunk: int = 16384, distributed=None, match_fn=None, verify: bool = True, verify_fn=None) -> dict:
    """Returns a small stats dict (pairs, matches, seconds); the result proper is in the database.
    Multi-rank (`distributed`): rank 0 — the only process that touches the SQLite file — reads descriptors and keypoints
    and broadcasts them; EVERY rank matches and geometrically verifies its share of the pair list (pair p -> rank
    p % world); match lists and two-view geometries are gathered to rank 0, which writes them in pair order.  An error
    on any rank (e.g. rank 0's database) is raised on all of them (dist.raise_if_any_failed).
    `match_fn(block, counts, pairs, max_ratio, max_distance, cross_check) -> list of match lists` and
    `verify_fn(keypoints, pair_images, pair_ids, lists) -> list of results` replace the HIP matcher / scorer (the CPU
    tests of the multi-rank path pass the oracles; the product never does)."""
    sift = _sift_options(matching_options, sift_options)
    max_ratio, max_distance, cross_check = float(sift.max_ratio), float(sift.max_distance), bool(sift.cross_check)
    if distributed is None:
        distributed = vd.is_distributed()
    if distributed and not vd.is_distributed():
        raise RuntimeError("distributed=True needs an initialised torch.distributed process group with > 1 rank")
    rank, world = vd.rank_world() if distributed else (0, 1)
    if match_fn is None:
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("match_exhaustive needs an MI355X: the matcher is HIP-only (no CPU fallback)")

        def match_fn(block, counts, pairs, r, dmax, cc):
            return hip_match_blocks(block, counts, pairs, r, dmax, cc, device=device, pair_chunk=pair_chunk)

    from .two_view import read_keypoints_by_index, verify_pair_lists, write_two_view_rows

    t0 = time.perf_counter()
    db = None
    try:
        ids = block = counts = D = kp_arr = kp_cnt = None
        err = None
        if rank == 0:                                                          # rank 0 is the only reader and writer
            try:
                db = SqliteColmapDatabase(str(database_path))
                ids, block, counts, D = load_descriptor_blocks(db)
                if verify and D != 0:
                    kp_arr, kp_cnt = _pack_keypoints(read_keypoints_by_index(db, ids), len(ids))
            except Exception as e:  # noqa: BLE001 - handed to every rank below
                err = e
        if distributed:
            vd.raise_if_any_failed(err, "reading the database")
            ids, D = vd.broadcast_object((ids, D), 0)
            block = vd.broadcast_array(block, 0, device)
            counts = vd.broadcast_array(counts, 0, device)
            if verify and D != 0:
                kp_arr = vd.broadcast_array(kp_arr, 0, device)
                kp_cnt = vd.broadcast_array(kp_cnt, 0, device)
        elif err is not None:
            raise err
        n = len(ids)
        stats = dict(images=n, pairs=n * (n - 1) // 2, matches=0, gpu_s=0.0, db_s=0.0, verified_pairs=0, ranks=world)
        if n < 2:
            return stats
        my_pairs = vd.pairs_for_rank(n, rank, world)
        t1 = time.perf_counter()
        err, lists, results = None, [], None
        try:
            if D == 0:
                lists = [np.zeros((0, 2), np.uint32) for _ in my_pairs]      # no descriptors anywhere: every pair is empty
            else:
                lists = match_fn(block, counts, my_pairs, max_ratio, max_distance, cross_check)
                if verify:                                                     # this rank verifies the pairs it matched
                    vdev = device if (verify_fn is not None or torch.cuda.is_available()) else "cpu"
                    results = verify_pair_lists(_unpack_keypoints(kp_arr, kp_cnt), ids, my_pairs, lists, device=vdev,
                                                verify_fn=verify_fn)
        except Exception as e:  # noqa: BLE001
            err = e
        if distributed:
            vd.raise_if_any_failed(err, "matching / verification")
        elif err is not None:
            raise err
        stats["gpu_s"] = time.perf_counter() - t1
        merged = vd.gather_pair_lists(my_pairs, lists, dst=0) if distributed else \
            {(int(a), int(b)): m for (a, b), m in zip(my_pairs, lists)}
        verified = None
        if results is not None:
            verified = vd.gather_pair_results(my_pairs, results, dst=0) if distributed else \
                {(int(a), int(b)): r for (a, b), r in zip(my_pairs, results)}
        err = None
        if rank == 0:
            try:
                t2 = time.perf_counter()
                for a, b in exhaustive_pairs(n).numpy():                       # COLMAP's pair order, whatever rank matched it
                    m = merged[(int(a), int(b))]
                    db.write_matches(ids[a], ids[b], m, commit=False)
                    stats["matches"] += len(m)
                db.commit()
                if verified is not None:
                    stats["verified_pairs"] = write_two_view_rows(db, ids, verified)
                stats["db_s"] = time.perf_counter() - t2
            except Exception as e:  # noqa: BLE001
                err = e
        if distributed:
            vd.raise_if_any_failed(err, "writing the database")
            stats = vd.broadcast_object(stats, 0)                              # every rank returns rank 0's totals
        elif err is not None:
            raise err
        stats["total_s"] = time.perf_counter() - t0
        logger.info("matched %d pairs (%d matches) on %d rank(s): gpu %.3f s, db %.3f s", stats["pairs"], stats["matches"],
                    world, stats["gpu_s"], stats["db_s"])
        return stats
    finally:
        if db is not None:
            db.close()
