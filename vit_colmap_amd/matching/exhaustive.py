"""Exhaustive matching over a COLMAP database — the DB-in / DB-out contract of
`pycolmap.match_exhaustive(database_path=..., matching_options=... | sift_options=...)` that the
reference calls at vit_colmap/pipeline/run_pipeline.py:351-363, on the HIP matcher.

Reads every image's uint8 descriptors, matches all unordered pairs (a < b in image-id order) on
the GPU, and writes one `matches` row per pair (also when it is empty, as COLMAP does [recalled]).
`two_view_geometries` stays empty: geometric verification is outside the hot path.
"""
import logging
import time

import numpy as np
import torch

from .. import _lib
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


def match_exhaustive(database_path: str, matching_options=None, sift_options=None, device="cuda",
                     pair_chunk: int = 16384) -> dict:
    """Returns a small stats dict (pairs, matches, seconds); the result proper is in the database."""
    sift = _sift_options(matching_options, sift_options)
    max_ratio, max_distance, cross_check = float(sift.max_ratio), float(sift.max_distance), bool(sift.cross_check)
    if not torch.cuda.is_available():
        raise _lib.HipLibraryError("match_exhaustive needs an MI355X: the matcher is HIP-only (no CPU fallback)")
    t0 = time.perf_counter()
    db = SqliteColmapDatabase(str(database_path))
    try:
        ids, block, counts, D = load_descriptor_blocks(db)
        n = len(ids)
        stats = dict(images=n, pairs=n * (n - 1) // 2, matches=0, gpu_s=0.0, db_s=0.0)
        if n < 2:
            return stats
        pairs = exhaustive_pairs(n)
        if D == 0 or block.shape[1] > _lib.VC_MAX_KEYPOINTS or D > _lib.VC_MAX_DESC_DIM:
            if D != 0:
                raise _lib.HipLibraryError(
                    f"descriptor blocks of {block.shape[1]} x {D} exceed the kernels' limits "
                    f"({_lib.VC_MAX_KEYPOINTS} keypoints, {_lib.VC_MAX_DESC_DIM} bytes)")
            for a, b in pairs.numpy():   # no descriptors anywhere: every pair is empty
                db.write_matches(ids[a], ids[b], np.zeros((0, 2), np.uint32), commit=False)
            db.commit()
            return stats
        n_max = block.shape[1]
        t1 = time.perf_counter()
        d_desc = torch.from_numpy(block).to(device)
        d_counts = torch.from_numpy(counts).to(device)
        prepared = prepare_descriptors(d_desc, d_counts)
        for s in range(0, len(pairs), pair_chunk):
            chunk = pairs[s:s + pair_chunk].contiguous()
            m, c = match_pairs(prepared, d_counts, n, n_max, D, chunk.to(device), max_ratio, max_distance, cross_check)
            c_np = c.cpu().numpy()
            m_np = m.cpu().numpy().view(np.uint32)
            torch.cuda.synchronize()
            stats["gpu_s"] += time.perf_counter() - t1
            t2 = time.perf_counter()
            for p, (a, b) in enumerate(chunk.numpy()):
                db.write_matches(ids[a], ids[b], m_np[p, : c_np[p]], commit=False)
            db.commit()
            stats["matches"] += int(c_np.sum())
            stats["db_s"] += time.perf_counter() - t2
            t1 = time.perf_counter()
        stats["total_s"] = time.perf_counter() - t0
        logger.info("matched %d pairs (%d matches): gpu %.3f s, db %.3f s", stats["pairs"], stats["matches"],
                    stats["gpu_s"], stats["db_s"])
        return stats
    finally:
        db.close()
