"""Extraction + matching sharded over the GPUs of one node — what `Pipeline.run` does when a
torch.distributed process group with more than one rank exists (one process per GPU, backend "nccl" = RCCL).

  1. every rank lists the images (sorted, as the reference does) and takes a contiguous block of them;
  2. rank 0 — the only process that ever touches the SQLite file — writes the camera and ONE image row per readable
     image before any inference (reference vit_extractor.py:739: a failed image still has its row);
  3. every rank extracts its block on its own GPU; the per-image uint8 descriptor blocks, padded to a common
     (n_max, D), are ALL-GATHERED (the one collective of the data path, RCCL over xGMI), keypoints with them;
  4. rank 0 writes keypoints / descriptors; the exhaustive pair list is dealt round-robin, every rank matches
     and geometrically verifies its share from the gathered blocks and keypoints (no database read); the match lists
     and two-view geometries are gathered to rank 0, which writes them in pair order.
An error on any rank (rank 0's database included) is raised on every rank instead of leaving the others in a collective
(dist.raise_if_any_failed).
The reference is single-process; there is no counterpart to cite beyond the plugin API it keeps
(`extract(image_dir, db_path, camera_model, camera_params)`, run_pipeline.py:343, and the match call :351-363).

`feature_fn(list of BGR arrays) -> list of (keypoints (N, k) float32, descriptors (N, D) uint8)` and
`match_fn(blocks, counts, pairs, max_ratio, max_distance, cross_check) -> list of match lists` and
`verify_fn(keypoints, pair_images, pair_ids, lists) -> list of results` default to the HIP extractor / matcher / scorer;
the world-size-2 gloo test passes host stand-ins (there is no GPU in that container).
"""
import logging
from pathlib import Path

import numpy as np
import torch

from .. import dist as vd
from ..features.base_extractor import default_camera_params, list_images
from ..utils import image_io

logger = logging.getLogger(__name__)


def run_sharded(image_dir, db_path, camera_model, camera_params=None, feature_fn=None, matching_options=None,
                match_fn=None, do_matching=True, verify=True, device="cuda", batch_size=50, verify_fn=None) -> dict:
    from ..database.colmap_db import Camera, ColmapDatabase
    from ..matching.exhaustive import _sift_options, hip_match_blocks

    rank, world = vd.rank_world()
    image_files = list_images(Path(image_dir))
    if not image_files:
        raise ValueError(f"No images found in {image_dir}")
    n = len(image_files)
    lo, hi = vd.shard_range(n, rank, world)
    per = (n + world - 1) // world
    cdev = vd.comm_device(device)

    # ---- this rank's block ------------------------------------------------------------------------------------------
    readable = np.zeros(per, np.int32)
    feats = [None] * per
    first_shape = np.zeros(2, np.int64)
    pending = []

    def flush():
        if pending:
            for (k, _), r in zip(pending, feature_fn([img for _, img in pending])):
                feats[k] = r
            pending.clear()

    for k, f in enumerate(image_files[lo:hi]):
        img = image_io.imread(f)
        if img is None:
            print(f"{f.name}: ⚠ failed to read image, skipping")
            continue
        readable[k] = 1
        if lo + k == 0:
            first_shape[:] = img.shape[:2]
        if pending and (pending[0][1].shape != img.shape or len(pending) >= batch_size):
            flush()
        pending.append((k, img))
    flush()
    kdim = max([f[0].shape[1] for f in feats if f is not None] + [2])
    n_max, D, kdim = vd.max_over_ranks(max([len(f[0]) for f in feats if f is not None] + [1]),
                                       max([f[1].shape[1] for f in feats if f is not None] + [1]), kdim)
    desc = np.zeros((per, n_max, D), np.uint8)
    kps = np.zeros((per, n_max, kdim), np.float32)
    counts = np.zeros(per, np.int32)
    for k, f in enumerate(feats):
        if f is not None and len(f[0]):
            counts[k] = len(f[0])
            kps[k, : counts[k]] = f[0]
            desc[k, : counts[k]] = f[1]

    # ---- the collective: descriptor blocks (+ counts, keypoints, readability) of every rank ------------------------------
    all_desc, all_counts = vd.all_gather_descriptors(torch.from_numpy(desc).to(cdev), torch.from_numpy(counts).to(cdev))
    all_kps = vd.all_gather_rows(torch.from_numpy(kps).to(cdev))
    all_readable = vd.all_gather_rows(torch.from_numpy(readable).to(cdev)).cpu().numpy()[:n]
    shape0 = vd.broadcast_array(first_shape, 0, device)
    if not all_readable[0]:
        raise ValueError(f"Failed to read first image: {image_files[0]}")

    # ---- rank 0: image rows in file order, then features ----------------------------------------------------------------
    stats = dict(images=int(all_readable.sum()), ranks=world, pairs=0, matches=0, verified_pairs=0)
    ids = None
    db = None
    err = None
    cnt = all_counts.cpu().numpy()
    kp_np = all_kps.cpu().numpy()
    if rank == 0:
        try:
            db = ColmapDatabase(str(db_path))
            height, width = int(shape0[0]), int(shape0[1])
            if camera_params is None:
                camera_params = default_camera_params(camera_model, width, height)
            cam = db.db.write_camera(Camera(model=camera_model, width=width, height=height, params=camera_params))
            ids = [db.add_image(f.name, camera_id=cam) if all_readable[k] else None for k, f in enumerate(image_files)]
            d_np = all_desc.cpu().numpy()
            for k, image_id in enumerate(ids):
                if image_id is not None and cnt[k] > 0:
                    db.add_keypoints(image_id, kp_np[k, : cnt[k]])
                    db.add_descriptors(image_id, d_np[k, : cnt[k]])
            db.commit()
        except Exception as e:  # noqa: BLE001 - handed to every rank below
            err = e
    try:
        vd.raise_if_any_failed(err, "writing images / features")
        if not do_matching:
            return stats
        # ---- matching + verification: images with a database row, in id order; pairs dealt round-robin -----------------
        from ..matching.two_view import verify_pair_lists, write_two_view_rows

        ids = vd.broadcast_object(ids, 0)                                   # pair ids seed the verification sampler
        keep = np.nonzero(all_readable)[0]
        m = len(keep)
        stats["pairs"] = m * (m - 1) // 2
        sift = _sift_options(matching_options, None)
        r_, d_, c_ = float(sift.max_ratio), float(sift.max_distance), bool(sift.cross_check)
        blocks = all_desc[torch.from_numpy(keep).to(all_desc.device)]
        bcounts = all_counts[torch.from_numpy(keep).to(all_counts.device)]
        kept_ids = [ids[k] for k in keep]
        my_pairs = vd.pairs_for_rank(m, rank, world)
        err, lists, results = None, [], None
        try:
            if match_fn is None:
                lists = hip_match_blocks(blocks, bcounts, my_pairs, r_, d_, c_, device=device)
            else:
                lists = match_fn(blocks.cpu().numpy(), bcounts.cpu().numpy(), my_pairs, r_, d_, c_)
            if verify:                                                          # every rank verifies the pairs it matched
                kps = {i: kp_np[k, : cnt[k], :2] for i, k in enumerate(keep)}
                results = verify_pair_lists(kps, kept_ids, my_pairs, lists, device=device, verify_fn=verify_fn)
        except Exception as e:  # noqa: BLE001
            err = e
        vd.raise_if_any_failed(err, "matching / verification")
        merged = vd.gather_pair_lists(my_pairs, lists, dst=0)
        verified = vd.gather_pair_results(my_pairs, results, dst=0) if results is not None else None
        err = None
        if rank == 0:
            try:
                for (a, b), lst in sorted(merged.items()):
                    db.db.write_matches(kept_ids[a], kept_ids[b], lst, commit=False)
                    stats["matches"] += len(lst)
                db.commit()
                if verified is not None:
                    stats["verified_pairs"] = write_two_view_rows(db.db, kept_ids, verified)
            except Exception as e:  # noqa: BLE001
                err = e
        vd.raise_if_any_failed(err, "writing matches / two-view geometries")
        return vd.broadcast_object(stats, 0)                                    # every rank returns rank 0's totals
    finally:
        if db is not None:
            db.db.close()
