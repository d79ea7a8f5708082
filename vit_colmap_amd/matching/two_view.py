"""Two-view geometric verification of the putative matches — the step that follows descriptor matching inside
`pycolmap.match_exhaustive` (reference call site vit_colmap/pipeline/run_pipeline.py:351-363) and fills the
`two_view_geometries(rows, config, ...)` rows the reference's matching metrics read
(vit_colmap/utils/metrics.py:207-243).  Specification: oracle/two_view_oracle.py (the build's own RANSAC on F and H
with a published sampler; parity with COLMAP's estimator is unpinned — COLMAP is an absent third-party wheel).

Where the work runs
  HIP     scoring of every hypothesis against every match of every pair, and the inlier masks
          (csrc/two_view.hip, vc_two_view_score / vc_two_view_inliers): O(pairs x hypotheses x matches)
  torch   the minimal solvers: batched 8x8 linear systems (float64) for all pairs and hypotheses at once, the
          normal equations of the one refit, a 3x3 SVD per pair for the stored F — plumbing around the kernels
  host    gathering matched keypoints from the database, writing the rows (rank 0 only in a multi-GPU run)
All pairs are verified in one batch: nothing loops over pairs on the device side.
"""
import numpy as np
import torch

from .. import _lib
from ..database.colmap_db import pair_id_of

CONFIG_UNDEFINED, CONFIG_DEGENERATE, CONFIG_CALIBRATED, CONFIG_UNCALIBRATED = 0, 1, 2, 3
CONFIG_PLANAR, CONFIG_PANORAMIC, CONFIG_PLANAR_OR_PANORAMIC = 4, 5, 6
MIN_NUM_INLIERS = 15
MAX_ERROR = 4.0
MAX_H_INLIER_RATIO = 0.8
MIN_INLIER_RATIO = 0.25
NUM_HYP_F, NUM_HYP_H = 512, 128
NUM_CANDIDATES = 32
SALT = {"F": 0x0F0F0F0F, "H": 0x3C3C3C3C}
MODEL_CODE = {"F": 0, "H": 1}
_M32 = 0xFFFFFFFF


def _lowbias32(x):
    """int64 tensor holding 32-bit values -> lowbias32 hash (Python-int constants keep the products below 2^63)."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    return x ^ (x >> 16)


def _sample_indices(seeds, counts, n_hyp, S, salt):
    """seeds, counts int64 (P,) -> int64 (P, n_hyp, S): the first S distinct values of hash(seed, k, j) mod M, -1 if void."""
    dev = seeds.device
    P = seeds.shape[0]
    k = torch.arange(n_hyp, dtype=torch.int64, device=dev)[None, :, None]
    j = torch.arange(NUM_CANDIDATES, dtype=torch.int64, device=dev)[None, None, :]
    # 32-bit wrap-around arithmetic on int64: every product is reduced before it can reach 2^63
    x = ((seeds[:, None, None] & _M32) * 0x9E3779B1) & _M32
    x = (x + ((k * 0x85EBCA6B) & _M32) + ((j * 0xC2B2AE35) & _M32) + salt) & _M32
    cand = _lowbias32(x) % counts[:, None, None].clamp(min=1)
    chosen = torch.full((P, n_hyp, S), -1, dtype=torch.int64, device=dev)
    count = torch.zeros((P, n_hyp), dtype=torch.int64, device=dev)
    slot = torch.arange(S, dtype=torch.int64, device=dev)[None, None, :]
    for jj in range(NUM_CANDIDATES):
        c = cand[:, :, jj]
        take = ~(chosen == c[:, :, None]).any(dim=2) & (count < S)
        put = take[:, :, None] & (slot == count[:, :, None])
        chosen = torch.where(put, c[:, :, None], chosen)
        count = count + take.to(torch.int64)
    return torch.where((count < S)[:, :, None], torch.full_like(chosen, -1), chosen)


def _rows(model, x1, y1, x2, y2):
    """Equations of the parametrisation with the last matrix entry fixed to 1: A (..., n_eq, 8), b (..., n_eq)."""
    if model == "F":
        return torch.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1], dim=-1), -torch.ones_like(x1)
    z, o = torch.zeros_like(x1), torch.ones_like(x1)
    ax = torch.stack([x1, y1, o, z, z, z, -x2 * x1, -x2 * y1], dim=-1)
    ay = torch.stack([z, z, z, x1, y1, o, -y2 * x1, -y2 * y1], dim=-1)
    return torch.cat([ax, ay], dim=-2), torch.cat([x2, y2], dim=-1)


def _to_matrix(f8):
    return torch.cat([f8, torch.ones(f8.shape[:-1] + (1,), dtype=f8.dtype, device=f8.device)], dim=-1).reshape(f8.shape[:-1] + (3, 3))


def _denormalise(model, Mn, T1, T2):
    """Mn (P, K, 3, 3) in normalised coordinates -> pixel coordinates."""
    if model == "F":
        return T2.transpose(-1, -2)[:, None] @ Mn @ T1[:, None]
    return torch.linalg.inv(T2)[:, None] @ Mn @ T1[:, None]


def _score(pts, offsets, hyp, model, max_error):
    lib = _lib.load()
    P, K, _ = hyp.shape
    counts = torch.zeros((P, K), dtype=torch.int32, device=pts.device)
    _lib.check(lib.vc_two_view_score(_lib.ptr(pts), _lib.ptr(offsets), P, _lib.ptr(hyp), K, MODEL_CODE[model], float(max_error),
                                     _lib.ptr(counts), _lib.stream_ptr()), "vc_two_view_score")
    return counts


def _mask(pts, offsets, models, model, max_error):
    lib = _lib.load()
    mask = torch.zeros((pts.shape[0],), dtype=torch.uint8, device=pts.device)
    _lib.check(lib.vc_two_view_inliers(_lib.ptr(pts), _lib.ptr(offsets), models.shape[0], _lib.ptr(models), MODEL_CODE[model],
                                       float(max_error), _lib.ptr(mask), _lib.stream_ptr()), "vc_two_view_inliers")
    return mask.bool()


def _estimate(model, pts, offsets, pair_of, seeds, n_hyp, max_error):
    """All pairs at once -> best model float32 (P, 9) (NaN where none), inlier mask bool (total,), counts int64 (P,)."""
    dev = pts.device
    P = offsets.shape[0] - 1
    M = (offsets[1:] - offsets[:-1]).to(torch.int64)
    S = 8 if model == "F" else 4
    p64 = pts.to(torch.float64)

    def norm_T(xy):      # per pair: x~ = (x - mean) * sqrt(2) / mean distance
        ones = torch.ones(xy.shape[0], dtype=torch.float64, device=dev)
        cnt = torch.zeros(P, dtype=torch.float64, device=dev).index_add_(0, pair_of, ones).clamp(min=1)
        mu = torch.zeros((P, 2), dtype=torch.float64, device=dev).index_add_(0, pair_of, xy) / cnt[:, None]
        dist = torch.sqrt(((xy - mu[pair_of]) ** 2).sum(dim=1))
        md = torch.zeros(P, dtype=torch.float64, device=dev).index_add_(0, pair_of, dist) / cnt
        s = torch.where(md > 0, np.sqrt(2.0) / md, torch.ones_like(md))
        T = torch.zeros((P, 3, 3), dtype=torch.float64, device=dev)
        T[:, 0, 0] = s
        T[:, 1, 1] = s
        T[:, 0, 2] = -s * mu[:, 0]
        T[:, 1, 2] = -s * mu[:, 1]
        T[:, 2, 2] = 1.0
        return T

    T1, T2 = norm_T(p64[:, :2]), norm_T(p64[:, 2:])
    n1 = p64[:, :2] * T1[pair_of, 0, 0][:, None] + T1[pair_of, :2, 2]
    n2 = p64[:, 2:] * T2[pair_of, 0, 0][:, None] + T2[pair_of, :2, 2]

    idx = _sample_indices(seeds, M, n_hyp, S, SALT[model])                    # (P, K, S) into the pair's own list
    void = idx[:, :, 0] < 0
    g = (idx.clamp(min=0) + offsets[:-1].to(torch.int64)[:, None, None]).clamp(max=max(pts.shape[0] - 1, 0))
    A, b = _rows(model, n1[g, 0], n1[g, 1], n2[g, 0], n2[g, 1])              # (P, K, 8, 8), (P, K, 8)
    sol = torch.linalg.solve_ex(A, b.unsqueeze(-1)).result.squeeze(-1)        # singular systems give inf / nan
    hyp = _denormalise(model, _to_matrix(sol), T1, T2).reshape(P, n_hyp, 9)
    hyp = torch.where(void[:, :, None] | ~torch.isfinite(sol).all(dim=-1, keepdim=True), torch.full_like(hyp, float("nan")), hyp)
    hyp32 = hyp.to(torch.float32).contiguous()
    counts = _score(pts, offsets, hyp32, model, max_error).to(torch.int64)
    # most inliers, lowest k on ties
    key = counts * n_hyp + (n_hyp - 1 - torch.arange(n_hyp, device=dev))[None, :]
    kbest = (n_hyp - 1) - (key.max(dim=1).values % n_hyp)
    best = hyp32[torch.arange(P, device=dev), kbest].contiguous()
    nbest = counts[torch.arange(P, device=dev), kbest]
    mask = _mask(pts, offsets, best, model, max_error)
    # one refit over the inliers of the best hypothesis: normal equations per pair (float64)
    Ar, br = _rows(model, n1[:, 0], n1[:, 1], n2[:, 0], n2[:, 1])             # F: (total, 8); H: (2 total, 8) stacked x then y
    if model == "H":
        w = torch.cat([mask, mask]).to(torch.float64)
        seg = torch.cat([pair_of, pair_of])
    else:
        w, seg = mask.to(torch.float64), pair_of
    AtA = torch.zeros((P, 8, 8), dtype=torch.float64, device=dev).index_add_(0, seg, (Ar[:, :, None] * Ar[:, None, :]) * w[:, None, None])
    Atb = torch.zeros((P, 8), dtype=torch.float64, device=dev).index_add_(0, seg, Ar * (br * w)[:, None])
    rsol = torch.linalg.solve_ex(AtA, Atb.unsqueeze(-1)).result.squeeze(-1)
    refit = _denormalise(model, _to_matrix(rsol)[:, None], T1, T2).reshape(P, 9)
    ok = torch.isfinite(rsol).all(dim=-1) & (nbest >= S)
    refit32 = torch.where(ok[:, None], refit, torch.full_like(refit, float("nan"))).to(torch.float32).contiguous()
    rcount = _score(pts, offsets, refit32[:, None, :].contiguous(), model, max_error).to(torch.int64)[:, 0]
    use = ok & (rcount >= nbest)
    final = torch.where(use[:, None], refit32, best).contiguous()
    fmask = _mask(pts, offsets, final, model, max_error)
    fcount = torch.where(use, rcount, nbest)
    final = torch.where((fcount > 0)[:, None], final, torch.full_like(final, float("nan")))
    return final, fmask, fcount


@torch.no_grad()
def verify_pairs(keypoints, pair_images, pair_ids, match_lists, device="cuda", num_f=NUM_HYP_F, num_h=NUM_HYP_H,
                 max_error=MAX_ERROR, chunk_pairs: int = 1024):
    """keypoints: dict image index -> float32 (N, >= 2); pair_images: list of (a, b); pair_ids: COLMAP pair ids;
    match_lists: list of uint32 (M, 2).  -> list of dict(config, inlier_matches, F, H, n_f, n_h), one per pair."""
    if not torch.cuda.is_available():
        raise _lib.HipLibraryError("geometric verification scores its hypotheses on the GPU (no CPU fallback)")
    results = [dict(config=CONFIG_DEGENERATE, inlier_matches=np.zeros((0, 2), np.uint32), F=np.zeros((3, 3)),
                    H=np.zeros((3, 3)), n_f=0, n_h=0) for _ in pair_images]
    todo = [i for i, m in enumerate(match_lists) if len(m) >= MIN_NUM_INLIERS]
    for c0 in range(0, len(todo), chunk_pairs):
        sel = todo[c0:c0 + chunk_pairs]
        pts_np, offs = [], [0]
        for i in sel:
            a, b = pair_images[i]
            m = np.asarray(match_lists[i], np.int64).reshape(-1, 2)
            pts_np.append(np.concatenate([keypoints[a][m[:, 0], :2], keypoints[b][m[:, 1], :2]], axis=1).astype(np.float32))
            offs.append(offs[-1] + len(m))
        pts = torch.from_numpy(np.concatenate(pts_np)).to(device).contiguous()
        offsets = torch.tensor(offs, dtype=torch.int32, device=device)
        P = len(sel)
        pair_of = torch.repeat_interleave(torch.arange(P, device=device), (offsets[1:] - offsets[:-1]).to(torch.int64))
        seeds = torch.tensor([int(pair_ids[i]) & _M32 for i in sel], dtype=torch.int64, device=device)
        with torch.cuda.device(pts.device):
            f9, fmask, nf = _estimate("F", pts, offsets, pair_of, seeds, num_f, max_error)
            h9, hmask, nh = _estimate("H", pts, offsets, pair_of, seeds, num_h, max_error)
            # stored F: closest rank-2 matrix, unit Frobenius norm
            F = torch.nan_to_num(f9.to(torch.float64)).reshape(P, 3, 3)
            U, s, Vt = torch.linalg.svd(F)
            s = s.clone()
            s[:, 2] = 0
            F2 = U @ torch.diag_embed(s) @ Vt
            nrm = torch.linalg.norm(F2.reshape(P, 9), dim=1).clamp(min=1e-300)
            F2 = (F2 / nrm[:, None, None]).cpu().numpy()
        H = torch.nan_to_num(h9.to(torch.float64)).reshape(P, 3, 3).cpu().numpy()
        fmask, hmask, nf, nh = fmask.cpu().numpy(), hmask.cpu().numpy(), nf.cpu().numpy(), nh.cpu().numpy()
        for q, i in enumerate(sel):
            r = results[i]
            r["n_f"], r["n_h"] = int(nf[q]), int(nh[q])
            if r["n_f"] < max(MIN_NUM_INLIERS, MIN_INLIER_RATIO * len(match_lists[i])):
                continue
            r["F"] = F2[q]
            r["H"] = H[q] / H[q][2, 2] if H[q][2, 2] != 0 else H[q]
            lo, hi = offs[q], offs[q + 1]
            if r["n_h"] / r["n_f"] > MAX_H_INLIER_RATIO:
                r["config"] = CONFIG_PLANAR_OR_PANORAMIC
                mask = hmask[lo:hi] if r["n_h"] > r["n_f"] else fmask[lo:hi]
            else:
                r["config"] = CONFIG_UNCALIBRATED
                mask = fmask[lo:hi]
            r["inlier_matches"] = np.asarray(match_lists[i], np.uint32).reshape(-1, 2)[mask]
    return results


def read_keypoints_by_index(db, ids):
    """{image index: float32 (N, >= 2)} for every image of `ids` (empty array where the database has no keypoints)."""
    kps = {}
    for k, image_id in enumerate(ids):
        kp = None if image_id is None else db.read_keypoints(image_id)
        kps[k] = np.zeros((0, 2), np.float32) if kp is None else np.asarray(kp, np.float32)
    return kps


def verify_pair_lists(kps, ids, pairs, lists, device="cuda", verify_fn=None):
    """This rank's share of the verification: pairs (P, 2) image indices with their match lists -> one result dict per
    pair (verify_pairs' format).  The sampler is seeded by the COLMAP pair id, so a pair's result does not depend on the
    rank that verifies it.  `verify_fn(kps, pair_images, pair_ids, lists)` replaces verify_pairs in the CPU tests."""
    pair_images = [(int(a), int(b)) for a, b in pairs]
    pids = [pair_id_of(ids[a], ids[b]) for a, b in pair_images]
    if verify_fn is not None:
        return verify_fn(kps, pair_images, pids, lists)
    return verify_pairs(kps, pair_images, pids, lists, device=device)


def write_two_view_rows(db, ids, results) -> int:
    """results {(a, b): verify_pairs result} -> two_view_geometries rows in pair order (one per matched pair, as COLMAP
    does [recalled]: pairs that fail keep config DEGENERATE and zero inlier rows).  Returns the number of verified pairs."""
    n_ok = 0
    for (a, b) in sorted(results):
        r = results[(a, b)]
        db.write_two_view_geometry(ids[a], ids[b], r["inlier_matches"], r["config"], F=r["F"], H=r["H"], commit=False)
        n_ok += r["config"] != CONFIG_DEGENERATE
    db.commit()
    return int(n_ok)


def verify_database_pairs(db, ids, merged, device="cuda", verify_fn=None) -> int:
    """Single-process form: verify every matched pair of a database that is open for writing and write its rows.
    `merged`: {(a, b): uint32 (M, 2)} with a < b image indices into `ids`.  Returns the number of verified pairs."""
    pairs = sorted(merged)
    res = verify_pair_lists(read_keypoints_by_index(db, ids), ids, pairs, [merged[p] for p in pairs], device=device,
                            verify_fn=verify_fn)
    return write_two_view_rows(db, ids, dict(zip(pairs, res)))
