"""Seeded synthetic inputs for the golden vectors (numpy only; shared by generator and tests).

Feature maps are regenerated from `np.random.RandomState(seed)` (stable across numpy versions)
and never stored. `kind`:
  * "white"  — i.i.d. standard normal.
  * "smooth" — low-frequency field + 10 % white noise, closer to what ViT patch tokens look like
               (neighbouring tokens correlate), so Harris scores are not dominated by noise.
Layout is the reference's (C, H, W) fp32 (`vit_extractor.py:194`).
"""
import numpy as np

CASES = [
    # BASELINE config 2/3: ViT-S grid 34x45, 512 keypoints, 384-D (no projection)
    dict(name="c2_s384", seed=0, kind="white", C=384, H=34, W=45, num_keypoints=512,
         descriptor_dim=384, method="harris", orig_wh=(640, 480)),
    dict(name="c2_smooth", seed=6, kind="smooth", C=384, H=34, W=45, num_keypoints=512,
         descriptor_dim=384, method="harris", orig_wh=(640, 480)),
    # other detection methods
    dict(name="dog_s384", seed=1, kind="smooth", C=384, H=34, W=45, num_keypoints=512,
         descriptor_dim=384, method="dog", orig_wh=(640, 480)),
    dict(name="comb_s384", seed=7, kind="smooth", C=384, H=34, W=45, num_keypoints=512,
         descriptor_dim=384, method="combined", orig_wh=(640, 480)),
    # BASELINE config 5: ViT-B, 2048 keypoints, stored projection 768 -> 256
    dict(name="c5_b768", seed=2, kind="white", C=768, H=34, W=45, num_keypoints=2048,
         descriptor_dim=256, method="harris", orig_wh=(640, 480)),
    # reference default ctor: ViT-B 768 -> 128
    dict(name="default_b128", seed=8, kind="smooth", C=768, H=34, W=45, num_keypoints=2048,
         descriptor_dim=128, method="harris", orig_wh=(644, 476)),
    # edge: grid smaller than one bin (single clipped bin)
    dict(name="edge_7x9", seed=3, kind="white", C=64, H=7, W=9, num_keypoints=32,
         descriptor_dim=64, method="harris", orig_wh=(126, 98)),
    # edge: exactly one full bin, every cell a candidate
    dict(name="edge_16x16", seed=4, kind="white", C=64, H=16, W=16, num_keypoints=100,
         descriptor_dim=64, method="harris", orig_wh=(224, 224)),
    # DTU-size grid 85x114 (1600x1200 -> 1596x1190): 5x7 bins, ragged right/bottom margins
    dict(name="dtu_85x114", seed=23, kind="smooth", C=96, H=85, W=114, num_keypoints=2048,
         descriptor_dim=96, method="harris", orig_wh=(1600, 1200)),
]

CASE_BY_NAME = {c["name"]: c for c in CASES}


def make_feature_map(case):
    rs = np.random.RandomState(case["seed"])
    C, H, W = case["C"], case["H"], case["W"]
    if case["kind"] == "white":
        return rs.standard_normal((C, H, W)).astype(np.float32)
    coarse = rs.standard_normal((C, H // 4 + 2, W // 4 + 2)).astype(np.float32)
    up = np.kron(coarse, np.ones((1, 4, 4), np.float32))[:, :H, :W]
    # separable 1-2-1 smoothing so 4x4 blocks blend
    for ax in (1, 2):
        up = 0.25 * np.roll(up, 1, ax) + 0.5 * up + 0.25 * np.roll(up, -1, ax)
    noise = rs.standard_normal((C, H, W)).astype(np.float32)
    return np.ascontiguousarray(up + 0.1 * noise, dtype=np.float32)


def make_projection(case):
    """Stored projection matrix fed as an INPUT (never recomputed for parity)."""
    rs = np.random.RandomState(1000 + case["seed"])
    C, dd = case["C"], case["descriptor_dim"]
    return (rs.standard_normal((C, dd)) / np.sqrt(C)).astype(np.float32)
