"""Seeded inputs of the hybrid-extractor goldens (shared by make_golden_hybrid.py and the tests; numpy only)."""
import numpy as np

CASES = [
    dict(name="s384", seed=11, C=384, H=34, W=45, n=300, dd=384, original_wh=(640, 480), feature_wh=(630, 476)),
    dict(name="b768_p128", seed=12, C=768, H=34, W=45, n=257, dd=128, original_wh=(640, 480), feature_wh=(630, 476)),
    dict(name="dtu_p128", seed=13, C=384, H=85, W=114, n=500, dd=128, original_wh=(1600, 1200), feature_wh=(1596, 1190)),
    dict(name="border", seed=14, C=64, H=7, W=9, n=40, dd=64, original_wh=(126, 98), feature_wh=(126, 98)),
]


def make_inputs(case):
    """-> feature map float32 (C, H, W), keypoints float32 (n, 2) in original pixels, projection float32 (C, dd) or None."""
    rs = np.random.RandomState(case["seed"])
    fmap = rs.standard_normal((case["C"], case["H"], case["W"])).astype(np.float32)
    w, h = case["original_wh"]
    kp = np.stack([rs.uniform(0, w, case["n"]), rs.uniform(0, h, case["n"])], axis=1).astype(np.float32)
    # corners, exact cell centres and points beyond the image (the border clamp of grid_sample)
    kp[:6] = [[0, 0], [w - 1, h - 1], [w, h], [w * 1.2, -3.0], [14.0, 14.0], [w / 2, h / 2]]
    proj = None
    if case["C"] > case["dd"]:
        proj = (rs.standard_normal((case["C"], case["dd"])) / np.sqrt(case["C"])).astype(np.float32)
    return fmap, kp, proj
