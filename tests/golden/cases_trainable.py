"""Seeded synthetic head outputs for the TrainableViTExtractor golden vectors (numpy only; shared by the
generator and the tests).  The reference's model emits `keypoints` (4, H/4, W/4) = score logit, dx, dy, orientation
and unit-norm `descriptors` (D, H/4, W/4) (`vit_colmap/model/vit_feature_model.py:214-229`); these functions stand
in for the model so that everything AFTER it (`trainable_vit_extractor.py:170-267`) is exercised.  Inputs are
regenerated from `np.random.RandomState(seed)` and never stored."""
import numpy as np

CASES = [
    # image divisible by 14: no resize, unit scale factors
    dict(name="sq448", seed=11, orig_hw=(448, 448), num_keypoints=2048, descriptor_dim=128, score_threshold=0.0, nms_radius=4),
    # 640x480 -> 630x476 resize branch, map 119x157, fewer candidates than asked for after NMS
    dict(name="vga", seed=12, orig_hw=(480, 640), num_keypoints=2048, descriptor_dim=128, score_threshold=0.0, nms_radius=4),
    # top-k actually truncates; threshold above 0.5
    dict(name="vga_k256", seed=13, orig_hw=(480, 640), num_keypoints=256, descriptor_dim=128, score_threshold=0.55, nms_radius=2),
    # tiny map, radius 1, 64-D descriptors
    dict(name="tiny", seed=14, orig_hw=(61, 75), num_keypoints=50, descriptor_dim=64, score_threshold=0.5, nms_radius=1),
    # nothing passes the threshold: empty outputs
    dict(name="empty", seed=15, orig_hw=(112, 140), num_keypoints=100, descriptor_dim=128, score_threshold=0.999999, nms_radius=4),
    # radius 0: every cell above the threshold is a candidate (top-k over the whole map)
    dict(name="r0", seed=16, orig_hw=(224, 280), num_keypoints=300, descriptor_dim=32, score_threshold=0.0, nms_radius=0),
]
CASE_BY_NAME = {c["name"]: c for c in CASES}


def map_hw(case):
    """(h_new, w_new) after the multiple-of-14 resize and the (H/4, W/4) head resolution."""
    h, w = case["orig_hw"]
    h_new, w_new = (h // 14) * 14, (w // 14) * 14
    return (h_new, w_new), (h_new // 4, w_new // 4)


def make_head_outputs(case):
    rs = np.random.RandomState(case["seed"])
    _, (H, W) = map_hw(case)
    D = case["descriptor_dim"]
    coarse = rs.standard_normal((H // 3 + 2, W // 3 + 2)).astype(np.float32)
    logit = np.kron(coarse, np.ones((3, 3), np.float32))[:H, :W] * 1.5
    for ax in (0, 1):
        logit = 0.25 * np.roll(logit, 1, ax) + 0.5 * logit + 0.25 * np.roll(logit, -1, ax)
    logit = (logit + 0.3 * rs.standard_normal((H, W))).astype(np.float32)
    if case["name"] == "empty":
        logit = (logit - 3.0).astype(np.float32)
    dx = rs.uniform(-0.5, 0.5, (H, W)).astype(np.float32)
    dy = rs.uniform(-0.5, 0.5, (H, W)).astype(np.float32)
    ori = rs.uniform(-np.pi, np.pi, (H, W)).astype(np.float32)
    kp = np.stack([logit, dx, dy, ori]).astype(np.float32)
    d = rs.standard_normal((D, H, W)).astype(np.float32)
    d = (d / np.sqrt((d * d).sum(0, keepdims=True))).astype(np.float32)
    return kp, d
