"""CPU ORACLE (test infrastructure, NOT product code) — TrainableViTExtractor post-model path.

A numpy restatement of what the reference does with the outputs of its `ViTFeatureModel`
(`/root/reference/vit_colmap/features/trainable_vit_extractor.py:170-267`): sigmoid, max-pool NMS (`:114-138`),
score threshold, top-k, sub-pixel offsets with the x4 and original-size scaling (`:219-237`), the 6-column keypoint
rows (`:244-254`) and the `(d + 1) * 127.5` descriptor quantiser (`:265-267`).  Pinned by golden vectors that the
reference's own `_run_inference` produced here (`tests/golden/make_golden_trainable.py` ->
`tests/golden/trainable_*.npz`, checked by `tests/test_trainable_oracle.py`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.

Spec choices where the reference leaves room:
* sigmoid(x) := float32(1 / (1 + exp(-float64(x)))) — the correctly rounded value, so that device and oracle agree on
  every input; torch's float32 sigmoid differs from it by at most 1 ulp (the golden test compares the score column
  with that tolerance and everything else exactly).
* ties in `torch.topk` (order unspecified): score descending, then position (y * W + x) ascending.
"""
import numpy as np

F32 = np.float32


def sigmoid32(x: np.ndarray) -> np.ndarray:
    """trainable_vit_extractor.py:181 — torch.sigmoid, as the correctly rounded float32 value."""
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def simple_nms(scores: np.ndarray, radius: int) -> np.ndarray:
    """trainable_vit_extractor.py:114-138 — keep = scores == max_pool2d(scores, 2r+1, stride 1, padding r)
    (max_pool2d pads with -inf)."""
    H, W = scores.shape
    p = np.full((H + 2 * radius, W + 2 * radius), -np.inf, F32)
    p[radius:radius + H, radius:radius + W] = scores
    mp = np.full((H, W), -np.inf, F32)
    for dy in range(2 * radius + 1):
        for dx in range(2 * radius + 1):
            mp = np.maximum(mp, p[dy:dy + H, dx:dx + W])
    return scores == mp


def select(kp_map: np.ndarray, num_keypoints: int, score_threshold: float, nms_radius: int):
    """trainable_vit_extractor.py:174-210 — returns (flat positions (K,) int64 in selection order, scores (K,) f32)."""
    s = sigmoid32(kp_map[0])
    valid = (s > F32(score_threshold)) & simple_nms(s, nms_radius)
    pos = np.flatnonzero(valid.reshape(-1))                      # row-major = torch.nonzero order
    sc = s.reshape(-1)[pos]
    order = np.lexsort((pos, -sc.astype(np.float64)))            # score descending, position ascending
    order = order[: min(num_keypoints, len(order))]
    return pos[order].astype(np.int64), sc[order].astype(F32)


def run_inference_post(kp_map, desc_map, orig_hw, new_hw, num_keypoints, score_threshold, nms_radius):
    """trainable_vit_extractor.py:170-267 from the model outputs on.
    kp_map (4, H, W) float32 = logit, dx, dy, orientation; desc_map (D, H, W) float32 (unit norm).
    Returns keypoints (K, 6) float32 = (x, y, 1, orientation, score, 0) and descriptors (K, D) uint8."""
    D, H, W = desc_map.shape
    pos, sc = select(kp_map, num_keypoints, score_threshold, nms_radius)
    if len(pos) == 0:                                            # :195-200
        return np.zeros((0, 6), F32), np.zeros((0, D), np.uint8)
    ys, xs = pos // W, pos % W
    dx, dy, ori = kp_map[1].reshape(-1)[pos], kp_map[2].reshape(-1)[pos], kp_map[3].reshape(-1)[pos]
    h_o, w_o = orig_hw
    h_n, w_n = new_hw
    sx, sy = F32(w_o / w_n), F32(h_o / h_n)                      # python float scalars act as float32 on a float32 tensor
    x = (((xs.astype(F32) + dx) + F32(0.5)) * F32(4.0)) * sx     # :226-231
    y = (((ys.astype(F32) + dy) + F32(0.5)) * F32(4.0)) * sy
    x = np.clip(x, F32(0), F32(w_o - 1)).astype(F32)             # :234-235
    y = np.clip(y, F32(0), F32(h_o - 1)).astype(F32)
    kps = np.stack([x, y, np.ones_like(x), ori.astype(F32), sc, np.zeros_like(x)], axis=1).astype(F32)
    d = desc_map.reshape(D, -1)[:, pos].T                        # :257
    du8 = ((d + F32(1.0)) * F32(127.5)).clip(0, 255).astype(np.uint8)   # :265-267
    return kps, du8
