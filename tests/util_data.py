"""Seeded synthetic descriptor sets shared by CPU and GPU tests (numpy only)."""
import numpy as np


def quantize(x):
    x = x / np.sqrt((x * x).sum(axis=1, keepdims=True))
    return np.clip(x * 512.0, 0, 255).astype(np.uint8)


def image_set(seed, n_images, n_max, d, kind="scene", counts=None, noise=0.2):
    """uint8 [n_images][n_max][d] + counts.
    kind "scene": every image sees a shuffled, noisy subset of one non-negative descriptor pool
                  (SIFT-like rows of length ~512: a realistic share of rows matches);
         "vit":   signed normal rows, L2-normalised, negatives clipped (what the reference's
                  quantiser produces: vit_extractor.py:243-250);
         "full":  uniform 0..255 bytes (DummyExtractor-like, saturating similarities).
    """
    rs = np.random.RandomState(seed)
    desc = np.zeros((n_images, n_max, d), np.uint8)
    if counts is None:
        counts = np.full(n_images, n_max, np.int32)
    counts = np.asarray(counts, np.int32)
    pool = np.abs(rs.standard_normal((2 * n_max, d))).astype(np.float32)
    for k in range(n_images):
        n = int(counts[k])
        if n == 0:
            continue
        if kind == "scene":
            sel = rs.permutation(2 * n_max)[:n]
            x = np.abs(pool[sel] + noise * rs.standard_normal((n, d)).astype(np.float32))
            desc[k, :n] = quantize(x)
        elif kind == "vit":
            desc[k, :n] = quantize(rs.standard_normal((n, d)).astype(np.float32))
        elif kind == "full":
            desc[k, :n] = rs.randint(0, 256, (n, d)).astype(np.uint8)
        else:
            raise ValueError(kind)
    return desc, counts


def synthetic_descriptors(k: int, n: int, d: int) -> np.ndarray:
    """Matcher micro-bench input: RandomState(2000+k) normal (n, d),
    L2-normalised, quantised with the reference's own rule (vit_extractor.py:250)."""
    x = np.random.RandomState(2000 + k).standard_normal((n, d)).astype(np.float32)
    x /= np.sqrt((x * x).sum(axis=1, keepdims=True, dtype=np.float32))
    return np.clip(x * np.float32(512.0), 0, 255).astype(np.uint8)
