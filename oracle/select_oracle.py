"""CPU ORACLE (test infrastructure, NOT product code) — keypoint selection + descriptor path.

A numpy restatement of what the reference computes after the ViT forward
(`/root/reference/vit_colmap/features/vit_extractor.py:168-653`), one function per reference
method, each citing the lines it follows.  It is pinned against golden vectors produced by the
reference itself (`tests/golden/make_golden.py` -> `tests/golden/select_*.npz`,
checked by `tests/test_oracle_golden.py`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product path (`vit_colmap_amd/`) never does: it runs the HIP kernels or fails.

All arithmetic is float32 unless noted, in the operation order of the reference, so results
agree with the golden vectors to summation-order noise (score maps) and exactly (indices,
given an identical score map).

Tie rule.  The reference orders candidates with `torch.topk` / `torch.argsort`, whose tie order
is unspecified (and differs between its CPU and CUDA back ends).  This oracle — and the HIP
kernels that are tested against it — define the total order **score descending, then position
ascending** (position = flat index inside the bin for the per-bin top-k, list position for the
global top-k and for the NMS sort).  Golden cases are generated tie-free.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------------
# Gaussian kernels and zero-padded correlation
# --------------------------------------------------------------------------------------------
def gaussian_kernel_1d(kernel_size: int, sigma: float) -> np.ndarray:
    """vit_extractor.py:396-402 — exp(-x^2 / (2 sigma^2)), normalised to sum 1, float32."""
    x = np.arange(kernel_size, dtype=F32) - F32(kernel_size // 2)
    g = np.exp(-(x * x) / F32(2.0 * sigma * sigma)).astype(F32)
    return (g / g.sum(dtype=F32)).astype(F32)


def gaussian_kernel_2d(kernel_size: int, sigma: float) -> np.ndarray:
    """vit_extractor.py:401 — outer product of the normalised 1-D kernel with itself."""
    k = gaussian_kernel_1d(kernel_size, sigma)
    return (k[None, :] * k[:, None]).astype(F32)


def correlate2d_zero_pad(img: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """`F.conv2d(x, k, padding=k//2)` for one channel (vit_extractor.py:318-326, 383-384):
    cross-correlation with zero padding, float32 accumulate."""
    H, W = img.shape
    kh, kw = kern.shape
    ph, pw = kh // 2, kw // 2
    padded = np.zeros((H + 2 * ph, W + 2 * pw), dtype=F32)
    padded[ph:ph + H, pw:pw + W] = img
    out = np.zeros((H, W), dtype=F32)
    for dy in range(kh):
        for dx in range(kw):
            out += kern[dy, dx] * padded[dy:dy + H, dx:dx + W]
    return out


def _minmax01(m: np.ndarray) -> np.ndarray:
    """vit_extractor.py:344-346 / 390-392 — subtract min; divide by the new max if it is > 0."""
    m = (m - m.min()).astype(F32)
    mx = m.max()
    if mx > 0:
        m = (m / mx).astype(F32)
    return m


# --------------------------------------------------------------------------------------------
# Score maps
# --------------------------------------------------------------------------------------------
def structure_tensor_means(fmap: np.ndarray):
    """vit_extractor.py:298-309 — forward differences, zero in the last column / row, then the
    channel MEANS of gx^2, gy^2, gx*gy.  fmap is (C, H, W) float32."""
    C, H, W = fmap.shape
    gx = np.zeros_like(fmap)
    gy = np.zeros_like(fmap)
    gx[:, :, : W - 1] = fmap[:, :, 1:] - fmap[:, :, : W - 1]
    gy[:, : H - 1, :] = fmap[:, 1:, :] - fmap[:, : H - 1, :]
    ixx = (gx * gx).mean(axis=0, dtype=F32)
    iyy = (gy * gy).mean(axis=0, dtype=F32)
    ixy = (gx * gy).mean(axis=0, dtype=F32)
    return ixx.astype(F32), iyy.astype(F32), ixy.astype(F32)


def harris_from_tensor(ixx, iyy, ixy) -> np.ndarray:
    """vit_extractor.py:312-348 — 3x3 sigma=1 smoothing, R = det - 0.04 tr^2,
    edge = sqrt(Ixx + Iyy), 0.7 R + 0.3 edge, min-max."""
    g = gaussian_kernel_2d(3, 1.0)
    ixx = correlate2d_zero_pad(ixx, g)
    iyy = correlate2d_zero_pad(iyy, g)
    ixy = correlate2d_zero_pad(ixy, g)
    det = ixx * iyy - ixy * ixy
    trace = ixx + iyy
    corner = det - F32(0.04) * (trace * trace)
    edge = np.sqrt(ixx + iyy)
    score = F32(0.7) * corner + F32(0.3) * edge
    return _minmax01(score.astype(F32))


def harris_response(fmap: np.ndarray) -> np.ndarray:
    """vit_extractor.py:281-348."""
    return harris_from_tensor(*structure_tensor_means(np.asarray(fmap, dtype=F32)))


def dog_response(fmap: np.ndarray) -> np.ndarray:
    """vit_extractor.py:350-394 — channel mean, Gaussians sigma 1.0 (k=7) and 1.6 (k=11),
    |difference|, min-max."""
    avg = np.asarray(fmap, dtype=F32).mean(axis=0, dtype=F32).astype(F32)

    def ksize(sigma):
        k = int(6 * sigma + 1)
        return k if k % 2 == 1 else k + 1

    s1 = correlate2d_zero_pad(avg, gaussian_kernel_2d(ksize(1.0), 1.0))
    s2 = correlate2d_zero_pad(avg, gaussian_kernel_2d(ksize(1.6), 1.6))
    return _minmax01(np.abs(s1 - s2).astype(F32))


def distinctiveness(fmap: np.ndarray, method: str = "harris") -> np.ndarray:
    """vit_extractor.py:254-279."""
    if method == "harris":
        return harris_response(fmap)
    if method == "dog":
        return dog_response(fmap)
    if method == "combined":
        h = harris_response(fmap)
        d = dog_response(fmap)
        h = (h - h.min()) / (h.max() - h.min() + F32(1e-8))
        d = (d - d.min()) / (d.max() - d.min() + F32(1e-8))
        return (F32(0.5) * h.astype(F32) + F32(0.5) * d.astype(F32)).astype(F32)
    raise ValueError(f"Unknown detection method: {method}")


# --------------------------------------------------------------------------------------------
# Selection (integer work on float keys)
# --------------------------------------------------------------------------------------------
def _topk_desc_stable(values: np.ndarray, k: int):
    """Top-k by (value desc, position asc) — the oracle's total order."""
    order = np.argsort(-values.astype(np.float64), kind="stable")[:k]
    return values[order], order


def spatial_binning_selection(score: np.ndarray, target: int, bin_size: int = 16):
    """vit_extractor.py:404-485.  Returns coords int64 (K,2) as (y,x) and scores float32 (K,)."""
    H, W = score.shape
    nbh = max(1, H // bin_size)
    nbw = max(1, W // bin_size)
    per_bin = max(1, target // (nbh * nbw))
    coords, scores = [], []
    for i in range(nbh):
        for j in range(nbw):
            y0, y1 = i * bin_size, min((i + 1) * bin_size, H)
            x0, x1 = j * bin_size, min((j + 1) * bin_size, W)
            tile = score[y0:y1, x0:x1]
            if tile.size == 0:
                continue
            flat = tile.reshape(-1)
            k = min(per_bin, flat.size)
            vals, idx = _topk_desc_stable(flat, k)
            bw = x1 - x0
            coords.append(np.stack([idx // bw + y0, idx % bw + x0], axis=1))
            scores.append(vals)
    if not coords:
        return np.zeros((0, 2), np.int64), np.zeros((0,), F32)
    coords = np.concatenate(coords, axis=0).astype(np.int64)
    scores = np.concatenate(scores, axis=0).astype(F32)
    if len(coords) > target:
        scores, idx = _topk_desc_stable(scores, target)
        coords = coords[idx]
    return coords, scores


def simple_topk_selection(score: np.ndarray, k: int):
    """vit_extractor.py:487-498."""
    flat = score.reshape(-1)
    k = min(k, flat.size)
    vals, idx = _topk_desc_stable(flat, k)
    W = score.shape[1]
    return np.stack([idx // W, idx % W], axis=1).astype(np.int64), vals.astype(F32)


def apply_nms(coords: np.ndarray, scores: np.ndarray, nms_radius: float = 1.5):
    """vit_extractor.py:500-543 — greedy suppression in score order; a kept point removes every
    other point at Euclidean distance d with 0 < d < radius.  Output stays in score order."""
    n = len(coords)
    if n == 0:
        return coords, scores
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    c = coords[order].astype(np.int64)
    s = scores[order]
    keep = np.ones(n, dtype=bool)
    r2 = float(nms_radius) ** 2
    for i in range(n):
        if not keep[i]:
            continue
        d2 = ((c - c[i]) ** 2).sum(axis=1)
        # sqrt is monotone and exact at 0, so (0 < d < r) == (0 < d^2 < r^2) for integer d^2
        # and r^2 = 2.25 (float32 sqrt of 1, 2, 4 ... is exact or far from 1.5).
        keep[(d2 > 0) & (d2 < r2)] = False
    return c[keep], s[keep]


# --------------------------------------------------------------------------------------------
# Descriptors
# --------------------------------------------------------------------------------------------
def gather_descriptors(fmap: np.ndarray, coords: np.ndarray) -> np.ndarray:
    """vit_extractor.py:545-586 — `grid_sample(bilinear, border, align_corners=True)` at
    coordinates normalised as 2*c/(dim-1)-1.  Restated with the float32 steps torch takes
    (normalise, un-normalise ((g+1)/2)*(size-1), clamp to [0,size-1], 4 taps); at integer
    coordinates this is the token at (y,x) up to a ~1e-7 weight error."""
    C, H, W = fmap.shape
    n = len(coords)
    if n == 0:
        return np.zeros((0, C), F32)
    cy = coords[:, 0].astype(F32)
    cx = coords[:, 1].astype(F32)
    gy = F32(2.0) * cy / F32(H - 1) - F32(1.0)
    gx = F32(2.0) * cx / F32(W - 1) - F32(1.0)
    iy = ((gy + F32(1.0)) / F32(2.0)) * F32(H - 1)
    ix = ((gx + F32(1.0)) / F32(2.0)) * F32(W - 1)
    iy = np.clip(iy, F32(0), F32(H - 1)).astype(F32)
    ix = np.clip(ix, F32(0), F32(W - 1)).astype(F32)
    y0 = np.floor(iy)
    x0 = np.floor(ix)
    wy1 = (iy - y0).astype(F32)
    wx1 = (ix - x0).astype(F32)
    wy0 = (F32(1.0) - wy1).astype(F32)
    wx0 = (F32(1.0) - wx1).astype(F32)
    y0i = y0.astype(np.int64)
    x0i = x0.astype(np.int64)
    y1i = y0i + 1
    x1i = x0i + 1

    def tap(yi, xi, w):
        inside = (yi >= 0) & (yi < H) & (xi >= 0) & (xi < W)
        v = fmap[:, np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]  # (C, n)
        return np.where(inside[None, :], v, F32(0)) * w[None, :]

    out = tap(y0i, x0i, wy0 * wx0) + tap(y0i, x1i, wy0 * wx1) \
        + tap(y1i, x0i, wy1 * wx0) + tap(y1i, x1i, wy1 * wx1)
    return np.ascontiguousarray(out.T.astype(F32))


def sample_descriptors(fmap: np.ndarray, fy: np.ndarray, fx: np.ndarray) -> np.ndarray:
    """`grid_sample(bilinear, border, align_corners=True)` at fractional grid coordinates (fy, fx) float32 — the
    arithmetic of gather_descriptors without the integer assumption (hybrid_extractor.py:256-277)."""
    C, H, W = fmap.shape
    gy = F32(2.0) * fy.astype(F32) / F32(H - 1) - F32(1.0)
    gx = F32(2.0) * fx.astype(F32) / F32(W - 1) - F32(1.0)
    iy = np.clip(((gy + F32(1.0)) / F32(2.0)) * F32(H - 1), F32(0), F32(H - 1)).astype(F32)
    ix = np.clip(((gx + F32(1.0)) / F32(2.0)) * F32(W - 1), F32(0), F32(W - 1)).astype(F32)
    y0, x0 = np.floor(iy), np.floor(ix)
    wy1, wx1 = (iy - y0).astype(F32), (ix - x0).astype(F32)
    wy0, wx0 = (F32(1.0) - wy1).astype(F32), (F32(1.0) - wx1).astype(F32)
    y0i, x0i = y0.astype(np.int64), x0.astype(np.int64)

    def tap(yi, xi, w):
        inside = (yi >= 0) & (yi < H) & (xi >= 0) & (xi < W)
        v = fmap[:, np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]
        return np.where(inside[None, :], v, F32(0)) * w[None, :]

    out = tap(y0i, x0i, wy0 * wx0) + tap(y0i, x0i + 1, wy0 * wx1) + tap(y0i + 1, x0i, wy1 * wx0) + tap(y0i + 1, x0i + 1, wy1 * wx1)
    return np.ascontiguousarray(out.T.astype(F32))


def rootsift_normalize(desc: np.ndarray) -> np.ndarray:
    """hybrid_extractor.py:285-288: F.normalize(p=1) (eps 1e-12), sqrt(clamp(min=1e-8)), F.normalize(p=2)."""
    d = desc.astype(F32)
    l1 = np.maximum(np.abs(d).sum(axis=1, keepdims=True, dtype=F32), F32(1e-12))
    d = np.sqrt(np.maximum(d / l1, F32(1e-8))).astype(F32)
    return l2_normalize(d)


def descriptors_at_keypoints(fmap, keypoints_xy, original_wh, feature_wh, descriptor_dim, projection=None):
    """Reference hybrid extractor, `_extract_descriptors_at_keypoints` (hybrid_extractor.py:224-294):
    keypoints (N, 2) float32 in original-image pixels -> (uint8 (N, D), float32 (N, D))."""
    fmap = np.asarray(fmap, F32)
    C, H, W = fmap.shape
    kp = np.asarray(keypoints_xy, F32).reshape(-1, 2)
    if len(kp) == 0:
        return np.zeros((0, descriptor_dim), np.uint8), np.zeros((0, descriptor_dim), F32)
    w_o, h_o = original_wh
    w_f, h_f = feature_wh
    fx = (kp[:, 0] * F32(w_f / w_o)) * F32(W / w_f)       # float32 array times Python doubles, one after the other
    fy = (kp[:, 1] * F32(h_f / h_o)) * F32(H / h_f)
    desc = sample_descriptors(fmap, fy, fx)
    if desc.shape[1] > descriptor_dim:
        if projection is None:
            raise ValueError("projection matrix must be supplied (it is an input, not refit)")
        desc = project(desc, projection)
    desc = rootsift_normalize(desc)
    return quantize_u8(desc), desc


def map_keypoints(coords: np.ndarray, grid_hw, resized_wh, original_wh) -> np.ndarray:
    """vit_extractor.py:229-236 — (x + 0.5) * (w_resized / W) * (w_orig / w_resized), float32
    tensor times Python-double scalars applied one after the other; columns are (x, y)."""
    H, W = grid_hw
    w_r, h_r = resized_wh
    w_o, h_o = original_wh
    sx1, sy1 = w_r / W, h_r / H
    sx2, sy2 = w_o / w_r, h_o / h_r
    x = (coords[:, 1].astype(F32) + F32(0.5)) * F32(sx1) * F32(sx2)
    y = (coords[:, 0].astype(F32) + F32(0.5)) * F32(sy1) * F32(sy2)
    return np.stack([x, y], axis=1).astype(F32)


def project(desc: np.ndarray, projection: np.ndarray) -> np.ndarray:
    """vit_extractor.py:651 — desc @ P, no re-centring.  The matrix is an INPUT (the reference
    fits it once by SVD or unseeded randn, vit_extractor.py:601-648)."""
    return (desc.astype(F32) @ projection.astype(F32)).astype(F32)


def l2_normalize(desc: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """vit_extractor.py:243 — F.normalize(p=2, dim=1): x / max(||x||_2, eps)."""
    n = np.sqrt((desc * desc).sum(axis=1, dtype=F32)).astype(F32)
    return (desc / np.maximum(n, F32(eps))[:, None]).astype(F32)


def quantize_u8(desc_f32: np.ndarray) -> np.ndarray:
    """vit_extractor.py:250 — (d * 512).clip(0, 255).astype(uint8): truncation, negatives -> 0."""
    return np.clip(desc_f32.astype(F32) * F32(512.0), F32(0), F32(255)).astype(np.uint8)


def dense_to_sparse(fmap, original_wh, resized_wh, num_keypoints, descriptor_dim,
                    method="harris", projection=None, score=None):
    """vit_extractor.py:168-252.  `score` may be supplied to run the integer stages on an
    externally computed map (that is how the HIP kernels are checked bit-exactly)."""
    fmap = np.asarray(fmap, dtype=F32)
    C, H, W = fmap.shape
    if score is None:
        score = distinctiveness(fmap, method)
    coords, scores = spatial_binning_selection(score, num_keypoints, 16)
    if len(coords) == 0:
        coords, scores = simple_topk_selection(score, num_keypoints)
    coords, scores = apply_nms(coords, scores, 1.5)
    desc = gather_descriptors(fmap, coords)
    kp = map_keypoints(coords, (H, W), resized_wh, original_wh)
    if desc.shape[1] > descriptor_dim:
        if projection is None:
            raise ValueError("projection matrix must be supplied (it is an input, not refit)")
        desc = project(desc, projection)
    desc = l2_normalize(desc)
    return dict(score=score, coords=coords, scores=scores, keypoints=kp,
                desc_f32=desc, desc_u8=quantize_u8(desc))


# --------------------------------------------------------------------------------------------
# DummyExtractor rule
# --------------------------------------------------------------------------------------------
def dummy_features(height: int, width: int, step: int = 32, seed: int = 42):
    """dummy_extractor.py:95-111 — grid keypoints at step/2 + k*step (rows ordered y-major,
    columns (x, y)), descriptor = RandomState(seed + 1000*int(x/step) + int(y/step))
    .randint(0, 256, 128) — a function of the grid cell only."""
    ys = np.arange(step // 2, height, step, dtype=F32)
    xs = np.arange(step // 2, width, step, dtype=F32)
    kp = np.empty((len(ys) * len(xs), 2), F32)
    kp[:, 0] = np.tile(xs, len(ys))
    kp[:, 1] = np.repeat(ys, len(xs))
    desc = np.empty((len(kp), 128), np.uint8)
    for r, (x, y) in enumerate(kp):
        cell = seed + int(x / step) * 1000 + int(y / step)
        desc[r] = np.random.RandomState(cell).randint(0, 256, size=128, dtype=np.uint8)
    return kp, desc


def default_camera_params(model: str, width: int, height: int):
    """vit_extractor.py:706-716 — f = max(w, h), principal point at the image centre."""
    f = max(width, height)
    if model == "SIMPLE_PINHOLE":
        return [f, width / 2.0, height / 2.0]
    if model == "PINHOLE":
        return [f, f, width / 2.0, height / 2.0]
    raise ValueError(f"Unsupported camera model: {model}")

