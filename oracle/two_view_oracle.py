"""CPU ORACLE (test infrastructure, NOT product code) — two-view geometric verification of putative matches.

What it restates: the verification step that follows descriptor matching inside `pycolmap.match_exhaustive`
(reference call site /root/reference/vit_colmap/pipeline/run_pipeline.py:351-363) and produces the
`two_view_geometries(rows, config, ...)` rows every matching metric of the reference reads
(/root/reference/vit_colmap/utils/metrics.py:207-243).  COLMAP's source and wheel are absent from the container
(pycolmap==3.12.6, uv.lock:773-774), so this is a published specification of the build's own estimator in the
spirit of COLMAP's uncalibrated two-view geometry [recalled: RANSAC on F and H, max_error 4 px, min_num_inliers 15,
max_H_inlier_ratio 0.8, configs DEGENERATE=1 / UNCALIBRATED=3 / PLANAR_OR_PANORAMIC=6]: PARITY UNPINNED against
COLMAP (which uses 7-point F, LO-RANSAC with its own sampler and a dynamic trial count).

Specification (the HIP scoring kernels in vit_colmap_amd/csrc/two_view.hip and the host code in
vit_colmap_amd/matching/two_view.py follow it step by step):
  sampling    hypothesis k of a pair with M matches uses the first S DISTINCT values of
              r_j = lowbias32(seed * 0x9E3779B1 + k * 0x85EBCA6B + j * 0xC2B2AE35 + salt) mod M, j = 0..31
              (seed = pair_id mod 2^32; fewer than S distinct values: the hypothesis is void)
  normalise   per pair and image: x~ = (x - mean) * sqrt(2) / mean |x - mean|   (float64, all matches of the pair)
  F (S = 8)   f_33 = 1; the eight epipolar equations form an 8x8 linear system (float64); F = T2' F~ T1
  H (S = 4)   h_33 = 1; the four correspondences form an 8x8 linear system (float64);   H = T2^-1 H~ T1
  score       float32, no division:  F: (x2'Fx1)^2 <= e^2 (|Fx1|_xy^2 + |F'x2|_xy^2);  H: |p_xy - x2 p_w|^2 <= e^2 p_w^2
  best        most inliers, lowest k on ties; one refit by linear least squares over the best hypothesis' inliers
              (same parametrisation, normal equations in float64), kept if it has at least as many inliers
  decision    F inliers < max(15, 0.25 M): DEGENERATE, no inlier matches [recalled: COLMAP's min_num_inliers and RANSAC
              min_inlier_ratio];  H inliers / F inliers > 0.8: PLANAR_OR_PANORAMIC
              (inliers of whichever model has more); else UNCALIBRATED (inliers of F)
  stored F    closest rank-2 matrix (SVD) of the chosen F, scaled to unit Frobenius norm; H scaled to h_33 = 1
"""
import numpy as np

CONFIG_UNDEFINED, CONFIG_DEGENERATE, CONFIG_CALIBRATED, CONFIG_UNCALIBRATED = 0, 1, 2, 3
CONFIG_PLANAR, CONFIG_PANORAMIC, CONFIG_PLANAR_OR_PANORAMIC = 4, 5, 6
MIN_NUM_INLIERS = 15
MAX_ERROR = 4.0
MAX_H_INLIER_RATIO = 0.8
MIN_INLIER_RATIO = 0.25
NUM_HYP_F, NUM_HYP_H = 512, 128
NUM_CANDIDATES = 32
SALT_F, SALT_H = 0x0F0F0F0F, 0x3C3C3C3C
M32 = np.uint64(0xFFFFFFFF)


def lowbias32(x):
    x = np.asarray(x, np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def sample_indices(seed: int, n_hyp: int, S: int, M: int, salt: int):
    """-> int64 (n_hyp, S) indices into the pair's match list, -1 in every slot of a void hypothesis."""
    k = np.arange(n_hyp, dtype=np.uint64)[:, None]
    j = np.arange(NUM_CANDIDATES, dtype=np.uint64)[None, :]
    x = (np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x9E3779B1) + k * np.uint64(0x85EBCA6B) + j * np.uint64(0xC2B2AE35)
         + np.uint64(salt)) & M32
    cand = (lowbias32(x) % np.uint64(M)).astype(np.int64)
    chosen = np.full((n_hyp, S), -1, np.int64)
    count = np.zeros(n_hyp, np.int64)
    rows = np.arange(n_hyp)
    for jj in range(NUM_CANDIDATES):
        c = cand[:, jj]
        take = ~(chosen == c[:, None]).any(axis=1) & (count < S)
        chosen[rows[take], count[take]] = c[take]
        count[take] += 1
    chosen[count < S] = -1
    return chosen


def normalisation(xy):
    """(M, 2) float64 -> 3x3 similarity T with T x = (x - mean) * sqrt(2) / mean distance."""
    mu = xy.mean(axis=0)
    dist = np.sqrt(((xy - mu) ** 2).sum(axis=1)).mean()
    s = np.sqrt(2.0) / dist if dist > 0 else 1.0
    return np.array([[s, 0, -s * mu[0]], [0, s, -s * mu[1]], [0, 0, 1.0]])


def _solve(A, b):
    try:
        return np.linalg.solve(A, b)
    except np.linalg.LinAlgError:
        return np.full(8, np.nan)


def rows_f(x1, y1, x2, y2):
    """One equation per correspondence: a . f = -1 with F~ = [[f0 f1 f2] [f3 f4 f5] [f6 f7 1]]."""
    return np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1], axis=-1), -np.ones_like(x1)


def rows_h(x1, y1, x2, y2):
    """Two equations per correspondence with H~ = [[h0 h1 h2] [h3 h4 h5] [h6 h7 1]]."""
    z, o = np.zeros_like(x1), np.ones_like(x1)
    ax = np.stack([x1, y1, o, z, z, z, -x2 * x1, -x2 * y1], axis=-1)
    ay = np.stack([z, z, z, x1, y1, o, -y2 * x1, -y2 * y1], axis=-1)
    return np.concatenate([ax, ay], axis=-2), np.concatenate([x2, y2], axis=-1)


def to_matrix(f8):
    return np.concatenate([f8, np.ones(f8.shape[:-1] + (1,))], axis=-1).reshape(f8.shape[:-1] + (3, 3))


def denormalise(model, Mn, T1, T2):
    return T2.T @ Mn @ T1 if model == "F" else np.linalg.inv(T2) @ Mn @ T1


def inliers_f32(model, m9, pts, max_error=MAX_ERROR):
    """float32 scoring of one 3x3 model (row-major 9-vector) against pts (M, 4) float32 -> bool (M,)."""
    f = np.float32
    m = np.asarray(m9, np.float32).reshape(9)
    x1, y1, x2, y2 = (pts[:, i].astype(np.float32) for i in range(4))
    t2 = f(f(max_error) * f(max_error))
    with np.errstate(all="ignore"):
        if model == "F":
            fx0 = m[0] * x1 + m[1] * y1 + m[2]
            fx1 = m[3] * x1 + m[4] * y1 + m[5]
            fx2 = m[6] * x1 + m[7] * y1 + m[8]
            ft0 = m[0] * x2 + m[3] * y2 + m[6]
            ft1 = m[1] * x2 + m[4] * y2 + m[7]
            c = x2 * fx0 + y2 * fx1 + fx2
            den = fx0 * fx0 + fx1 * fx1 + ft0 * ft0 + ft1 * ft1
            return c * c <= t2 * den
        p0 = m[0] * x1 + m[1] * y1 + m[2]
        p1 = m[3] * x1 + m[4] * y1 + m[5]
        pw = m[6] * x1 + m[7] * y1 + m[8]
        dx = p0 - x2 * pw
        dy = p1 - y2 * pw
        return (pw != 0) & (dx * dx + dy * dy <= t2 * (pw * pw))


def hypotheses(model, pts, seed, n_hyp):
    """-> float32 (n_hyp, 9) hypotheses of one pair (NaN rows for void samples), and the normalisations."""
    S, salt, rows = (8, SALT_F, rows_f) if model == "F" else (4, SALT_H, rows_h)
    p64 = pts.astype(np.float64)
    T1, T2 = normalisation(p64[:, :2]), normalisation(p64[:, 2:])
    n1 = p64[:, :2] * T1[0, 0] + T1[:2, 2]
    n2 = p64[:, 2:] * T2[0, 0] + T2[:2, 2]
    idx = sample_indices(seed, n_hyp, S, len(pts), salt)
    out = np.full((n_hyp, 9), np.nan, np.float32)
    for k in range(n_hyp):
        if idx[k, 0] < 0:
            continue
        s = idx[k]
        A, b = rows(n1[s, 0], n1[s, 1], n2[s, 0], n2[s, 1])
        sol = _solve(A, b)
        if np.all(np.isfinite(sol)):
            out[k] = denormalise(model, to_matrix(sol), T1, T2).reshape(9).astype(np.float32)
    return out, (T1, T2, n1, n2)


def refit(model, mask, norm):
    T1, T2, n1, n2 = norm
    rows = rows_f if model == "F" else rows_h
    A, b = rows(n1[mask, 0], n1[mask, 1], n2[mask, 0], n2[mask, 1])
    sol = _solve(A.T @ A, A.T @ b)
    if not np.all(np.isfinite(sol)):
        return None
    return denormalise(model, to_matrix(sol), T1, T2).reshape(9).astype(np.float32)


def estimate_model(model, pts, seed, n_hyp):
    """-> (model9 float32 or None, inlier mask bool (M,))."""
    hyp, norm = hypotheses(model, pts, seed, n_hyp)
    counts = np.array([int(inliers_f32(model, h, pts).sum()) for h in hyp])
    k = int(np.argmax(counts))                       # first maximum: lowest k on ties
    if counts[k] == 0:
        return None, np.zeros(len(pts), bool)
    best, mask = hyp[k], inliers_f32(model, hyp[k], pts)
    if mask.sum() >= (8 if model == "F" else 4):
        r = refit(model, mask, norm)
        if r is not None:
            rmask = inliers_f32(model, r, pts)
            if rmask.sum() >= mask.sum():
                best, mask = r, rmask
    return best, mask


def stored_f(f9):
    F = np.asarray(f9, np.float64).reshape(3, 3)
    U, s, Vt = np.linalg.svd(F)
    F2 = U @ np.diag([s[0], s[1], 0.0]) @ Vt
    n = np.linalg.norm(F2)
    return F2 / n if n > 0 else F2


def verify_pair(kp1, kp2, matches, pair_id, num_f=NUM_HYP_F, num_h=NUM_HYP_H):
    """kp (N, >=2) float32 keypoints, matches (M, 2) uint32 -> dict(config, inlier_matches, F, H, n_f, n_h)."""
    matches = np.asarray(matches, np.uint32).reshape(-1, 2)
    res = dict(config=CONFIG_DEGENERATE, inlier_matches=np.zeros((0, 2), np.uint32), F=np.zeros((3, 3)), H=np.zeros((3, 3)),
               n_f=0, n_h=0)
    if len(matches) < MIN_NUM_INLIERS:
        return res
    pts = np.concatenate([kp1[matches[:, 0], :2], kp2[matches[:, 1], :2]], axis=1).astype(np.float32)
    seed = int(pair_id) & 0xFFFFFFFF
    f9, fmask = estimate_model("F", pts, seed, num_f)
    h9, hmask = estimate_model("H", pts, seed, num_h)
    res["n_f"], res["n_h"] = int(fmask.sum()), int(hmask.sum())
    if res["n_f"] < max(MIN_NUM_INLIERS, MIN_INLIER_RATIO * len(matches)):
        return res
    res["F"] = stored_f(f9)
    if h9 is not None:
        H = np.asarray(h9, np.float64).reshape(3, 3)
        res["H"] = H / H[2, 2] if H[2, 2] != 0 else H
    if res["n_h"] / res["n_f"] > MAX_H_INLIER_RATIO:
        res["config"] = CONFIG_PLANAR_OR_PANORAMIC
        mask = hmask if res["n_h"] > res["n_f"] else fmask
    else:
        res["config"] = CONFIG_UNCALIBRATED
        mask = fmask
    res["inlier_matches"] = matches[mask]
    return res


def synthetic_two_view(seed, n_points=300, outlier_frac=0.3, planar=False, width=640, height=480, noise=0.5):
    """A seeded two-view scene for tests: 3-D points seen by two pinhole cameras (or a plane: every point on z = 6),
    pixel noise, and a fraction of wrong matches.  -> kp1 (N, 2), kp2 (N, 2) float32, matches (N, 2) uint32, is_inlier."""
    rs = np.random.RandomState(seed)
    K = np.array([[600.0, 0, width / 2], [0, 600.0, height / 2], [0, 0, 1]])
    X = np.stack([rs.uniform(-3, 3, n_points), rs.uniform(-2, 2, n_points),
                  np.full(n_points, 6.0) if planar else rs.uniform(4, 9, n_points)], axis=1)
    a = 0.12
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([-0.8, 0.05, 0.1])
    p1 = (K @ X.T).T
    p2 = (K @ (R @ X.T + t[:, None])).T
    kp1 = (p1[:, :2] / p1[:, 2:]) + rs.normal(0, noise, (n_points, 2))
    kp2 = (p2[:, :2] / p2[:, 2:]) + rs.normal(0, noise, (n_points, 2))
    is_in = rs.uniform(size=n_points) >= outlier_frac
    perm = rs.permutation(n_points)
    j = np.where(is_in, np.arange(n_points), perm)
    is_in &= j == np.arange(n_points)
    matches = np.stack([np.arange(n_points), j], axis=1).astype(np.uint32)
    return kp1.astype(np.float32), kp2.astype(np.float32), matches, is_in
