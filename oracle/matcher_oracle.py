"""CPU ORACLE (test infrastructure, NOT product code) — exhaustive descriptor matcher.

What it restates
----------------
The reference does not contain a matcher: it calls `pycolmap.match_exhaustive`
(`/root/reference/vit_colmap/pipeline/run_pipeline.py:351-363`, options built at
`/root/reference/vit_colmap/utils/config.py:64-96`: max_ratio 0.8, max_distance 0.7,
cross_check True).  The arithmetic lives in third-party COLMAP C++ (`pycolmap==3.12.6`,
`/root/reference/uv.lock:773-774`), which is neither vendored (`third_party/colmap` is an empty
directory) nor installed here.  This file restates COLMAP's published CPU brute-force SIFT
matcher semantics from memory (SURVEY.md §8 a-M, marked [recalled]):

    dists(i, j)  = sum_k int(d1[i,k]) * int(d2[j,k])              (int32, uint8 descriptors)
    one way      : per row, scan columns in ascending order keeping best / second-best
                   similarity with strict '>' (best starts at 0, so a zero similarity never
                   matches and the lowest column index wins ties)
    angle        : theta = acos(min(dist / 512^2, 1))              (float32)
    reject       : theta_best > max_distance, or theta_best >= max_ratio * theta_second
    cross check  : the same on the transpose; keep (i, j) iff m12[i] == j and m21[j] == i
    output       : uint32 (M, 2), ordered by i ascending

PARITY UNPINNED for match *contents*: the reference's only test at this boundary asserts
`num_matched_image_pairs >= 1` (`/root/reference/tests/test_smoke_e2e.py:61-65`), and no
golden vectors exist.  The spec above is pinned by the hand-built known-answer tests in
`tests/test_matcher_oracle.py` and by agreement between the three restatements here
(sequential scan, vectorised numpy, C in `matcher_oracle.c`).

One deliberate, documented choice: `theta` is defined as the float64 `acos` rounded to
float32 (correctly rounded `acosf`), not "whatever the platform's `acosf` returns", so the
HIP kernel and this oracle can agree bit-for-bit on every one of the 262 145 possible inputs
(`tests/test_matcher_gpu.py::test_theta_table_exhaustive`).  It differs from glibc's `acosf`
by at most 1 ulp on a minority of inputs.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.
"""
import numpy as np

K_NORM = np.float32(1.0 / (512.0 * 512.0))  # exactly 2^-18
S_SAT = 512 * 512  # similarity at which min(s / 512^2, 1) saturates


def similarity(d1: np.ndarray, d2: np.ndarray) -> np.ndarray:
    """int32 dot products between every row of d1 (n1, D) and d2 (n2, D), uint8 inputs."""
    return d1.astype(np.int32) @ d2.astype(np.int32).T


def theta_f32(s) -> np.ndarray:
    """theta(s) = RN_f32(acos_f64(min(f32(s) * 2^-18, 1))) for integer similarity s >= 0."""
    x = np.minimum(np.asarray(s).astype(np.float32) * K_NORM, np.float32(1.0))
    return np.arccos(x.astype(np.float64)).astype(np.float32)


def accept(best: np.ndarray, second: np.ndarray, max_ratio: float, max_distance: float):
    """Distance + ratio test on integer (best, second) similarities -> bool mask."""
    tb = theta_f32(best)
    ts = theta_f32(second)
    ok = best > 0
    ok &= ~(tb > np.float32(max_distance))
    ok &= ~(tb >= np.float32(max_ratio) * ts)
    return ok


def top2_scan(S: np.ndarray):
    """Sequential restatement (pure Python; small cases only): per row best idx / best / second."""
    n1, n2 = S.shape
    idx = np.full(n1, -1, np.int32)
    best = np.zeros(n1, np.int32)
    second = np.zeros(n1, np.int32)
    for i in range(n1):
        b_i, b, s2 = -1, 0, 0
        for j in range(n2):
            d = int(S[i, j])
            if d > b:
                b_i, s2, b = j, b, d
            elif d > s2:
                s2 = d
        idx[i], best[i], second[i] = b_i, b, s2
    return idx, best, second


def top2(S: np.ndarray):
    """Vectorised equivalent of `top2_scan` (first occurrence of the row maximum wins)."""
    n1, n2 = S.shape
    if n2 == 0:
        z = np.zeros(n1, np.int32)
        return np.full(n1, -1, np.int32), z, z.copy()
    idx = np.argmax(S, axis=1).astype(np.int32)
    best = S[np.arange(n1), idx].astype(np.int32)
    if n2 > 1:
        masked = S.copy()
        masked[np.arange(n1), idx] = -1
        second = masked.max(axis=1).astype(np.int32)
    else:
        second = np.zeros(n1, np.int32)
    second = np.maximum(second, 0)
    none = best <= 0
    idx[none] = -1
    best[none] = 0
    second[none] = 0
    return idx, best, second


def one_way(S, max_ratio, max_distance, scan=False):
    idx, best, second = (top2_scan if scan else top2)(S)
    m = np.where(accept(best, second, max_ratio, max_distance), idx, -1).astype(np.int32)
    return m


def match_pair(d1: np.ndarray, d2: np.ndarray, max_ratio: float = 0.8,
               max_distance: float = 0.7, cross_check: bool = True, scan: bool = False):
    """uint8 (n1, D), (n2, D) -> uint32 (M, 2) matches ordered by i."""
    if len(d1) == 0 or len(d2) == 0:
        return np.zeros((0, 2), np.uint32)
    S = similarity(d1, d2)
    m12 = one_way(S, max_ratio, max_distance, scan)
    if cross_check:
        m21 = one_way(np.ascontiguousarray(S.T), max_ratio, max_distance, scan)
        i = np.nonzero(m12 >= 0)[0]
        keep = m21[m12[i]] == i
        i = i[keep]
    else:
        i = np.nonzero(m12 >= 0)[0]
    return np.stack([i, m12[i]], axis=1).astype(np.uint32)


def exhaustive_pairs(n_images: int) -> np.ndarray:
    """All unordered pairs (a < b) in row-major order: (0,1), (0,2) ... (n-2, n-1)."""
    a, b = np.triu_indices(n_images, k=1)
    return np.stack([a, b], axis=1).astype(np.int32)


def pair_id(image_id1: int, image_id2: int) -> int:
    """COLMAP pair id [recalled]: id1 * 2147483647 + id2 with id1 < id2 (1-based image ids)."""
    if image_id1 > image_id2:
        image_id1, image_id2 = image_id2, image_id1
    return image_id1 * 2147483647 + image_id2
