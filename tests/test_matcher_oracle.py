"""Known-answer tests that pin the matcher specification (the reference's
matcher is third-party and unobservable here -> parity unpinned; these KATs ARE the spec),
plus agreement between the three oracle restatements (scan / numpy / C)."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import matcher_oracle as mo


def unit_rows(rs, n, d):
    """Rows normalised to length 512 (as SIFT / the ViT quantiser produce), uint8."""
    x = np.abs(rs.standard_normal((n, d))).astype(np.float32)
    x /= np.sqrt((x * x).sum(axis=1, keepdims=True))
    return np.clip(x * 512.0, 0, 255).astype(np.uint8)


def correlated_sets(seed, n, d, noise=0.15, shuffle=True):
    """Two views of one descriptor pool: realistic input where many rows DO match."""
    rs = np.random.RandomState(seed)
    # non-negative (SIFT-like) so rows keep length ~512 after the uint8 clip; signed ViT-style
    # rows lose half their energy to the clip and fall outside max_distance = 0.7
    base = np.abs(rs.standard_normal((n, d))).astype(np.float32)
    a = np.abs(base + noise * rs.standard_normal((n, d)).astype(np.float32))
    b = np.abs(base + noise * rs.standard_normal((n, d)).astype(np.float32))
    perm = rs.permutation(n) if shuffle else np.arange(n)
    b = b[perm]

    def q(x):
        x = x / np.sqrt((x * x).sum(axis=1, keepdims=True))
        return np.clip(x * 512.0, 0, 255).astype(np.uint8)

    return q(a), q(b), perm


ALL = [lambda a, b, **k: mo.match_pair(a, b, scan=True, **k),
       lambda a, b, **k: mo.match_pair(a, b, **k),
       lambda a, b, **k: c_oracle.match_pair(a, b, **k)]
NAMES = ["scan", "numpy", "c"]


@pytest.mark.parametrize("match", ALL, ids=NAMES)
def test_kat_identity(match):
    """(i) identical, well-separated 512-normalised sets -> identity matches."""
    d = np.zeros((6, 128), np.uint8)
    for i in range(6):
        d[i, i * 16:(i + 1) * 16] = 128  # orthogonal rows, |row| = 128*4 = 512
    m = match(d, d)
    assert m.dtype == np.uint32
    assert np.array_equal(m, np.stack([np.arange(6), np.arange(6)], 1))


@pytest.mark.parametrize("match", ALL, ids=NAMES)
def test_kat_equal_best_rejected(match):
    """(ii) two equal best columns -> theta_best == theta_second -> ratio test rejects."""
    d1 = np.zeros((1, 128), np.uint8)
    d1[0, :16] = 128
    d2 = np.zeros((3, 128), np.uint8)
    d2[0, :16] = 128
    d2[1, :16] = 128
    d2[2, 16:32] = 128
    assert len(match(d1, d2, cross_check=False)) == 0
    idx, best, second = mo.top2(mo.similarity(d1, d2))
    assert idx[0] == 0 and best[0] == second[0] == 128 * 128 * 16  # lowest index wins the tie


@pytest.mark.parametrize("match", ALL, ids=NAMES)
def test_kat_cross_check(match):
    """(iii) asymmetric nearest neighbour is removed by the cross check only."""
    def blocks(*w):
        v = np.zeros(64, np.uint8)
        for k, x in enumerate(w):
            v[16 * k:16 * (k + 1)] = x
        return v
    d1 = np.stack([blocks(128), blocks(115, 55)])               # both closest to d2[0]
    d2 = np.stack([blocks(128), blocks(0, 0, 128)])
    one = match(d1, d2, cross_check=False)
    both = match(d1, d2, cross_check=True)
    assert np.array_equal(one, [[0, 0], [1, 0]])
    assert np.array_equal(both, [[0, 0]])


@pytest.mark.parametrize("match", ALL, ids=NAMES)
def test_kat_max_distance_boundary(match):
    """(iv) theta_best just below / above max_distance = 0.7 (cos 0.7 = 0.76484...)."""
    d2 = np.zeros((2, 64), np.uint8)
    d2[0, 0] = 255
    d2[0, 1] = 255
    d2[1, 5] = 255
    def row(s_target):
        # one query whose dot product with d2[0] is s_target*? : pick integer entries a,b
        r = np.zeros((1, 64), np.uint8)
        r[0, 0], r[0, 1] = s_target
        return r
    # s = 255*(a+b); s/2^18 vs cos(0.7)=0.764842 -> threshold s = 200498.9
    hi = row((255, 255))   # s = 130050 -> x = 0.496 -> theta = 1.05 > 0.7 : reject
    assert len(match(hi, d2, cross_check=False)) == 0
    d2b = d2.copy(); d2b[0, 2] = 255; d2b[0, 3] = 255
    q = np.zeros((1, 64), np.uint8); q[0, :4] = (255, 255, 255, 22)   # s = 255*787 = 200685
    assert mo.theta_f32(200685) < np.float32(0.7)
    assert np.array_equal(match(q, d2b, cross_check=False), [[0, 0]])
    q[0, 3] = 21                                                       # s = 200430 -> theta > 0.7
    assert mo.theta_f32(200430) > np.float32(0.7)
    assert len(match(q, d2b, cross_check=False)) == 0


@pytest.mark.parametrize("match", ALL, ids=NAMES)
def test_kat_dummy_descriptors_saturate(match):
    """(v) un-normalised Dummy descriptors: s >> 512^2 -> theta = 0 for best and second ->
    0 >= 0.8*0 -> every match rejected (SURVEY.md §8 a-M)."""
    from oracle.select_oracle import dummy_features
    _, desc = dummy_features(480, 640)
    assert len(match(desc, desc)) == 0


@pytest.mark.parametrize("match", ALL, ids=NAMES)
def test_kat_zero_similarity_never_matches(match):
    d1 = np.zeros((2, 32), np.uint8); d1[0, 0] = 200
    d2 = np.zeros((2, 32), np.uint8); d2[0, 1] = 200
    assert len(match(d1, d2, cross_check=False)) == 0


@pytest.mark.parametrize("match", ALL, ids=NAMES)
def test_empty_inputs(match):
    a = np.zeros((0, 128), np.uint8)
    b = unit_rows(np.random.RandomState(0), 5, 128)
    assert match(a, b).shape == (0, 2)
    assert match(b, a).shape == (0, 2)


@pytest.mark.parametrize("seed,n1,n2,d", [(0, 40, 37, 128), (1, 64, 64, 384), (2, 33, 130, 256), (3, 100, 7, 64)])
def test_restatements_agree(seed, n1, n2, d):
    rs = np.random.RandomState(seed)
    n = max(n1, n2)
    a, b, _ = correlated_sets(seed, n, d)
    a, b = a[:n1], b[:n2]
    a[rs.randint(n1)] = a[0]                      # duplicate rows -> ties
    S = mo.similarity(a, b)
    for x, y in zip(mo.top2_scan(S), mo.top2(S)):
        assert np.array_equal(x, y)
    r = c_oracle.top2_both(a, b)
    for x, y in zip(mo.top2(S), r[:3]):
        assert np.array_equal(x, y)
    for x, y in zip(mo.top2(np.ascontiguousarray(S.T)), r[3:]):
        assert np.array_equal(x, y)
    for cc in (True, False):
        m0 = mo.match_pair(a, b, cross_check=cc, scan=True)
        assert np.array_equal(m0, mo.match_pair(a, b, cross_check=cc))
        assert np.array_equal(m0, c_oracle.match_pair(a, b, cross_check=cc))


def test_correlated_sets_recover_permutation():
    a, b, perm = correlated_sets(5, 200, 128, noise=0.1)
    m = mo.match_pair(a, b)
    assert len(m) > 150
    inv = np.argsort(perm)
    assert (inv[m[:, 0]] == m[:, 1]).mean() > 0.99
    assert np.all(np.diff(m[:, 0].astype(int)) > 0)           # ordered by i


def test_theta_c_equals_numpy_everywhere():
    s = np.arange(0, mo.S_SAT + 1, 97)
    t = mo.theta_f32(s)
    tc = np.array([c_oracle.theta(int(v)) for v in s], np.float32)
    assert np.array_equal(t, tc)
    assert mo.theta_f32(mo.S_SAT) == 0 and mo.theta_f32(10**8) == 0


def test_batch_api_and_pair_order():
    rs = np.random.RandomState(9)
    n_img, n_max, d = 5, 48, 128
    desc = np.zeros((n_img, n_max, d), np.uint8)
    counts = np.array([48, 40, 0, 17, 48], np.int32)
    pool, _, _ = correlated_sets(9, n_max, d)
    for k in range(n_img):
        noisy = pool.astype(np.int32) + rs.randint(-6, 7, pool.shape)
        desc[k, :counts[k]] = np.clip(noisy, 0, 255).astype(np.uint8)[:counts[k]]
    pairs = mo.exhaustive_pairs(n_img)
    assert pairs.tolist()[:4] == [[0, 1], [0, 2], [0, 3], [0, 4]] and len(pairs) == 10
    out, cnt, used = c_oracle.match_pairs(desc, counts, pairs)
    assert used >= 1
    for p, (a, b) in enumerate(pairs):
        ref = mo.match_pair(desc[a, :counts[a]], desc[b, :counts[b]])
        assert cnt[p] == len(ref)
        assert np.array_equal(out[p, :cnt[p]], ref)
    assert mo.pair_id(1, 2) == 2147483647 + 2 and mo.pair_id(3, 1) == 2147483647 + 3
