"""GPU parity: HIP matcher (through the C ABI) vs the CPU oracle — bit-exact."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import matcher_oracle as mo
from util_data import image_set

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


SENTINEL = -77777   # out_counts is pre-filled with it: a count the kernel never wrote cannot pass for a result


def run_batch(desc, counts, pairs, **kw):
    from vit_colmap_amd.matching import match_pairs, prepare_descriptors

    n_images, n_max, d = desc.shape
    prepared = prepare_descriptors(dev(desc), dev(counts))
    out_counts = torch.full((len(pairs),), SENTINEL, dtype=torch.int32, device="cuda")
    m, c = match_pairs(prepared, dev(counts), n_images, n_max, d, dev(pairs), out_counts=out_counts, **kw)
    torch.cuda.synchronize()
    return m.cpu().numpy().view(np.uint32), c.cpu().numpy()


def assert_batch_equal(desc, counts, pairs, **kw):
    gm, gc = run_batch(desc, counts, pairs, **kw)
    assert not np.any(gc == SENTINEL), f"counts never written for pairs {np.nonzero(gc == SENTINEL)[0][:10]}"
    om, oc, _ = c_oracle.match_pairs(desc, counts, pairs, **kw)
    assert np.array_equal(gc, oc), f"match counts differ: {np.nonzero(gc != oc)[0][:10]}"
    for p in range(len(pairs)):
        assert np.array_equal(gm[p, :gc[p]], om[p, :oc[p]]), f"pair {p} {pairs[p]}"
    return gc


def test_theta_table_exhaustive():
    """Every possible angle input: GPU == oracle, so accept() agrees for all integer inputs."""
    from vit_colmap_amd.matching import theta_table

    n = mo.S_SAT + 4
    g = theta_table(n).cpu().numpy()                       # the build-time table the pair kernel reads (clamped)
    e = theta_table(n, evaluate=True).cpu().numpy()        # the same expression evaluated on the device
    o = mo.theta_f32(np.arange(n))
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    assert np.array_equal(e.view(np.uint32), o.view(np.uint32))


@pytest.mark.parametrize("n1,n2,d", [(512, 512, 384), (300, 300, 128), (33, 65, 128), (1, 7, 64),
                                     (257, 31, 256), (640, 1000, 96), (2048, 2048, 256), (100, 50, 768),
                                     (64, 64, 1024), (45, 77, 40)])
@pytest.mark.parametrize("kind", ["scene", "full"])
def test_knn_top2_matches_oracle(n1, n2, d, kind):
    from vit_colmap_amd.matching import knn_top2

    desc, _ = image_set(n1 * 7 + n2 + d, 2, max(n1, n2), d, kind=kind)
    a, b = desc[0, :n1].copy(), desc[1, :n2].copy()
    a[n1 // 2] = a[0]                                  # duplicate row
    if n2 > 3:
        b[n2 - 1] = b[1]                               # duplicate column -> tie on the best value
    idx, best, second = [t.cpu().numpy() for t in knn_top2(dev(a), dev(b))]
    o = c_oracle.top2_both(a, b)
    assert np.array_equal(best, o[1])
    assert np.array_equal(second, o[2])
    assert np.array_equal(idx, o[0])


def test_knn_top2_then_mutual_ratio_equals_oracle():
    from vit_colmap_amd.matching import knn_top2, mutual_ratio

    desc, _ = image_set(11, 2, 400, 128, kind="scene", noise=0.1)
    a, b = desc[0, :400].copy(), desc[1, :333].copy()
    r12 = knn_top2(dev(a), dev(b))
    r21 = knn_top2(dev(b), dev(a))
    for cc in (True, False):
        out, cnt = mutual_ratio(r12, r21, 400, 333, cross_check=cc)
        n = int(cnt.item())
        ref = mo.match_pair(a, b, cross_check=cc)
        assert n == len(ref) and n > 20
        assert np.array_equal(out[:n].cpu().numpy().view(np.uint32), ref)


@pytest.mark.parametrize("kind,d", [("scene", 384), ("vit", 384), ("full", 128), ("scene", 256)])
def test_match_pairs_batch_exact(kind, d):
    n_images, n_max = 10, 512
    desc, counts = image_set(5, n_images, n_max, d, kind=kind)
    pairs = mo.exhaustive_pairs(n_images)
    gc = assert_batch_equal(desc, counts, pairs)
    if kind == "scene":
        assert gc.sum() > 1000            # the accept path is really exercised
    if kind == "full":
        assert gc.sum() == 0              # saturated similarities are all rejected


@pytest.mark.parametrize("d", [384, 256, 200])
def test_match_pairs_bright_rows_packed_sums(d):
    """Arbitrary uint8 input the C ABI accepts: rows whose bytes beyond the first 128 dimensions sum to more than
    2^15 (the packed head / tail row sums gave the tail 15 bits) against sparse rows, so that the
    similarities stay below saturation and planted matches pass the angle tests."""
    rs = np.random.RandomState(d)
    n = 96
    bright = rs.randint(150, 200, (n, d)).astype(np.uint8)
    special = np.stack([rs.permutation(d)[:4] for _ in range(n)])      # 4 dimensions per row that hold 255
    for i in range(n):
        bright[i, special[i]] = 255
    if d == 384:   # the tail of the packed word: dimensions 128.. (KS = 12, head of 4 k-steps)
        assert bright[:, 128:].astype(np.int64).sum(axis=1).min() >= 1 << 15
    sparse = np.zeros((n, d), np.uint8)
    perm = rs.permutation(n)
    for j in range(n):
        sparse[j, special[perm[j]]] = 250                              # s = 255000 with its partner, ~175000 otherwise
    desc = np.ascontiguousarray(np.stack([bright, sparse, bright[::-1], sparse[perm]]))
    counts = np.full(4, n, np.int32)
    pairs = np.array([[0, 1], [1, 0], [2, 1], [1, 2], [0, 3], [0, 2], [1, 3]], np.int32)
    gc = assert_batch_equal(desc, counts, pairs)
    assert gc[0] == n and gc[1] == n and gc[4] == n                    # every planted partner is found
    assert_batch_equal(desc, counts, pairs, max_ratio=0.95, max_distance=1.2)


def test_match_pairs_ragged_counts_and_empty_images():
    n_images, n_max, d = 9, 300, 128
    counts = np.array([300, 0, 1, 31, 32, 33, 257, 299, 64], np.int32)
    desc, counts = image_set(6, n_images, n_max, d, kind="scene", counts=counts, noise=0.1)
    pairs = mo.exhaustive_pairs(n_images)
    gc = assert_batch_equal(desc, counts, pairs)
    assert gc.sum() > 100
    assert_batch_equal(desc, counts, pairs, cross_check=False)
    assert_batch_equal(desc, counts, pairs, max_ratio=0.95, max_distance=1.2)


def test_match_pairs_max_size_and_order_of_pairs():
    n_images, n_max, d = 3, 2048, 256
    desc, counts = image_set(7, n_images, n_max, d, kind="scene", noise=0.15)
    pairs = np.array([[2, 0], [0, 1], [1, 1], [1, 2]], np.int32)   # arbitrary order, self pair
    assert_batch_equal(desc, counts, pairs)


def test_identical_images_give_identity_matches():
    desc, counts = image_set(8, 1, 512, 384, kind="scene")
    desc = np.concatenate([desc, desc], axis=0)
    counts = np.concatenate([counts, counts])
    gm, gc = run_batch(desc, counts, np.array([[0, 1]], np.int32))
    m = gm[0, :gc[0]]
    assert np.array_equal(m[:, 0], m[:, 1]) and gc[0] > 400


def test_full_size_checksum_properties_c3():
    """BASELINE config 3 size (50 images, 512 x 384, 1225 pairs): size-independent properties
    plus an exact comparison of every pair (the C oracle does all 1225 in seconds)."""
    n_images, n_max, d = 50, 512, 384
    desc, counts = image_set(9, n_images, n_max, d, kind="scene")
    pairs = mo.exhaustive_pairs(n_images)
    gm, gc = run_batch(desc, counts, pairs)
    # symmetry: matching (b, a) gives the transposed match list
    gm2, gc2 = run_batch(desc, counts, np.ascontiguousarray(pairs[:, ::-1]))
    for p in range(0, len(pairs), 37):
        m = gm[p, :gc[p]]
        mt = gm2[p, :gc2[p]]
        assert gc[p] == gc2[p]
        assert np.array_equal(m[np.argsort(m[:, 1], kind="stable")][:, ::-1], mt)
        assert np.all(np.diff(m[:, 0].astype(np.int64)) > 0)          # sorted, no duplicate rows
        assert len(np.unique(m[:, 1])) == len(m)                      # one-to-one
    om, oc, _ = c_oracle.match_pairs(desc, counts, pairs)
    assert np.array_equal(gc, oc)
    for p in range(len(pairs)):
        assert np.array_equal(gm[p, :gc[p]], om[p, :oc[p]])


# ---- the persistent kernel's own structure: ranges of pairs per workgroup, A reuse, the ring across pairs ----------
@pytest.mark.parametrize("n_images,n_max,d,kind", [(40, 96, 128, "scene"), (27, 160, 256, "scene"), (33, 64, 64, "vit"),
                                                   (24, 512, 384, "scene")])
def test_persistent_ranges_more_pairs_than_cus(n_images, n_max, d, kind):
    """More pairs than CUs (every workgroup walks several pairs, reusing image a), ragged and EMPTY images in
    between, images shorter than the LDS ring (the producer runs into the next pair and has to idle)."""
    rs = np.random.RandomState(n_images)
    counts = rs.randint(1, n_max + 1, n_images).astype(np.int32)
    counts[[3, 4, n_images - 1]] = 0                      # empty images: runs of pairs without work
    counts[[5, 6]] = [1, 32]                              # one column tile: shorter than the ring
    counts[7] = n_max
    desc, counts = image_set(100 + n_images, n_images, n_max, d, kind=kind, counts=counts, noise=0.1)
    pairs = mo.exhaustive_pairs(n_images)
    gc = assert_batch_equal(desc, counts, pairs)
    if kind == "scene":
        assert gc.sum() > 500
    # a pair list that is NOT sorted by image a, with repeats and self pairs (ranges then reload A more often)
    perm = rs.permutation(len(pairs))[: 300]
    shuffled = np.ascontiguousarray(np.concatenate([pairs[perm], pairs[perm[:7], ::-1], [[7, 7]]]).astype(np.int32))
    assert_batch_equal(desc, counts, shuffled)


@pytest.mark.parametrize("n_images,n_max,d", [(40, 512, 384), (36, 300, 256), (44, 200, 128), (30, 700, 64)])
def test_sparse_pairs_with_planted_matches(n_images, n_max, d):
    """Mostly non-matching descriptors (every tile cut short by the early-out, two column tiles per barrier) with
    descriptors planted in some images: those pairs fail the head test on a few tiles (the restart-free continuation),
    produce matches, and switch their workgroup between the one- and two-tile modes from pair to pair."""
    rs = np.random.RandomState(7 * n_images + d)
    desc, counts = image_set(300 + n_images, n_images, n_max, d, kind="vit")
    counts = counts.copy()
    counts[rs.permutation(n_images)[:6]] = rs.randint(1, n_max, 6)          # a few ragged images
    donors = rs.permutation(n_images)[: n_images // 3]
    for k in donors:                                                        # image k shares rows with image (k + 3) % n
        j = (k + 3) % n_images
        n = int(min(counts[k], counts[j]))
        rows = rs.permutation(n)[: max(n // 8, 1)]
        desc[j, rows] = desc[k, rs.permutation(n)[: len(rows)]]
    desc = np.ascontiguousarray(desc)
    pairs = mo.exhaustive_pairs(n_images)
    # (negatives clipped by the reference's quantiser leave a row at ~0.5 similarity with ITSELF: angle 1.05, beyond the
    # default max_distance of 0.7 — planted rows only match under a wider one)
    gc = assert_batch_equal(desc, counts, pairs, max_distance=1.1)
    assert gc.sum() > 100 and (gc == 0).sum() > len(pairs) // 2            # both kinds of pairs are present
    perm = rs.permutation(len(pairs))
    assert_batch_equal(desc, counts, np.ascontiguousarray(pairs[perm]), max_distance=1.1)   # no runs of image a


@pytest.mark.parametrize("d", [384, 256, 128])
@pytest.mark.parametrize("n_long", [8 * 32, 16 * 32 - 5, 64 * 32])
def test_short_image_a_against_long_image_b(d, n_long):
    """image a with 1 / 2 / 31 rows (seven of the eight waves own no live row, and the wave that does
    fails the early-out test while its SIMD partners pass it) against images of 8 / 16 / 64 column tiles, on both
    ring modes (D = 128: two tiles per barrier; D = 256 / 384 at 2048 rows: one), counts pre-filled with a sentinel."""
    n_short = [1, 2, 31]
    n_images = 9
    counts = np.array(n_short + [n_long, n_long - 7, n_long, 1, 33, n_long], np.int32)
    desc, counts = image_set(n_long + d, n_images, n_long, d, kind="scene", counts=counts, noise=0.05)
    # every short image against every long one, in both orders, short-short pairs in between
    pairs = [(a, b) for a in range(3) for b in range(3, 9)] + [(b, a) for a in range(3) for b in (3, 5, 8)]
    pairs += [(0, 1), (0, 2), (1, 2), (6, 0), (6, 7)]
    pairs = np.array(pairs, np.int32)
    gc = assert_batch_equal(desc, counts, pairs)
    assert gc.sum() > 20
    # few pairs: every workgroup holds exactly one (the producer stays active through its only pair)
    assert_batch_equal(desc, counts, np.ascontiguousarray(pairs[:7]))
    # and "vit" rows (no relevant tile at all: two tiles per barrier from the first round on)
    desc2, _ = image_set(n_long + d + 1, n_images, n_long, d, kind="vit", counts=counts)
    assert_batch_equal(desc2, counts, pairs)


@pytest.mark.parametrize("n_pairs", [1, 2, 7, 255, 256, 257, 263])
def test_persistent_ranges_pair_counts_around_the_grid_size(n_pairs):
    n_images, n_max, d = 24, 64, 128
    desc, counts = image_set(55, n_images, n_max, d, kind="scene", noise=0.1)
    pairs = np.ascontiguousarray(mo.exhaustive_pairs(n_images)[:n_pairs])
    assert_batch_equal(desc, counts, pairs)


def test_persistent_multi_pass_pairs_share_image_a():
    """n > 512 rows: several row passes per pair (A changes every pass), three pairs in one range."""
    n_images, n_max, d = 4, 1100, 128
    counts = np.array([1100, 700, 513, 1024], np.int32)
    desc, counts = image_set(77, n_images, n_max, d, kind="scene", counts=counts, noise=0.1)
    assert_batch_equal(desc, counts, mo.exhaustive_pairs(n_images))


# ---- more rows than one kernel block holds (VC_MAX_KEYPOINTS = 2048): sub-blocks + associative top-2 merge ------------
@pytest.mark.parametrize("n1,n2,d", [(2500, 2049, 128), (4100, 1000, 64), (20480, 20480, 128)])
def test_blocked_matcher_beyond_max_keypoints(n1, n2, d):
    """The reference's trainable_vit pipeline asks for 20 480 keypoints per image (run_pipeline.py:328-333)."""
    from vit_colmap_amd.matching.exhaustive import hip_match_blocks

    n_max = max(n1, n2)
    desc, counts = image_set(n1 + d, 2, n_max, d, kind="scene", counts=[n1, n2], noise=0.1)
    desc[0, n1 // 3] = desc[0, 5]                         # duplicate rows: ties across sub-block boundaries
    desc[1, min(n2 - 1, 2100)] = desc[1, 7]
    pairs = np.array([[0, 1], [1, 0]], np.int32)
    lists = hip_match_blocks(desc, counts, pairs)
    om, oc, _ = c_oracle.match_pairs(desc, counts, pairs)
    for p in range(2):
        assert len(lists[p]) == oc[p] > 100
        assert np.array_equal(lists[p], om[p, : oc[p]])
    no_cc = hip_match_blocks(desc, counts, pairs[:1], cross_check=False)
    om2, oc2, _ = c_oracle.match_pairs(desc, counts, pairs[:1], cross_check=False)
    assert np.array_equal(no_cc[0], om2[0, : oc2[0]])
