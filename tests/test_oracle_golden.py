"""Pin the selection oracle against vectors produced by the reference's own functions
(tests/golden/make_golden.py).  Integer stages are compared exactly on the golden score map;
float stages to float32 summation-order noise."""
import hashlib
import os

import numpy as np
import pytest

from cases import CASES, make_feature_map, make_projection
from oracle import select_oracle as so

GOLD = os.path.join(os.path.dirname(__file__), "golden")
IDS = [c["name"] for c in CASES]


def load(case):
    return np.load(os.path.join(GOLD, f"select_{case['name']}.npz"))


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_score_maps(case):
    g = load(case)
    f = make_feature_map(case)
    np.testing.assert_allclose(so.harris_response(f), g["harris"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(so.dog_response(f), g["dog"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(so.distinctiveness(f, "combined"), g["combined"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(so.distinctiveness(f, case["method"]), g["score"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_binning_topk_nms_exact(case):
    g = load(case)
    coords, scores = so.spatial_binning_selection(g["score"], case["num_keypoints"], 16)
    assert np.array_equal(coords, g["bin_coords"])
    assert np.array_equal(scores, g["bin_scores"])
    tk_c, tk_s = so.simple_topk_selection(g["score"], min(case["num_keypoints"], 64))
    assert np.array_equal(tk_c, g["topk_coords"])
    assert np.array_equal(tk_s, g["topk_scores"])
    kept, kept_s = so.apply_nms(g["bin_coords"], g["bin_scores"], 1.5)
    assert np.array_equal(kept, g["nms_coords"])
    assert np.array_equal(kept_s, g["nms_scores"])


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_descriptors_and_keypoints(case):
    g = load(case)
    f = make_feature_map(case)
    H, W = case["H"], case["W"]
    kept = g["nms_coords"]
    desc = so.gather_descriptors(f, kept)
    n = len(g["desc_gather_head"])
    np.testing.assert_allclose(desc[:n], g["desc_gather_head"], rtol=1e-5, atol=1e-6)
    kp = so.map_keypoints(kept, (H, W), (W * 14, H * 14), case["orig_wh"])
    assert np.array_equal(kp, g["keypoints"])
    if case["C"] > case["descriptor_dim"]:
        desc = so.project(desc, make_projection(case))
    d = so.l2_normalize(desc)
    np.testing.assert_allclose(d[:n], g["desc_f32_head"], rtol=1e-4, atol=1e-6)
    # quantiser is exact on identical float input
    assert np.array_equal(so.quantize_u8(g["desc_f32_head"]), g["desc_u8"][:n])
    # end to end uint8: at most 1 LSB away, on a small fraction of entries (truncation flips)
    q = so.quantize_u8(d)
    diff = np.abs(q.astype(np.int32) - g["desc_u8"].astype(np.int32))
    assert diff.max() <= 1
    assert (diff != 0).mean() < 2e-3
    sha = hashlib.sha256(np.ascontiguousarray(g["desc_u8"]).tobytes()).digest()
    assert np.array_equal(np.frombuffer(sha, np.uint8), g["desc_u8_sha256"])


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_dense_to_sparse_end_to_end(case):
    g = load(case)
    f = make_feature_map(case)
    H, W = case["H"], case["W"]
    proj = make_projection(case) if case["C"] > case["descriptor_dim"] else None
    out = so.dense_to_sparse(f, case["orig_wh"], (W * 14, H * 14), case["num_keypoints"],
                             case["descriptor_dim"], case["method"], proj, score=g["score"])
    assert np.array_equal(out["coords"], g["nms_coords"])
    assert np.array_equal(out["keypoints"], g["keypoints"])
    assert out["desc_u8"].shape == g["desc_u8"].shape
    assert np.abs(out["desc_u8"].astype(int) - g["desc_u8"].astype(int)).max() <= 1


def test_dummy_features_match_reference():
    g = np.load(os.path.join(GOLD, "dummy_640x480.npz"))
    kp, desc = so.dummy_features(480, 640, step=32, seed=42)
    assert kp.dtype == np.float32 and desc.dtype == np.uint8
    assert np.array_equal(kp, g["keypoints"])
    assert np.array_equal(desc, g["descriptors"])
    assert so.default_camera_params("PINHOLE", 640, 480) == list(g["camera_params"])


def test_binning_margin_cells_are_never_candidates():
    """at 34x45 only y<32, x<32 are covered by the 2x2 bins."""
    rs = np.random.RandomState(0)
    score = rs.rand(34, 45).astype(np.float32)
    score[33, 44] = 5.0
    coords, _ = so.spatial_binning_selection(score, 512, 16)
    assert coords[:, 0].max() < 32 and coords[:, 1].max() < 32
    assert len(coords) == 512
