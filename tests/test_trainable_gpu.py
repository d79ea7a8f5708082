"""GPU parity of the trainable-extractor path (csrc/heatmap.hip through the C ABI) against
oracle/trainable_oracle.py and the golden vectors produced by the reference's own `_run_inference`.
Everything is asserted bit-exact (selection, order, float32 keypoint rows incl. the score, uint8 descriptors) except the
golden score column, where torch's float32 sigmoid may differ from the correctly rounded one by 1 ulp."""
import os

import numpy as np
import pytest
import torch

from cases_trainable import CASES, make_head_outputs, map_hw
from oracle import trainable_oracle as to

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
IDS = [c["name"] for c in CASES]


def run_hip(kp_maps, d_maps, k, thr, r, orig_hw, new_hw, channels_last=False):
    from vit_colmap_amd.features import hip_select as hs

    kp = torch.from_numpy(np.ascontiguousarray(np.stack(kp_maps))).cuda()
    d = torch.from_numpy(np.ascontiguousarray(np.stack(d_maps))).cuda()
    if channels_last:
        d = d.contiguous(memory_format=torch.channels_last)
    res = hs.heatmap_keypoints(kp, d, k, thr, r, (orig_hw[1], orig_hw[0]), (new_hw[1], new_hw[0]))
    torch.cuda.synchronize()
    return res["keypoints"].cpu().numpy(), res["desc_u8"].cpu().numpy(), res["count"].cpu().numpy()


def assert_image(kps, du8, cnt, okps, odu8, kmax):
    assert cnt == len(okps)
    assert np.array_equal(kps[:cnt].view(np.uint32), okps.view(np.uint32))
    assert np.array_equal(du8[:cnt], odu8)
    assert not kps[cnt:].any() and not du8[cnt:].any()          # padded rows are zero (whole blocks for the matcher)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_kernel_matches_oracle_and_reference_golden(case):
    kp_map, d_map = make_head_outputs(case)
    new_hw, _ = map_hw(case)
    k = case["num_keypoints"]
    kps, du8, cnt = run_hip([kp_map], [d_map], k, case["score_threshold"], case["nms_radius"], case["orig_hw"], new_hw)
    okps, odu8 = to.run_inference_post(kp_map, d_map, case["orig_hw"], new_hw, k, case["score_threshold"], case["nms_radius"])
    assert_image(kps[0], du8[0], int(cnt[0]), okps, odu8, k)
    g = np.load(os.path.join(GOLD, f"trainable_{case['name']}.npz"))
    gk, gd = g["keypoints"], g["descriptors"]
    assert int(cnt[0]) == len(gk)
    if len(gk):
        cols = [0, 1, 2, 3, 5]
        assert np.array_equal(kps[0, : len(gk)][:, cols].view(np.uint32), gk[:, cols].view(np.uint32))
        assert np.all(np.abs(kps[0, : len(gk), 4].view(np.int32).astype(np.int64) - gk[:, 4].view(np.int32).astype(np.int64)) <= 1)
        assert np.array_equal(du8[0, : len(gk)], gd)


def test_batch_of_images_and_channels_last_descriptors():
    cases = [c for c in CASES if c["name"] in ("vga", "vga_k256")]
    maps = [make_head_outputs(c) for c in cases] + [make_head_outputs(dict(cases[0], seed=99))]
    new_hw, _ = map_hw(cases[0])
    for cl in (False, True):
        kps, du8, cnt = run_hip([m[0] for m in maps], [m[1] for m in maps], 700, 0.3, 3, (480, 640), new_hw, channels_last=cl)
        for i, (kp_map, d_map) in enumerate(maps):
            okps, odu8 = to.run_inference_post(kp_map, d_map, (480, 640), new_hw, 700, 0.3, 3)
            assert_image(kps[i], du8[i], int(cnt[i]), okps, odu8, 700)


def test_plateaus_and_ties_cut_in_position_order():
    """Saturated logits give score == 1.0f on whole regions: every cell of a plateau is a local maximum (equality test of
    the reference's NMS), and top-k cuts inside the tie — lowest positions first."""
    rs = np.random.RandomState(5)
    H, W, D = 40, 52, 32
    kp = rs.standard_normal((4, H, W)).astype(np.float32)
    kp[0] = -4.0 + 0.01 * rs.standard_normal((H, W)).astype(np.float32)
    kp[0, 5:12, 7:30] = 30.0                       # plateau of 161 cells at exactly 1.0
    kp[0, 20:22, 3:9] = 2.5                        # a second, lower plateau of 12 equal cells
    kp[0, 30, 40] = 2.5
    d = rs.standard_normal((D, H, W)).astype(np.float32)
    d /= np.sqrt((d * d).sum(0, keepdims=True))
    for k in (50, 161, 165, 174, 400):
        kps, du8, cnt = run_hip([kp], [d], k, 0.5, 2, (160, 208), (160, 208))   # hypothetical sizes: unit scale factors
        okps, odu8 = to.run_inference_post(kp, d, (160, 208), (160, 208), k, 0.5, 2)
        assert_image(kps[0], du8[0], int(cnt[0]), okps, odu8, k)
    assert int(cnt[0]) == 161 + 13


def test_twenty_thousand_keypoints_as_the_reference_pipeline_asks():
    """run_pipeline.py:326-333 builds the extractor with num_keypoints=20480, nms_radius=1, score_threshold=0.4: the
    selection list then lives in the workspace instead of LDS."""
    rs = np.random.RandomState(8)
    H, W, D = 300, 400, 128
    kps_maps, d_maps = [], []
    for _ in range(2):
        kp = rs.standard_normal((4, H, W)).astype(np.float32)
        kp[0] *= 2.0
        d = rs.standard_normal((D, H, W)).astype(np.float32)
        d /= np.sqrt((d * d).sum(0, keepdims=True))
        kps_maps.append(kp), d_maps.append(d)
    kps, du8, cnt = run_hip(kps_maps, d_maps, 20480, 0.4, 1, (1200, 1600), (1190, 1596))
    for i in range(2):
        okps, odu8 = to.run_inference_post(kps_maps[i], d_maps[i], (1200, 1600), (1190, 1596), 20480, 0.4, 1)
        assert 5000 < len(okps) <= 20480
        assert_image(kps[i], du8[i], int(cnt[i]), okps, odu8, 20480)
    # and a cut: fewer slots than candidates
    kps, du8, cnt = run_hip(kps_maps[:1], d_maps[:1], 6000, 0.4, 1, (1200, 1600), (1190, 1596))
    okps, odu8 = to.run_inference_post(kps_maps[0], d_maps[0], (1200, 1600), (1190, 1596), 6000, 0.4, 1)
    assert len(okps) == 6000
    assert_image(kps[0], du8[0], int(cnt[0]), okps, odu8, 6000)


def test_extractor_end_to_end_and_database(tmp_path):
    """TrainableViTExtractor (ViT-S backbone, seeded random weights): `_run_inference` equals the oracle applied to the
    GPU's own head maps, `extract` writes 6-column keypoints + descriptors like the reference (trainable_vit_extractor.py:271-392)."""
    import sqlite3

    from vit_colmap_amd.features.trainable_vit_extractor import TrainableViTExtractor
    from vit_colmap_amd.utils import image_io

    ex = TrainableViTExtractor(model_name="dinov2_vits14", num_keypoints=300, descriptor_dim=128, device="cuda",
                               score_threshold=0.5, nms_radius=2)
    rs = np.random.RandomState(3)
    imgs = [np.clip(np.kron(rs.randint(0, 255, (12, 16, 3)), np.ones((10, 10, 1))) + rs.randint(-20, 20, (120, 160, 3)), 0, 255).astype(np.uint8)
            for _ in range(3)]
    # MIOpen's convolutions are not run-to-run deterministic, so the head maps are computed once and the same tensors are
    # handed to the extractor's selection and to the oracle
    kp_map, d_map = ex.head_maps(torch.from_numpy(imgs[0][None]).cuda())
    assert tuple(kp_map.shape) == (1, 4, 112 // 4, 154 // 4)
    ex.head_maps = lambda images: (kp_map, d_map)
    kps, du8 = ex._run_inference(imgs[0])
    del ex.head_maps
    assert kps.dtype == np.float32 and kps.shape[1] == 6 and du8.dtype == np.uint8 and du8.shape == (len(kps), 128)
    kps2, _ = ex._run_inference(imgs[0])                        # the real forward again: same result up to conv noise
    assert abs(len(kps2) - len(kps)) <= max(3, len(kps) // 20)
    okps, odu8 = to.run_inference_post(kp_map[0].cpu().numpy(), d_map[0].cpu().numpy(), (120, 160), (112, 154), 300, 0.5, 2)
    assert len(okps) > 0
    assert np.array_equal(kps.view(np.uint32), okps.view(np.uint32)) and np.array_equal(du8, odu8)
    assert (kps[:, 0] >= 0).all() and (kps[:, 0] <= 159).all() and (kps[:, 1] <= 119).all() and (kps[:, 2] == 1).all()
    assert (np.diff(kps[:, 4]) <= 0).all()

    d = tmp_path / "images"
    d.mkdir()
    for i, im in enumerate(imgs):
        image_io.imwrite(d / f"im_{i:02d}.png", im)
    db = tmp_path / "db.db"
    ex.extract(d, db, "SIMPLE_RADIAL")
    con = sqlite3.connect(str(db))
    rows = con.execute("SELECT image_id, rows, cols FROM keypoints ORDER BY image_id").fetchall()
    assert len(rows) == 3 and all(r[2] == 6 for r in rows)
    drows = con.execute("SELECT image_id, rows, cols FROM descriptors ORDER BY image_id").fetchall()
    assert [r[1] for r in drows] == [r[1] for r in rows] and all(r[2] == 128 for r in drows)
    blob = con.execute("SELECT data FROM keypoints WHERE image_id = 1").fetchone()[0]
    stored = np.frombuffer(blob, np.float32).reshape(-1, 6)
    assert abs(len(stored) - len(kps)) <= max(3, len(kps) // 20) and (stored[:, 2] == 1).all() and (stored[:, 5] == 0).all()
    cam = con.execute("SELECT model, params FROM cameras").fetchone()
    assert np.allclose(np.frombuffer(cam[1], np.float64), [160, 80, 60, 0.0])
    con.close()


def test_bf16_heads_track_the_float32_heads():
    """precision="bf16" runs the convolutional heads in bf16 (MIOpen) after folding BatchNorm in float32; the maps stay
    close to the float32 evaluation of the same weights (8-bit mantissa through 7 convolutions)."""
    from vit_colmap_amd.features.trainable_vit_extractor import TrainableViTExtractor

    frames = torch.from_numpy(np.random.RandomState(2).randint(0, 255, (2, 112, 154, 3)).astype(np.uint8)).cuda()
    maps = {}
    for prec in ("fp32", "bf16"):
        ex = TrainableViTExtractor(model_name="dinov2_vits14", num_keypoints=100, device="cuda", precision=prec, seed=4)
        # identical backbone features for both: the comparison is about the heads
        if prec == "fp32":
            tokens = torch.randn(2, 8 * 11, 384, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
        feats = ex.model.tokens_to_grid(tokens.to(ex.dtype), 8, 11)
        with torch.inference_mode():
            out = ex.model.forward_from_backbone_features(feats, target_size=(28, 38))
        maps[prec] = (out["keypoints"].float(), out["descriptors"].float())
    for a, b in zip(maps["fp32"], maps["bf16"]):
        rel = float((a - b).norm() / a.norm())
        assert rel < 5e-2, rel


def test_hip_heads_match_the_library_convolutions():
    """The bf16 product path runs upsampler / trunk / heads on vc_conv_taps_bf16 (model/hip_heads.py); the same weights
    through PyTorch-ROCm's bf16 convolutions (what round 2 shipped) give the same maps up to bf16 rounding of the
    intermediate activations, on a 640 x 480 token grid with a resize in the middle, and both track the float32 heads."""
    from vit_colmap_amd.features.trainable_vit_extractor import TrainableViTExtractor

    ex = TrainableViTExtractor(model_name="dinov2_vits14", num_keypoints=100, device="cuda", precision="bf16", seed=6)
    assert ex.model._hip_heads is not None
    hp, wp = 34, 45
    tokens = torch.randn(2, hp * wp, 384, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).to(torch.bfloat16)
    feats = ex.model.tokens_to_grid(tokens, hp, wp)
    with torch.inference_mode():
        a = ex.model.forward_from_backbone_features(feats, target_size=(120, 160))
        heads, ex.model._hip_heads = ex.model._hip_heads, None
        b = ex.model.forward_from_backbone_features(feats, target_size=(120, 160))
        ex.model._hip_heads = heads
    for key in ("keypoints", "descriptors", "features"):
        assert a[key].shape == b[key].shape, key
        rel = float((a[key].float() - b[key].float()).norm() / b[key].float().norm())
        assert rel < 2e-2, (key, rel)


def test_hip_heads_split_batches_that_exceed_the_32_bit_offsets(monkeypatch):
    """A batch whose x4 activation tensor would pass 4 GiB goes through vc_conv_taps_bf16 in chunks: forcing the chunk size
    down to one image gives the same maps as the whole batch."""
    from vit_colmap_amd.features.trainable_vit_extractor import TrainableViTExtractor
    from vit_colmap_amd.model import hip_heads

    ex = TrainableViTExtractor(model_name="dinov2_vits14", num_keypoints=50, device="cuda", precision="bf16", seed=8)
    tokens = torch.randn(3, 8 * 11, 384, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)).to(torch.bfloat16)
    with torch.inference_mode():
        whole = ex.model._hip_heads(tokens, 8, 11, (28, 38))
        monkeypatch.setattr(hip_heads, "MAX_ACTIVATION_BYTES", 16 * 8 * 11 * 512 * 2 * 1.5)      # room for one image
        split = ex.model._hip_heads(tokens, 8, 11, (28, 38))
    for key in whole:
        assert whole[key].shape == split[key].shape and torch.equal(whole[key], split[key]), key
