"""Hybrid extractor descriptor sampling: sub-pixel bilinear sampling + RootSIFT.  The oracle is pinned by
goldens produced by the reference's own `_extract_descriptors_at_keypoints` (tests/golden/make_golden_hybrid.py); the HIP
kernel is compared with the oracle and the goldens through the C ABI."""
import os

import numpy as np
import pytest

from cases_hybrid import CASES, make_inputs
from oracle import select_oracle as so

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check_u8(got, ref):
    """Truncating quantiser on float32 values that may differ in the last bit (different summation order of the L1 / L2
    norms): at most 1 LSB, on a small share of the entries."""
    diff = np.abs(got.astype(int) - ref.astype(int))
    assert diff.max() <= 1, int(diff.max())
    assert (diff != 0).mean() < 5e-3, float((diff != 0).mean())


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_against_reference_golden(case):
    fmap, kp, proj = make_inputs(case)
    gold = np.load(os.path.join(GOLD, f"hybrid_{case['name']}.npz"))["desc_u8"]
    u8, f32 = so.descriptors_at_keypoints(fmap, kp, case["original_wh"], case["feature_wh"], case["dd"], proj)
    assert u8.shape == gold.shape
    _check_u8(u8, gold)
    assert np.allclose(np.linalg.norm(f32, axis=1), 1.0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_hip_describe_at_against_oracle_and_golden(case, dtype):
    import torch

    from vit_colmap_amd.features import hip_select as hs

    fmap, kp, proj = make_inputs(case)
    C, H, W = fmap.shape
    tok = torch.from_numpy(np.ascontiguousarray(fmap.reshape(C, H * W).T))[None].cuda()
    if dtype == "bf16":
        tok = tok.to(torch.bfloat16)
        fmap = tok[0].float().cpu().numpy().T.reshape(C, H, W).copy()       # the oracle sees the same rounded tokens
    n = len(kp)
    kmax = n + 3                                                             # rows beyond the count must come back zero
    kpb = np.zeros((2, kmax, 2), np.float32)
    kpb[0, :n] = kp
    kpb[1, : n // 2] = kp[: n // 2]
    cnt = torch.tensor([n, n // 2], dtype=torch.int32, device="cuda")
    toks = torch.cat([tok, tok]).contiguous()
    pj = None if proj is None else torch.from_numpy(proj).cuda()
    u8, f32 = hs.describe_at(toks, H, W, torch.from_numpy(kpb).cuda(), cnt, case["feature_wh"], case["original_wh"], pj,
                             rootsift=True, want_f32=True)
    u8, f32 = u8.cpu().numpy(), f32.cpu().numpy()
    ou8, of32 = so.descriptors_at_keypoints(fmap, kp, case["original_wh"], case["feature_wh"], case["dd"], proj)
    assert np.abs(f32[0, :n] - of32).max() <= 1e-3 * np.abs(of32).max()       # north_star: 1e-3 relative on float descriptors
    _check_u8(u8[0, :n], ou8)
    _check_u8(u8[1, : n // 2], ou8[: n // 2])
    assert not u8[0, n:].any() and not u8[1, n // 2:].any()
    if dtype == "f32":
        _check_u8(u8[0, :n], np.load(os.path.join(GOLD, f"hybrid_{case['name']}.npz"))["desc_u8"])


@pytest.mark.gpu
def test_hybrid_extractor_contract(tmp_path):
    """keypoints from a host callable (the detectors are OpenCV's and cv2 is absent here), descriptors from the GPU."""
    from test_e2e_gpu import synthetic_image
    from vit_colmap_amd.database import ColmapDatabase
    from vit_colmap_amd.features.hybrid_extractor import HybridViTExtractor
    from vit_colmap_amd.utils import image_io

    rs = np.random.RandomState(0)
    pts = np.stack([rs.uniform(0, 640, 400), rs.uniform(0, 480, 400)], axis=1).astype(np.float32)
    ex = HybridViTExtractor(model_name="dinov2_vits14", num_keypoints=400, descriptor_dim=128, keypoint_fn=lambda img: pts)
    kp, desc = ex._run_inference(synthetic_image(0))
    assert kp.dtype == np.float32 and np.array_equal(kp, pts) and desc.dtype == np.uint8 and desc.shape == (400, 128)
    assert tuple(ex.descriptor_projection.shape) == (384, 128)
    sq = (desc.astype(np.float64) ** 2).sum(axis=1)                              # RootSIFT rows: unit norm before the quantiser
    assert np.all(sq < 512.0 ** 2 * 1.01) and np.median(sq) > 0.8 * 512.0 ** 2
    d = tmp_path / "im"
    d.mkdir()
    for k in range(2):
        image_io.imwrite(d / f"i{k}.png", synthetic_image(k))
    ex.extract(d, tmp_path / "h.db", "PINHOLE")
    with ColmapDatabase.open_database(str(tmp_path / "h.db")) as h:
        assert h.num_images() == 2 and h.read_descriptors(2).shape == (400, 128) and h.read_keypoints(1).shape == (400, 2)
