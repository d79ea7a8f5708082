"""GPU end-to-end tests through the reference's own interfaces (plugin API, DB contract)."""
import os
import sqlite3

import numpy as np
import pytest
import torch

from oracle import matcher_oracle as mo
from oracle import preprocess_oracle as po
from oracle import select_oracle as so
from oracle import vit_oracle
from test_host_logic import checkerboard
from vit_colmap_amd.database import ColmapDatabase
from vit_colmap_amd.utils import Config, image_io

pytestmark = pytest.mark.gpu


def synthetic_image(k, w=640, h=480):
    """checkerboard (reference tests/test_smoke_e2e.py:10-17) shifted + seeded noise."""
    rs = np.random.RandomState(1000 + k)
    img = np.roll(checkerboard(w, h), (7 * k, 5 * k), (1, 0)).astype(np.int16)
    img += rs.randint(-40, 41, img.shape).astype(np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


def test_smoke_pipeline_like_reference(tmp_path):
    """reference tests/test_smoke_e2e.py:20-76 on the HIP matcher (BASELINE config 1)."""
    from vit_colmap_amd.pipeline import Pipeline

    img_dir = tmp_path / "images"
    img_dir.mkdir(parents=True)
    for i, shift in enumerate([(0, 0), (50, 30), (100, 60)]):
        image_io.imwrite(img_dir / f"image_{i:03d}.png", np.roll(checkerboard(), shift, (1, 0)))
    db_path = tmp_path / "database.db"
    config = Config()
    config.camera.model = "PINHOLE"
    config.extractor.extractor_type = "dummy"
    config.do_matching = True
    config.do_reconstruction = False
    result = Pipeline(config=config).run(image_dir=img_dir, output_dir=tmp_path / "output", db_path=db_path)
    assert db_path.exists() and (tmp_path / "output").exists() and result is None
    with ColmapDatabase.open_database(str(db_path)) as db:
        assert ColmapDatabase.get_db_count(db, "num_cameras") >= 1
        assert ColmapDatabase.get_db_count(db, "num_images") == 3
        assert ColmapDatabase.get_db_count(db, "num_matched_image_pairs") >= 1
        for img_id in range(1, 4):
            assert db.exists_keypoints(img_id) and db.exists_descriptors(img_id)
        # un-normalised Dummy descriptors saturate the angle: every match is rejected (SURVEY.md §8 a-M)
        assert db.num_matches() == 0 and db.num_matched_image_pairs() == 3


def test_match_exhaustive_db_in_db_out_equals_oracle(tmp_path):
    from util_data import image_set
    from vit_colmap_amd.matching import match_exhaustive

    desc, counts = image_set(3, 6, 300, 128, kind="scene", counts=[300, 250, 0, 300, 17, 128], noise=0.1)
    db_path = tmp_path / "m.db"
    db = ColmapDatabase(str(db_path))
    cam = db.add_pinhole_camera(640, 480, 640, 640, 320, 240)
    for k in range(6):
        i = db.add_image(f"im{k}.png", cam)
        if counts[k]:
            db.add_keypoints(i, np.zeros((counts[k], 2), np.float32))
            db.add_descriptors(i, desc[k, : counts[k]])
    db.db.close()
    opts = Config().matching.to_matching_options()
    stats = match_exhaustive(database_path=str(db_path), matching_options=opts)
    assert stats["pairs"] == 15
    total = 0
    with ColmapDatabase.open_database(str(db_path)) as h:
        assert h.num_matched_image_pairs() == 15                # one row per pair, empty ones included
        for a in range(6):
            for b in range(a + 1, 6):
                ref = mo.match_pair(desc[a, : counts[a]], desc[b, : counts[b]])
                got = h.read_matches(a + 1, b + 1)
                assert got.dtype == np.uint32 and np.array_equal(got, ref), (a, b)
                total += len(ref)
    assert total == stats["matches"] and total > 100
    legacy = Config().matching._to_sift_options_legacy()          # pycolmap 3.12 calling convention
    legacy.cross_check = False
    match_exhaustive(database_path=str(db_path), sift_options=legacy)
    with ColmapDatabase.open_database(str(db_path)) as h:
        assert np.array_equal(h.read_matches(1, 2), mo.match_pair(desc[0, :300], desc[1, :250], cross_check=False))


def test_preprocess_bit_exact_resize_and_layouts():
    from vit_colmap_amd.features import hip_preprocess as hp

    imgs = np.stack([synthetic_image(k) for k in range(3)])
    d = torch.from_numpy(imgs).cuda()
    nchw, resized = hp.preprocess(d, torch.float32, "nchw", want_resized=True)
    patches = hp.preprocess(d, torch.float32, "patches")
    p16 = hp.preprocess(d, torch.bfloat16, "patches")
    for k in range(3):
        x, r = po.preprocess(imgs[k])
        assert r.shape == (476, 630, 3)
        assert np.array_equal(resized[k].cpu().numpy(), r)                        # uint8 resize: bit-exact
        np.testing.assert_allclose(nchw[k].cpu().numpy(), x, rtol=1e-6, atol=1e-6)
        assert np.array_equal(patches[k].cpu().numpy(), po.patchify(nchw[k].cpu().numpy()))
    assert torch.equal(p16, patches.to(torch.bfloat16))                           # RN-even cast
    ppad = hp.preprocess(d, torch.bfloat16, "patches_pad")                        # wave-per-patch kernel of the ViT-S path
    assert tuple(ppad.shape) == (3, 34 * 45, 640)
    assert torch.equal(ppad[..., :588], p16) and not bool(ppad[..., 588:].any())
    odd = torch.from_numpy(np.ascontiguousarray(imgs[:2, :101, :211])).cuda()     # 7 x 15 patches: partial last workgroup
    assert torch.equal(hp.preprocess(odd, torch.bfloat16, "patches_pad")[..., :588], hp.preprocess(odd, torch.bfloat16, "patches"))
    # strong down-scaling through the C ABI (source window of a patch larger than its LDS staging: taps from global memory)
    # and an up-scaling: padded bf16 patches == unpadded bf16 patches
    from vit_colmap_amd import _lib
    lib = _lib.load()
    for (oh, ow) in ((56, 84), (140, 98), (476, 630), (966, 1274)):
        a = torch.empty((3, (oh // 14) * (ow // 14), 588), dtype=torch.bfloat16, device="cuda")
        b = torch.full((3, (oh // 14) * (ow // 14), 640), 7.0, dtype=torch.bfloat16, device="cuda")
        for lay, o in ((1, a), (2, b)):
            _lib.check(lib.vc_preprocess_u8(_lib.ptr(d), 3, 480, 640, oh, ow, 1, lay, _lib.ptr(o), None, _lib.stream_ptr()), "vc_preprocess_u8")
        torch.cuda.synchronize()
        assert torch.equal(b[..., :588], a) and not bool(b[..., 588:].any()), (oh, ow)
    same = hp.preprocess(torch.from_numpy(np.ascontiguousarray(imgs[:, :476, :630])).cuda(), torch.float32, "nchw",
                         want_resized=True)[1]
    assert np.array_equal(same.cpu().numpy(), imgs[:, :476, :630])                # no resize when already aligned


def test_vit_forward_gpu_against_fp32_oracle():
    from vit_colmap_amd.features import hip_preprocess as hp
    from vit_colmap_amd.vit import build_dinov2

    model = build_dinov2("dinov2_vits14").init_random(seed=11).eval()
    sd = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    imgs = np.stack([synthetic_image(k) for k in range(2)])
    x = torch.stack([torch.from_numpy(po.preprocess(im)[0]) for im in imgs])
    with torch.no_grad():
        ref = vit_oracle.forward_patch_tokens(sd, x, model.arch.heads)           # CPU float32
    d = torch.from_numpy(imgs).cuda()
    out = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = build_dinov2("dinov2_vits14")
        m.load_state_dict(model.state_dict())
        m = m.eval().fold_layerscale().to("cuda", dt)
        with torch.inference_mode():
            out[name] = m.forward_patch_tokens(hp.preprocess(d, dt, "patches"), 34, 45).float().cpu()
    err32 = (out["fp32"] - ref).abs().max().item()
    err16 = (out["bf16"] - ref).abs().max().item()
    rel16 = ((out["bf16"] - ref).norm() / ref.norm()).item()
    print(f"ViT-S tokens vs fp32 oracle: fp32 max abs err {err32:.2e}; bf16 max abs err {err16:.2e}, rel L2 {rel16:.2e}")
    assert err32 < 2e-3          # fp32 on GPU: summation-order noise through 12 layers
    assert rel16 < 3e-2          # bf16 compute: 8-bit mantissa through 12 layers (reported in DESIGN.md)


def test_vit_extractor_contract_like_reference(tmp_path):
    """reference tests/test_vit_integration.py:18-146, 205-231 (shapes, dtypes, row counts)."""
    from vit_colmap_amd.features.vit_extractor import ViTExtractor

    ex = ViTExtractor(model_name="dinov2_vits14", num_keypoints=1024, descriptor_dim=128, device="cuda")
    img = np.random.RandomState(0).randint(0, 255, (476, 644, 3), dtype=np.uint8)
    kp, desc = ex._run_inference(img)
    assert kp.dtype == np.float32 and desc.dtype == np.uint8
    assert kp.ndim == 2 and kp.shape[1] == 2 and desc.shape[1] == 128 and len(kp) == len(desc) > 0
    assert kp[:, 0].min() >= 0 and kp[:, 0].max() <= 644 and kp[:, 1].max() <= 476
    assert ex.descriptor_projection is not None and tuple(ex.descriptor_projection.shape) == (384, 128)

    d = tmp_path / "images"
    d.mkdir()
    for k in range(3):
        image_io.imwrite(d / f"test_{k}.png", synthetic_image(k, 644, 476))
    (d / "z_broken.png").write_bytes(b"nope")                                     # unreadable: skipped, no row
    db_path = tmp_path / "test.db"
    ex.extract(d, db_path, "SIMPLE_PINHOLE")
    conn = sqlite3.connect(str(db_path))
    cur = conn.cursor()
    assert cur.execute("SELECT COUNT(*) FROM images").fetchone()[0] == 3
    assert cur.execute("SELECT COUNT(*) FROM keypoints").fetchone()[0] == 3
    assert cur.execute("SELECT COUNT(*) FROM descriptors").fetchone()[0] == 3
    rows = cur.execute("SELECT k.rows, k.cols, d.rows, d.cols FROM keypoints k JOIN descriptors d USING(image_id)").fetchall()
    assert all(r[1] == 2 and r[3] == 128 and r[0] == r[2] > 0 for r in rows)
    conn.close()


def test_vit_extractor_fp32_equals_oracle_chain_on_its_own_tokens():
    """Whole extractor in float32: keypoints / descriptors equal the oracle's selection chain run on
    the tokens the GPU produced (bit-exact indices), and the tokens are close to the CPU oracle's."""
    from vit_colmap_amd.features.vit_extractor import ViTExtractor

    ex = ViTExtractor(model_name="dinov2_vits14", num_keypoints=512, descriptor_dim=384, device="cuda",
                      precision="fp32", seed=4)
    img = synthetic_image(5)
    kp, desc = ex._run_inference(img)
    d = torch.from_numpy(img[None]).cuda()
    tokens, hp, wp = ex._tokens(d)
    fmap = tokens[0].float().cpu().numpy().T.reshape(384, hp, wp)
    from vit_colmap_amd.features import hip_select as hs

    score = hs.score_map(hs.structure_tensor(tokens, hp, wp), hp, wp, "harris").cpu().numpy()[0]
    ref = so.dense_to_sparse(fmap, (640, 480), (630, 476), 512, 384, "harris", None, score=score)
    assert len(kp) == len(ref["keypoints"]) > 50
    assert np.array_equal(kp, ref["keypoints"])
    assert np.abs(desc.astype(int) - ref["desc_u8"].astype(int)).max() <= 1


def test_extract_then_match_full_path(tmp_path):
    """config 2 + 3 in miniature: ViT-S extract of 6 images, then exhaustive matching; the match rows
    equal the oracle matcher run on the descriptors that were written."""
    from vit_colmap_amd.features.vit_extractor import ViTExtractor
    from vit_colmap_amd.matching import match_exhaustive

    d = tmp_path / "images"
    d.mkdir()
    for k in range(6):
        image_io.imwrite(d / f"img_{k:02d}.png", synthetic_image(k))
    db_path = tmp_path / "db.db"
    ex = ViTExtractor(model_name="dinov2_vits14", num_keypoints=512, descriptor_dim=384, device="cuda", batch_size=4)
    ex.extract(d, db_path, "PINHOLE")
    stats = match_exhaustive(database_path=str(db_path), matching_options=Config().matching.to_matching_options())
    assert stats["pairs"] == 15
    with ColmapDatabase.open_database(str(db_path)) as h:
        assert h.num_images() == 6 and h.num_matched_image_pairs() == 15
        descs = [h.read_descriptors(i) for i in range(1, 7)]
        assert all(x is not None and x.shape[1] == 384 for x in descs)
        for a in range(6):
            for b in range(a + 1, 6):
                assert np.array_equal(h.read_matches(a + 1, b + 1), mo.match_pair(descs[a], descs[b]))


def test_rccl_one_rank_group_moves_device_descriptor_blocks():
    """RCCL itself (torch.distributed backend "nccl" on ROCm) executes on the visible GPU: a one-rank
    process group, the data path's collective (dist.all_gather_descriptors -> all_gather_into_tensor on device tensors)
    and the error hand-shake's all_reduce.  One rank is all a one-GPU box allows (RCCL refuses two ranks on one device);
    the multi-rank logic is covered by the world-size-2 gloo tests."""
    import socket

    import torch.distributed as dist

    from vit_colmap_amd import dist as vd

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl" and not vd.is_distributed()
        g = torch.Generator(device="cuda").manual_seed(3)
        desc = torch.randint(0, 256, (50, 512, 384), dtype=torch.uint8, device="cuda", generator=g)
        counts = torch.randint(0, 513, (50,), dtype=torch.int32, device="cuda", generator=g)
        all_desc, all_counts = vd.all_gather_descriptors(desc, counts, force_collective=True)
        torch.cuda.synchronize()
        assert all_desc.data_ptr() != desc.data_ptr()                  # went through the collective, not the early return
        assert torch.equal(all_desc, desc) and torch.equal(all_counts, counts)
        t = torch.tensor([7], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == 7
        print(f"[RCCL one-rank smoke] all_gather_into_tensor of {desc.numel() / 1e6:.1f} MB uint8 + counts on {torch.cuda.get_device_name(0)}: ok")
    finally:
        dist.destroy_process_group()
