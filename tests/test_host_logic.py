"""CPU tests of the host-side mirror of the reference interface: config, database writer,
Dummy extractor, image I/O, pipeline dispatch and error conventions."""
import sqlite3
from pathlib import Path

import numpy as np
import pytest

from vit_colmap_amd.database import ColmapDatabase, SqliteColmapDatabase, pair_id_of, pair_id_to_image_ids
from vit_colmap_amd.features import BaseExtractor, DummyExtractor
from vit_colmap_amd.features.base_extractor import default_camera_params, list_images
from vit_colmap_amd.utils import Config, MatchingConfig, image_io


def checkerboard(w=640, h=480, tile=40):
    """reference tests/test_smoke_e2e.py:10-17"""
    img = np.zeros((h, w, 3), np.uint8)
    for y in range(0, h, tile):
        for x in range(0, w, tile):
            if ((x // tile) + (y // tile)) % 2 == 0:
                img[y:y + tile, x:x + tile] = 255
    return img


def test_config_defaults_match_reference():
    c = Config()
    assert c.extractor.extractor_type == "vit" and c.camera.model == "SIMPLE_PINHOLE"
    assert (c.matching.max_ratio, c.matching.max_distance, c.matching.cross_check) == (0.8, 0.7, True)
    assert c.matching.use_gpu is True and c.matching.num_threads == -1
    assert c.do_matching and c.do_reconstruction and c.reconstruction.min_num_matches == 15
    o = c.matching.to_matching_options()
    assert (o.sift.max_ratio, o.sift.max_distance, o.sift.cross_check) == (0.8, 0.7, True)
    legacy = MatchingConfig(max_ratio=0.9)._to_sift_options_legacy()
    assert legacy.max_ratio == 0.9
    assert "Extractor: vit" in c.summary()
    assert c.camera.get_default_params(640, 480) == [640, 320.0, 240.0]

    class A:
        camera_model = "PINHOLE"; extractor = "dummy"; vit_weights = None; skip_matching = True
        skip_reconstruction = True; verbose = False
    c2 = Config.from_args(A())
    assert c2.camera.model == "PINHOLE" and c2.extractor.extractor_type == "dummy"
    assert not c2.do_matching and not c2.do_reconstruction


def test_camera_defaults_and_errors():
    assert default_camera_params("PINHOLE", 640, 480) == [640, 640, 320.0, 240.0]
    with pytest.raises(ValueError, match="Unsupported camera model"):
        default_camera_params("FISHEYE", 1, 1)


def test_database_schema_roundtrip_and_reference_sql(tmp_path):
    path = tmp_path / "database.db"
    db = ColmapDatabase(str(path))
    cam = db.add_pinhole_camera(640, 480, 640.0, 640.0, 320.0, 240.0)
    ids = [db.add_image(f"image_{i:03d}.png", cam) for i in range(3)]
    assert ids == [1, 2, 3]                                        # test_smoke_e2e.py:68 relies on 1..3
    kp = np.arange(12, dtype=np.float64).reshape(6, 2)
    desc = (np.arange(6 * 128) % 256).reshape(6, 128)
    db.add_keypoints(1, kp)
    db.add_descriptors(1, desc)
    db.add_matches(2, 1, np.array([[0, 5], [3, 4]]))               # reversed ids: stored relative to (1, 2)
    db.add_matches(1, 3, np.zeros((0, 2)))
    db.commit()
    db.db.close()
    with ColmapDatabase.open_database(str(path)) as h:
        assert ColmapDatabase.get_db_count(h, "num_cameras") == 1
        assert ColmapDatabase.get_db_count(h, "num_images") == 3
        assert ColmapDatabase.get_db_count(h, "num_matched_image_pairs") == 2
        assert h.exists_keypoints(1) and h.exists_descriptors(1) and not h.exists_keypoints(2)
        assert h.read_keypoints(1).dtype == np.float32 and np.array_equal(h.read_keypoints(1), kp)
        assert h.read_descriptors(1).dtype == np.uint8 and np.array_equal(h.read_descriptors(1), desc)
        assert np.array_equal(h.read_matches(1, 2), [[5, 0], [4, 3]])
        assert np.array_equal(h.read_matches(2, 1), [[0, 5], [3, 4]])
        assert h.read_matches(1, 3).shape == (0, 2)
        assert h.read_camera(1).model == "PINHOLE"
    # the SQL the reference itself runs (metrics.py:158,197,202,207; test_vit_integration.py:129-138,209-213)
    conn = sqlite3.connect(str(path))
    cur = conn.cursor()
    assert cur.execute("SELECT COUNT(*) FROM images").fetchone()[0] == 3
    assert cur.execute("SELECT image_id, rows, cols FROM keypoints").fetchall() == [(1, 6, 2)]
    assert cur.execute("SELECT image_id, rows, cols FROM descriptors").fetchall() == [(1, 6, 128)]
    assert sorted(cur.execute("SELECT pair_id, rows FROM matches").fetchall()) == [
        (pair_id_of(1, 2), 2), (pair_id_of(1, 3), 0)]
    assert cur.execute("SELECT pair_id, rows, config FROM two_view_geometries").fetchall() == []
    conn.close()
    assert pair_id_of(2, 1) == 2147483647 + 2 and pair_id_to_image_ids(pair_id_of(7, 3)) == (3, 7)


def test_unique_image_names_and_bad_camera(tmp_path):
    db = SqliteColmapDatabase(str(tmp_path / "d.db"))
    from vit_colmap_amd.database import Camera, Image

    cid = db.write_camera(Camera("SIMPLE_PINHOLE", 10, 10, [10, 5, 5]))
    db.write_image(Image("a.png", cid))
    with pytest.raises(sqlite3.IntegrityError):
        db.write_image(Image("a.png", cid))
    with pytest.raises(ValueError):
        db.write_camera(Camera("NOT_A_MODEL", 1, 1, []))
    db.close()


def test_image_io_roundtrip_and_listing(tmp_path):
    img = checkerboard()
    img[..., 0] = 10                                                # make B and R differ
    assert image_io.imwrite(tmp_path / "b.png", img)
    image_io.imwrite(tmp_path / "a.PNG", img)
    (tmp_path / "notes.txt").write_text("x")
    back = image_io.imread(tmp_path / "b.png")
    assert back.dtype == np.uint8 and np.array_equal(back, img)     # BGR in, BGR out
    assert image_io.imread(tmp_path / "notes.txt") is None
    assert [f.name for f in list_images(tmp_path)] == ["a.PNG", "b.png"]


def test_image_io_pillow_fallback_applies_exif_orientation(tmp_path, monkeypatch):
    """cv2.imread rotates by the EXIF orientation tag; the Pillow fallback has to do the same or keypoints of
    portrait photos land in a transposed frame."""
    PILImage = pytest.importorskip("PIL.Image")
    rgb = np.zeros((4, 6, 3), np.uint8)
    rgb[0, :, 0] = 255                                              # red top row
    im = PILImage.fromarray(rgb)
    exif = PILImage.Exif()
    exif[0x0112] = 6                                                # "rotate 90 CW to display"
    im.save(tmp_path / "r.png", exif=exif)                          # lossless: the colour test below is exact
    monkeypatch.setattr(image_io, "_cv2", None)
    back = image_io.imread(tmp_path / "r.png")
    assert back.shape == (6, 4, 3)                                  # rotated: 6 rows, 4 columns
    assert back[:, -1, 2].min() > 200 and back[:, 0, 2].max() < 60  # the red row is now the right column (BGR: channel 2)


def test_dummy_extractor_contract(tmp_path):
    assert issubclass(DummyExtractor, BaseExtractor)
    d = tmp_path / "images"
    d.mkdir()
    for i, shift in enumerate([(0, 0), (50, 30), (100, 60)]):
        image_io.imwrite(d / f"image_{i:03d}.png", np.roll(checkerboard(), shift, (1, 0)))
    db_path = tmp_path / "database.db"
    DummyExtractor(step=32).extract(d, db_path, "PINHOLE")
    g = np.load(Path(__file__).parent / "golden" / "dummy_640x480.npz")
    with ColmapDatabase.open_database(str(db_path)) as db:
        assert db.num_images() == 3 and db.num_cameras() == 1
        for i in (1, 2, 3):
            assert np.array_equal(db.read_keypoints(i), g["keypoints"])
            assert np.array_equal(db.read_descriptors(i), g["descriptors"])
        assert db.read_camera(1).params == list(g["camera_params"])
    with pytest.raises(ValueError):
        DummyExtractor().extract(d, tmp_path / "x.db", "FISHEYE")


def test_dummy_extractor_generates_images_when_dir_is_empty(tmp_path):
    d = tmp_path / "empty"
    DummyExtractor().extract(d, tmp_path / "db.db", "SIMPLE_PINHOLE")       # dummy_extractor.py:46-55
    assert len(list_images(d)) == 10
    with ColmapDatabase.open_database(str(tmp_path / "db.db")) as db:
        assert db.num_images() == 10


def test_vit_extractor_error_conventions(tmp_path):
    from vit_colmap_amd.features.vit_extractor import ViTExtractor

    with pytest.raises(ValueError, match="Unsupported model"):
        ViTExtractor(model_name="resnet50", device="cpu")
    with pytest.raises(ValueError, match="Unknown detection method"):
        ViTExtractor(model_name="dinov2_vits14", detection_method="fast", device="cpu")
    ex = ViTExtractor(model_name="dinov2_vits14", num_keypoints=64, descriptor_dim=384, device="cpu")
    empty = tmp_path / "none"
    empty.mkdir()
    with pytest.raises(ValueError, match="No images found"):
        ex.extract(empty, tmp_path / "a.db", "PINHOLE")
    (empty / "broken.png").write_bytes(b"not a png")
    with pytest.raises(ValueError, match="Failed to read first image"):
        ex.extract(empty, tmp_path / "b.db", "PINHOLE")
    d = tmp_path / "imgs"
    d.mkdir()
    image_io.imwrite(d / "x.png", checkerboard())
    with pytest.raises(ValueError, match="Unsupported camera model"):
        ex.extract(d, tmp_path / "c.db", "FISHEYE")
    from vit_colmap_amd._lib import HipLibraryError

    with pytest.raises(HipLibraryError):                           # no GPU here: must fail loudly
        ex._run_inference(checkerboard())


def test_pipeline_dispatch(tmp_path):
    from vit_colmap_amd.pipeline import Pipeline

    c = Config()
    c.extractor.extractor_type = "colmap_sift"
    with pytest.raises(NotImplementedError):
        Pipeline(c).run(tmp_path, tmp_path / "o", tmp_path / "d.db")
    c.extractor.extractor_type = "dummy"
    c.camera.model = "PINHOLE"
    c.do_matching = False
    c.do_reconstruction = False
    d = tmp_path / "images"
    d.mkdir()
    image_io.imwrite(d / "a.png", checkerboard())
    assert Pipeline(c).run(d, tmp_path / "out", tmp_path / "db" / "database.db") is None
    assert (tmp_path / "out").exists() and (tmp_path / "db" / "database.db").exists()


def test_metrics_extractor_reads_back_what_the_writer_stored(tmp_path):
    """the reference's database metrics (utils/metrics.py:144-268) over a database written here."""
    import json
    import sqlite3

    import numpy as np

    from vit_colmap_amd.database import ColmapDatabase
    from vit_colmap_amd.utils import MetricsExtractor

    db_path = tmp_path / "m.db"
    db = ColmapDatabase(str(db_path))
    cam = db.add_pinhole_camera(640, 480, 640.0, 640.0, 320.0, 240.0)
    counts = [5, 9, 2, 12]
    ids = []
    for i, n in enumerate(counts):
        iid = db.add_image(f"im{i}.png", cam)
        ids.append(iid)
        db.add_keypoints(iid, np.zeros((n, 2), np.float32))
        db.add_descriptors(iid, np.zeros((n, 8), np.uint8))
    raw = {(0, 1): 4, (0, 2): 0, (1, 3): 7}
    for (a, b), n in raw.items():
        db.add_matches(ids[a], ids[b], np.zeros((n, 2), np.uint32))
    db.commit()
    mx = MetricsExtractor(db_path, tmp_path)
    f = mx.extract_feature_metrics()
    assert (f.total_images, f.total_keypoints, f.min_keypoints, f.max_keypoints) == (4, 28, 2, 12)
    assert f.avg_keypoints_per_image == 7.0 and f.median_keypoints == 7.0
    m = mx.extract_matching_metrics(min_threshold=3)
    assert (m.total_image_pairs, m.matched_pairs, m.verified_pairs) == (6, 3, 0)
    assert m.match_rate == 50.0 and m.total_raw_matches == 11 and (m.min_raw_matches, m.max_raw_matches) == (0, 7)
    assert m.median_raw_matches == 4.0 and m.inlier_ratio == 0 and m.verification_rate == 0 and m.config_distribution == {}
    # verified pairs, as COLMAP's geometric verification would add them
    conn = sqlite3.connect(str(db_path))
    conn.execute("INSERT INTO two_view_geometries(pair_id, rows, cols, data, config) VALUES (?, 3, 2, NULL, 2)", (1,))
    conn.execute("INSERT INTO two_view_geometries(pair_id, rows, cols, data, config) VALUES (?, 5, 2, NULL, 3)", (2,))
    conn.commit()
    conn.close()
    m = mx.extract_matching_metrics(min_threshold=4)
    assert (m.verified_pairs, m.total_inlier_matches, m.pairs_above_threshold) == (2, 8, 1)
    assert abs(m.inlier_ratio - 8 / 11) < 1e-12 and abs(m.verification_rate - 200 / 3) < 1e-9
    assert m.config_distribution == {"CALIBRATED": 1, "UNCALIBRATED": 1}
    out = mx.export_json(tmp_path / "r" / "metrics.json", min_threshold=4)
    assert json.loads((tmp_path / "r" / "metrics.json").read_text()) == out and out["features"]["total_keypoints"] == 28


def test_two_view_geometry_of_a_swapped_pair_is_converted(tmp_path):
    """a geometry handed over as (larger id, smaller id) is stored for smaller -> larger — columns reversed,
    F / E transposed, H inverted, the relative pose inverted — and read back in whichever direction is asked for."""
    from vit_colmap_amd.database.colmap_db import SqliteColmapDatabase, _quat_to_rot

    db = SqliteColmapDatabase(str(tmp_path / "t.db"))
    rs = np.random.RandomState(0)
    H, F, E = rs.standard_normal((3, 3)), rs.standard_normal((3, 3)), rs.standard_normal((3, 3))
    q = rs.standard_normal(4)
    q /= np.linalg.norm(q)
    t = rs.standard_normal(3)
    m = np.array([[1, 2], [3, 4], [7, 0]], np.uint32)
    db.write_two_view_geometry(5, 3, m, 3, F=F, E=E, H=H, qvec=q, tvec=t)
    fwd, back = db.read_two_view_geometry(3, 5), db.read_two_view_geometry(5, 3)
    assert np.array_equal(fwd["inlier_matches"], m[:, ::-1]) and np.array_equal(back["inlier_matches"], m)
    assert np.allclose(fwd["F"], F.T) and np.allclose(fwd["E"], E.T) and np.allclose(fwd["H"], np.linalg.inv(H))
    assert np.allclose(back["F"], F) and np.allclose(back["H"], H) and np.allclose(back["qvec"], q) and np.allclose(back["tvec"], t)
    x = rs.standard_normal(3)                                   # a point through 5 -> 3 and back through the stored 3 -> 5
    y = _quat_to_rot(q) @ x + t
    assert np.allclose(_quat_to_rot(fwd["qvec"]) @ y + fwd["tvec"], x)
    db.close()
