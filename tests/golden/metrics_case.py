"""The database behind the metrics / export fixtures: written with THIS package's COLMAP writer from seeded numpy data
(used by make_golden_metrics.py, which runs the reference's reader over it, and by tests/test_metrics_golden.py)."""
import hashlib
import sqlite3

import numpy as np

KEYPOINT_COUNTS = [300, 114, 0, 512, 33, 256, 7]     # image 3 has no features: the extractors skip its rows
META = dict(dataset="SYNTH", scene="scene01", extractor_type="vit")
CONFIG = {"extractor": "vit", "num_keypoints": 512, "max_ratio": 0.8}


def build_database(path):
    """-> (number of images, number of matches rows, number of two_view_geometries rows)."""
    from vit_colmap_amd.database.colmap_db import ColmapDatabase

    rs = np.random.RandomState(20241005)
    db = ColmapDatabase(str(path))
    cam = db.add_pinhole_camera(640, 480, 640.0, 640.0, 320.0, 240.0)
    ids = []
    for k, n in enumerate(KEYPOINT_COUNTS):
        iid = db.add_image(f"img_{k:03d}.png", cam)
        ids.append(iid)
        if n == 0:
            continue
        db.add_keypoints(iid, (rs.rand(n, 2) * [640, 480]).astype(np.float32))
        db.add_descriptors(iid, rs.randint(0, 256, (n, 128)).astype(np.uint8))
    n_matches = n_tvg = 0
    for a in range(len(ids)):
        for b in range(a + 1, len(ids)):
            na, nb = KEYPOINT_COUNTS[a], KEYPOINT_COUNTS[b]
            if na == 0 or nb == 0:
                continue                                        # nothing to match: no row (as the matcher leaves it)
            m = int(rs.randint(0, min(na, nb) // 2 + 1)) if (a + b) % 4 else 0
            rows = np.stack([np.sort(rs.permutation(na)[:m]), rs.permutation(nb)[:m]], axis=1).astype(np.uint32)
            db.add_matches(ids[a], ids[b], rows)                # an empty list is still a row (COLMAP [recalled])
            n_matches += 1
            if m >= 15:                                         # verified pairs: inlier subset + configuration
                keep = np.sort(rs.permutation(m)[: max(15, int(m * (0.3 + 0.6 * rs.rand())))])
                config = 6 if (a * 7 + b) % 3 == 0 else 3       # PLANAR_OR_PANORAMIC / UNCALIBRATED
                F = rs.standard_normal((3, 3))
                H = rs.standard_normal((3, 3))
                db.db.write_two_view_geometry(ids[a], ids[b], rows[keep], config, F=F, H=H)
                n_tvg += 1
    db.db.commit()
    db.db.close()
    return len(ids), n_matches, n_tvg


def database_digest(path):
    """sha256 over the schema text and every row of every table (blobs included), in a fixed order."""
    conn = sqlite3.connect(str(path))
    h = hashlib.sha256()
    for name, sql in conn.execute("SELECT name, sql FROM sqlite_master WHERE type IN ('table','index') ORDER BY name"):
        h.update(repr((name, " ".join((sql or "").split()))).encode())
    for (name,) in conn.execute("SELECT name FROM sqlite_master WHERE type='table' AND name NOT LIKE 'sqlite_%' ORDER BY name").fetchall():
        for row in conn.execute(f"SELECT * FROM {name} ORDER BY 1"):
            h.update(repr(tuple(bytes(v) if isinstance(v, (bytes, memoryview)) else v for v in row)).encode())
    conn.close()
    return h.hexdigest()
