#!/usr/bin/env python3
"""Pin the metrics / export path and the database layout (§8b) with the REFERENCE's own reader.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden_metrics.py

What it does
------------
* writes a small COLMAP database with THIS package's writer (tests/golden/metrics_case.py: seeded keypoints,
  descriptors, match lists incl. empty ones, two_view_geometries rows with two configurations);
* imports the reference's `vit_colmap/utils/metrics.py` and `vit_colmap/utils/export.py` with a stand-in `pycolmap`
  module (the wheel is not installed): the only thing the code run here asks of it is
  `pycolmap.Database.open(path).num_images` (colmap_db.py:47-75 -> metrics.py:153-154), which the stand-in answers with
  `SELECT COUNT(*) FROM images`; everything else the reference does is raw sqlite3 on the file
  (metrics.py:158,197,202,207) — i.e. the reference's own SQL reads the schema this package wrote;
* runs `MetricsExtractor.extract_all_metrics` and `export_metrics` (export.py:254-280) and stores what they wrote —
  `{dataset}/{scene}/{extractor}.json` and `{dataset}/summary.csv` — as fixtures, with the wall-clock `timestamp`
  replaced by a constant, plus `extract_matching_metrics(min_threshold=40)` and a digest of the database.

tests/test_metrics_golden.py rebuilds the database, checks the digest, and requires vit_colmap_amd.utils.metrics /
export to reproduce both files field for field.
"""
import contextlib
import json
import os
import sqlite3
import sys
import tempfile
import types
from dataclasses import asdict
from pathlib import Path

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from metrics_case import CONFIG, META, build_database, database_digest  # noqa: E402

REF = "/root/reference"
TIMESTAMP = "2024-01-01T00:00:00"


def install_pycolmap_stand_in():
    mod = types.ModuleType("pycolmap")

    class Database:
        def __init__(self, path=None):
            self._path = path

        @staticmethod
        def open(path):
            return Database(path)

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def close(self):
            pass

        @property
        def num_images(self):
            with contextlib.closing(sqlite3.connect(self._path)) as c:
                return c.execute("SELECT COUNT(*) FROM images").fetchone()[0]

    mod.Database = Database
    mod.Reconstruction = type("Reconstruction", (), {})
    mod.__version__ = "3.13.0"
    sys.modules["pycolmap"] = mod


def main():
    install_pycolmap_stand_in()
    sys.path.insert(0, REF)
    import importlib

    ref_metrics = importlib.import_module("vit_colmap.utils.metrics")
    ref_export = importlib.import_module("vit_colmap.utils.export")
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        db_path = tmp / "database.db"
        n_images, n_matches, n_tvg = build_database(db_path)
        ex = ref_metrics.MetricsExtractor(db_path, tmp / "out")
        result = ex.extract_all_metrics(config=dict(CONFIG), **META)
        thresholded = ex.extract_matching_metrics(min_threshold=40)
        ref_export.export_metrics(result, tmp / "results")
        json_text = (tmp / "results" / META["dataset"] / META["scene"] / f"{META['extractor_type']}.json").read_text()
        csv_text = (tmp / "results" / META["dataset"] / "summary.csv").read_text()
        # a second export appends a row (export.py:264-267): the header must not repeat
        ref_export.export_metrics(result, tmp / "results")
        csv_text2 = (tmp / "results" / META["dataset"] / "summary.csv").read_text()
        stamp = result.timestamp
        digest = database_digest(db_path)
    assert json_text.count(stamp) == 1 and csv_text.count(stamp) == 1
    out = {
        "database_digest": digest,
        "rows": {"images": n_images, "matches": n_matches, "two_view_geometries": n_tvg},
        "json": json.loads(json_text.replace(stamp, TIMESTAMP)),
        "json_text": json_text.replace(stamp, TIMESTAMP),
        "csv_text": csv_text.replace(stamp, TIMESTAMP),
        "csv_text_two_rows": csv_text2.replace(stamp, TIMESTAMP),
        "matching_min_threshold_40": json.loads(json.dumps(asdict(thresholded))),
    }
    path = os.path.join(HERE, "metrics_ref.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    m = out["json"]["matching"]
    print(f"wrote {path}: {n_images} images, {m['matched_pairs']} matches rows, {m['verified_pairs']} verified, "
          f"configs {m['config_distribution']}, digest {digest[:16]}")


if __name__ == "__main__":
    main()
