"""SURVEY.md §8f-3 / §8b pinned by the REFERENCE's own reader: the fixture tests/golden/metrics_ref.json holds what
the reference's `MetricsExtractor` + `export_metrics` (vit_colmap/utils/metrics.py:144-391, export.py:14-280) produced
from a database written by this package (tests/golden/make_golden_metrics.py).  Here the database is rebuilt with the
same writer (digest must match: the layout the reference's SQL understood is the layout still written) and this
package's metrics / export code must reproduce the reference's JSON and CSV field for field."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from metrics_case import CONFIG, META, build_database, database_digest  # noqa: E402

from vit_colmap_amd.utils import export as ex  # noqa: E402
from vit_colmap_amd.utils.metrics import MetricsExtractor  # noqa: E402

TIMESTAMP = "2024-01-01T00:00:00"


@pytest.fixture(scope="module")
def ref():
    with open(os.path.join(HERE, "golden", "metrics_ref.json")) as f:
        return json.load(f)


@pytest.fixture()
def db_path(tmp_path, ref):
    path = tmp_path / "database.db"
    n_images, n_matches, n_tvg = build_database(path)
    assert {"images": n_images, "matches": n_matches, "two_view_geometries": n_tvg} == ref["rows"]
    return path


def test_writer_still_produces_the_database_the_reference_read(db_path, ref):
    assert database_digest(db_path) == ref["database_digest"]


def test_metrics_equal_the_reference_field_for_field(db_path, ref):
    mx = MetricsExtractor(db_path)
    from dataclasses import asdict

    assert asdict(mx.extract_feature_metrics()) == ref["json"]["features"]
    assert asdict(mx.extract_matching_metrics()) == ref["json"]["matching"]
    assert asdict(mx.extract_matching_metrics(min_threshold=40)) == ref["matching_min_threshold_40"]
    assert ref["matching_min_threshold_40"]["pairs_above_threshold"] >= 1
    assert ref["json"]["matching"]["min_raw_matches"] == 0          # empty match lists are rows too


def test_export_files_equal_the_reference(db_path, ref, tmp_path):
    result = ex.extract_all_metrics(db_path, config=dict(CONFIG), **META)
    result.timestamp = TIMESTAMP
    ex.export_metrics(result, tmp_path / "results")
    json_path = tmp_path / "results" / META["dataset"] / META["scene"] / f"{META['extractor_type']}.json"
    csv_path = tmp_path / "results" / META["dataset"] / "summary.csv"
    assert json.loads(json_path.read_text()) == ref["json"]
    assert json_path.read_text() == ref["json_text"]                # same key order and indentation
    assert csv_path.read_text() == ref["csv_text"]
    ex.export_metrics(result, tmp_path / "results")                 # a second scene appends a row, one header
    assert csv_path.read_text() == ref["csv_text_two_rows"]
    back = ex.MetricsExporter.load_json(json_path)                  # and the reference's JSON loads here
    assert back.matching.config_distribution == ref["json"]["matching"]["config_distribution"]
