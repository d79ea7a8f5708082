"""Feature / matching metrics read back from a COLMAP database.

Mirrors the database half of the reference's reporting — vit_colmap/utils/metrics.py:18-58 (`FeatureMetrics`,
`MatchingMetrics`), :144-180 (`MetricsExtractor.extract_feature_metrics`) and :182-268
(`extract_matching_metrics`) — with the same class / field / method names and the same SQL
(`keypoints(image_id, rows, cols)`, `images COUNT(*)`, `matches(pair_id, rows)`,
`two_view_geometries(pair_id, rows, config)`), over the standard library's sqlite3 only, so the reference's
JSON / CSV reporting keeps working on databases written by this package.  Reconstruction metrics
(metrics.py:270-, a pycolmap.Reconstruction walk) stay out of scope with the mapper.
`two_view_geometries` is written by matching/two_view.py (one row per matched pair, as COLMAP does [recalled]);
on a database without that step `verified_pairs`, the inlier statistics and `config_distribution` read 0 / {}
exactly as the reference's code does.
"""
import json
import sqlite3
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Dict, Optional

import numpy as np


@dataclass
class FeatureMetrics:
    total_images: int
    total_keypoints: int
    avg_keypoints_per_image: float
    min_keypoints: int
    max_keypoints: int
    median_keypoints: float


@dataclass
class MatchingMetrics:
    total_image_pairs: int
    matched_pairs: int
    verified_pairs: int
    match_rate: float            # percentage of possible pairs that have a matches row
    total_raw_matches: int
    avg_raw_matches: float
    min_raw_matches: int
    max_raw_matches: int
    median_raw_matches: float
    total_inlier_matches: int
    avg_inlier_matches: float
    min_inlier_matches: int
    max_inlier_matches: int
    median_inlier_matches: float
    inlier_ratio: float          # inliers / raw matches
    verification_rate: float = 0.0   # verified pairs / matched pairs (percentage)
    pairs_above_threshold: int = 0
    config_distribution: Dict[str, int] = field(default_factory=dict)


def _stats(counts):
    if not counts:
        return 0, 0.0, 0, 0, 0.0
    return int(sum(counts)), float(np.mean(counts)), int(min(counts)), int(max(counts)), float(np.median(counts))


class MetricsExtractor:
    """`MetricsExtractor(db_path, output_dir)` as in the reference (metrics.py:117-142); only the database is read."""

    # COLMAP TwoViewGeometry configuration names (metrics.py:120-131)
    CONFIG_NAMES = {0: "UNDEFINED", 1: "DEGENERATE", 2: "CALIBRATED", 3: "UNCALIBRATED", 4: "PLANAR",
                    5: "PANORAMIC", 6: "PLANAR_OR_PANORAMIC", 7: "WATERMARK", 8: "MULTIPLE", 9: "CALIBRATED_RIG"}

    def __init__(self, db_path, output_dir=None):
        self.db_path = Path(db_path)
        self.output_dir = None if output_dir is None else Path(output_dir)

    def _rows(self, sql):
        conn = sqlite3.connect(str(self.db_path))
        try:
            try:
                return conn.execute(sql).fetchall()
            except sqlite3.OperationalError:   # a table this database does not have (e.g. two_view_geometries)
                return []
        finally:
            conn.close()

    def extract_feature_metrics(self) -> FeatureMetrics:
        num_images = self._rows("SELECT COUNT(*) FROM images")[0][0]
        counts = [int(r[1]) for r in self._rows("SELECT image_id, rows, cols FROM keypoints")]
        total, avg, lo, hi, med = _stats(counts)
        return FeatureMetrics(total_images=int(num_images), total_keypoints=total, avg_keypoints_per_image=avg,
                              min_keypoints=lo, max_keypoints=hi, median_keypoints=med)

    def extract_matching_metrics(self, min_threshold: Optional[int] = None) -> MatchingMetrics:
        num_images = self._rows("SELECT COUNT(*) FROM images")[0][0]
        possible = num_images * (num_images - 1) // 2
        raw = [int(r[1]) for r in self._rows("SELECT pair_id, rows FROM matches")]
        tvg = self._rows("SELECT pair_id, rows, config FROM two_view_geometries")
        inl = [int(r[1]) for r in tvg]
        dist: Dict[str, int] = {}
        for r in tvg:
            name = self.CONFIG_NAMES.get(r[2], f"UNKNOWN({r[2]})")
            dist[name] = dist.get(name, 0) + 1
        t_raw, a_raw, lo_raw, hi_raw, med_raw = _stats(raw)
        t_inl, a_inl, lo_inl, hi_inl, med_inl = _stats(inl)
        above = sum(1 for c in inl if c >= min_threshold) if (min_threshold is not None and inl) else 0
        return MatchingMetrics(
            total_image_pairs=int(possible), matched_pairs=len(raw), verified_pairs=len(tvg),
            match_rate=(len(raw) / possible * 100) if possible > 0 else 0,
            total_raw_matches=t_raw, avg_raw_matches=a_raw, min_raw_matches=lo_raw, max_raw_matches=hi_raw,
            median_raw_matches=med_raw, total_inlier_matches=t_inl, avg_inlier_matches=a_inl,
            min_inlier_matches=lo_inl, max_inlier_matches=hi_inl, median_inlier_matches=med_inl,
            inlier_ratio=(t_inl / t_raw) if t_raw > 0 else 0,
            verification_rate=(len(tvg) / len(raw) * 100) if raw else 0,
            pairs_above_threshold=above, config_distribution=dist)

    def export_json(self, path, min_threshold: Optional[int] = None) -> dict:
        """{"features": ..., "matching": ...} with the reference's field names, written to `path`."""
        out = {"features": asdict(self.extract_feature_metrics()), "matching": asdict(self.extract_matching_metrics(min_threshold))}
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        Path(path).write_text(json.dumps(out, indent=2))
        return out
