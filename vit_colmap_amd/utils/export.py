"""Metrics export: `{results}/{dataset}/{scene}/{extractor}.json` plus one appended row in
`{results}/{dataset}/summary.csv` — the files the reference's comparison scripts read
(reference vit_colmap/utils/export.py:14-57 `export_json` / `load_json`, :59-171 `export_csv_row`, :254-280
`export_metrics`; called from pipeline/run_pipeline.py:211-271).  Same file layout, JSON keys and CSV column names,
so results written here and by the reference can sit in one results directory.
Reconstruction columns are filled with zeros: the mapper is outside the accelerated path."""
import csv
import json
import logging
from dataclasses import asdict, dataclass, field
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, List, Optional

from .metrics import FeatureMetrics, MatchingMetrics, MetricsExtractor

logger = logging.getLogger(__name__)

_META = ["dataset", "scene", "extractor_type", "timestamp"]
_FEATURE_COLS = ["total_images", "total_keypoints", "avg_keypoints_per_image", "min_keypoints", "max_keypoints", "median_keypoints"]
_MATCH_COLS = ["total_image_pairs", "matched_pairs", "verified_pairs", "match_rate", "total_raw_matches", "avg_raw_matches",
               "median_raw_matches", "total_inlier_matches", "avg_inlier_matches", "median_inlier_matches", "inlier_ratio"]
_RECON_COLS = ["num_reconstructions", "registered_images", "registration_rate", "total_3d_points", "avg_track_length",
               "avg_reprojection_error"]
CSV_COLUMNS = _META + _FEATURE_COLS + _MATCH_COLS + _RECON_COLS
_DECIMALS = {"inlier_ratio": 4, "avg_reprojection_error": 4}


@dataclass
class MetricsResult:
    dataset: str
    scene: str
    extractor_type: str
    timestamp: str
    features: FeatureMetrics
    matching: MatchingMetrics
    reconstruction: Optional[dict] = None
    config: Dict[str, Any] = field(default_factory=dict)

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "MetricsResult":
        return cls(dataset=d["dataset"], scene=d["scene"], extractor_type=d["extractor_type"], timestamp=d["timestamp"],
                   features=FeatureMetrics(**d["features"]), matching=MatchingMetrics(**d["matching"]),
                   reconstruction=d.get("reconstruction"), config=d.get("config", {}))


def extract_all_metrics(db_path, dataset: str, scene: str, extractor_type: str, config: Optional[dict] = None,
                        min_threshold: Optional[int] = None) -> MetricsResult:
    ex = MetricsExtractor(db_path)
    return MetricsResult(dataset=dataset, scene=scene, extractor_type=extractor_type,
                         timestamp=datetime.now().isoformat(), features=ex.extract_feature_metrics(),
                         matching=ex.extract_matching_metrics(min_threshold), config=dict(config or {}))


class MetricsExporter:
    @staticmethod
    def export_json(metrics: MetricsResult, output_path: Path, indent: int = 2, overwrite: bool = True) -> None:
        output_path = Path(output_path)
        output_path.parent.mkdir(parents=True, exist_ok=True)
        if output_path.exists() and not overwrite:
            logger.warning(f"File already exists, skipping: {output_path}")
            return
        output_path.write_text(json.dumps(metrics.to_dict(), indent=indent))
        logger.info(f"Exported metrics to: {output_path}")

    @staticmethod
    def load_json(input_path: Path) -> MetricsResult:
        return MetricsResult.from_dict(json.loads(Path(input_path).read_text()))

    @staticmethod
    def csv_row(metrics: MetricsResult) -> Dict[str, Any]:
        row: Dict[str, Any] = {k: getattr(metrics, k) for k in _META}
        for cols, obj in ((_FEATURE_COLS, metrics.features), (_MATCH_COLS, metrics.matching)):
            for c in cols:
                v = getattr(obj, c)
                row[c] = f"{v:.{_DECIMALS.get(c, 2)}f}" if isinstance(v, float) else v
        rec = metrics.reconstruction or {}
        for c in _RECON_COLS:
            v = rec.get(c, 0)
            row[c] = f"{v:.{_DECIMALS.get(c, 2)}f}" if c in ("registration_rate", "avg_track_length", "avg_reprojection_error") else v
        return row

    @staticmethod
    def export_csv_row(metrics: MetricsResult, output_path: Path, append: bool = True) -> None:
        output_path = Path(output_path)
        output_path.parent.mkdir(parents=True, exist_ok=True)
        new_file = not (append and output_path.exists())
        with open(output_path, "a" if append else "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=CSV_COLUMNS)
            if new_file:
                w.writeheader()
            w.writerow(MetricsExporter.csv_row(metrics))

    @staticmethod
    def load_all_metrics(results_dir: Path) -> List[MetricsResult]:
        out = []
        for p in sorted(Path(results_dir).rglob("*.json")):
            try:
                out.append(MetricsExporter.load_json(p))
            except (KeyError, TypeError, json.JSONDecodeError):
                logger.warning(f"not a metrics file, skipped: {p}")
        return out


def export_metrics(metrics: MetricsResult, base_dir: Path, formats=("json", "csv")) -> None:
    """{base_dir}/{dataset}/{scene}/{extractor_type}.json and a row appended to {base_dir}/{dataset}/summary.csv."""
    base_dir = Path(base_dir)
    scene_dir = base_dir / metrics.dataset / metrics.scene
    scene_dir.mkdir(parents=True, exist_ok=True)
    if "json" in formats:
        MetricsExporter.export_json(metrics, scene_dir / f"{metrics.extractor_type}.json")
    if "csv" in formats:
        MetricsExporter.export_csv_row(metrics, base_dir / metrics.dataset / "summary.csv", append=True)
    logger.info(f"Metrics exported to {scene_dir}")
