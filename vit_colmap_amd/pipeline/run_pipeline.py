"""Pipeline dispatcher for the hot path: extractor choice by `config.extractor.extractor_type`,
extraction into the COLMAP database, exhaustive matching (reference
vit_colmap/pipeline/run_pipeline.py:274-370, CLI :420-513).

Incremental mapping and plots stay outside the accelerated path; the database metrics and their
JSON / CSV export (§8f-3) are in utils/metrics.py and utils/export.py.  `do_reconstruction=True` hands the database to pycolmap if it
is importable and otherwise logs that the step is skipped.
"""
import argparse
import logging
from pathlib import Path
from typing import Optional

from ..database.colmap_db import ColmapDatabase
from ..features.base_extractor import BaseExtractor
from ..features.dummy_extractor import DummyExtractor
from ..utils.config import Config

logger = logging.getLogger(__name__)


class Pipeline:
    def __init__(self, config: Optional[Config] = None):
        self.config = config if config is not None else Config()
        self.last_stats = {}

    def _make_extractor(self) -> BaseExtractor:
        kind = self.config.extractor.extractor_type
        if kind == "dummy":                                   # run_pipeline.py:321-323
            logger.info("Using Dummy extractor")
            return DummyExtractor(step=32)
        if kind == "colmap_sift":                             # :323-325 — third-party C++ SIFT
            raise NotImplementedError(
                f"extractor_type={kind!r} is outside the accelerated hot path (SURVEY.md §2); "
                "use the reference implementation for it")
        if kind == "trainable_vit":                           # :326-333
            logger.info("Using Trainable ViT extractor")
            from ..features.trainable_vit_extractor import TrainableViTExtractor

            return TrainableViTExtractor(weights_path=self.config.extractor.vit_weights_path, num_keypoints=20480,
                                         nms_radius=1, score_threshold=0.4)
        logger.info("Using ViT extractor")                     # :335-339 (any other value -> ViT)
        from ..features.vit_extractor import ViTExtractor

        return ViTExtractor(weights_path=self.config.extractor.vit_weights_path)

    def run(self, image_dir: Path, output_dir: Path, db_path: Path, dataset: Optional[str] = None,
            scene: Optional[str] = None, results_dir: Optional[Path] = None):
        image_dir, output_dir, db_path = Path(image_dir), Path(output_dir), Path(db_path)
        camera_model = self.config.camera.model
        camera_params = self.config.camera.params
        output_dir.mkdir(parents=True, exist_ok=True)
        db_path.parent.mkdir(parents=True, exist_ok=True)

        extractor = self._make_extractor()
        from .. import dist as vd

        if vd.is_distributed():
            # one process per GPU (torchrun): images and pairs sharded, one descriptor all-gather, rank 0 writes (§8e)
            from .distributed import run_sharded

            if not hasattr(extractor, "_run_batch"):
                raise NotImplementedError(f"{type(extractor).__name__} has no batched device path to shard")
            logger.info("Extracting + matching on %d ranks...", vd.rank_world()[1])
            if hasattr(extractor, "sync_projection"):
                extractor.sync_projection(image_dir)      # the PCA / random projection is fitted once, on rank 0
            self.last_stats = run_sharded(image_dir, db_path, camera_model, camera_params, feature_fn=extractor._run_batch,
                                          matching_options=self.config.matching.to_matching_options(),
                                          do_matching=self.config.do_matching, device=str(getattr(extractor, "device", "cuda")))
            if vd.rank_world()[0] != 0:
                return None
        else:
            logger.info("Extracting features...")
            extractor.extract(image_dir, db_path, camera_model, camera_params)
            with ColmapDatabase.open_database(str(db_path)) as db_check:
                num_imgs = ColmapDatabase.get_db_count(db_check, "num_images")
                logger.info(f"Extracted features for {num_imgs} images")

            if self.config.do_matching:
                from ..matching import match_exhaustive

                logger.info("Running feature matching...")
                opts = self.config.matching.to_matching_options()
                self.last_stats = match_exhaustive(database_path=str(db_path), matching_options=opts)
        if self.config.do_matching:
            with ColmapDatabase.open_database(str(db_path)) as db_check:
                num_pairs = ColmapDatabase.get_db_count(db_check, "num_matched_image_pairs")
                logger.info(f"Matched {num_pairs} image pairs ({db_check.num_verified_image_pairs()} geometrically verified)")

        reconstructions = None
        if self.config.do_reconstruction:
            try:
                import pycolmap  # noqa: PLC0415
            except ImportError:
                logger.warning("3D reconstruction skipped: pycolmap is not installed and incremental mapping is "
                               "outside the accelerated path")
            else:
                with ColmapDatabase.open_database(str(db_path)) as db_check:
                    n_verified = db_check.num_verified_image_pairs()
                if n_verified == 0:
                    # the mapper only reads two_view_geometries: on a database without verified pairs it returns no model
                    # and says nothing
                    logger.warning("no geometrically verified image pair in the database (two_view_geometries is empty): "
                                   "incremental mapping will not find an initial pair")
                sparse_dir = output_dir / "sparse"
                sparse_dir.mkdir(parents=True, exist_ok=True)
                reconstructions = pycolmap.incremental_mapping(
                    database_path=str(db_path), image_path=str(image_dir), output_path=str(sparse_dir),
                    options=self.config.reconstruction.to_mapper_options())
        if dataset and scene:                                   # run_pipeline.py:406-415
            self.extract_and_export_metrics(db_path, output_dir, reconstructions, dataset, scene, results_dir)
        return reconstructions

    def extract_and_export_metrics(self, db_path, output_dir, reconstructions, dataset, scene, results_dir=None):
        """Database metrics -> `{results_dir}/{dataset}/{scene}/{extractor}.json` + summary.csv row
        (reference run_pipeline.py:211-271; reconstruction metrics stay empty: the mapper is out of scope)."""
        from ..utils.export import export_metrics, extract_all_metrics

        try:
            kind = self.config.extractor.extractor_type
            kind = "sift" if kind == "colmap_sift" else kind
            cfg = {"camera_model": self.config.camera.model, "min_num_matches": self.config.reconstruction.min_num_matches,
                   "matching_max_ratio": self.config.matching.max_ratio, "matching_use_gpu": self.config.matching.use_gpu}
            metrics = extract_all_metrics(db_path, dataset, scene, kind, cfg)
            if results_dir:
                export_metrics(metrics, Path(results_dir), formats=["json", "csv"])
            return metrics
        except Exception as e:  # noqa: BLE001 - reporting must never fail the run (run_pipeline.py:267-271)
            logger.error(f"Failed to extract/export metrics: {e}")
            return None


def main() -> None:
    """Same flags as the reference CLI for the path (run_pipeline.py:426-495)."""
    ap = argparse.ArgumentParser(description="vit-colmap hot path on MI355X")
    ap.add_argument("--images", type=Path, required=True)
    ap.add_argument("--output", type=Path, required=True)
    ap.add_argument("--db", type=Path, required=True)
    ap.add_argument("--camera-model", dest="camera_model", default="SIMPLE_PINHOLE")
    ap.add_argument("--extractor", choices=["vit", "trainable_vit", "colmap_sift", "dummy"], default="vit")
    ap.add_argument("--vit-weights", dest="vit_weights", type=Path, default=None)
    ap.add_argument("--skip-matching", dest="skip_matching", action="store_true")
    ap.add_argument("--skip-reconstruction", dest="skip_reconstruction", action="store_true")
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--scene", default=None)
    ap.add_argument("--export-metrics", dest="export_metrics", type=Path, default=None)
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args()
    config = Config.from_args(args)
    logging.getLogger(__name__).info(config.summary())
    Pipeline(config).run(args.images, args.output, args.db, args.dataset, args.scene, args.export_metrics)
