"""Configuration dataclasses — same field names and defaults as reference
vit_colmap/utils/config.py:9-198, without the pycolmap dependency: the option objects returned by
`to_matching_options()` are the plain classes below, consumed by
vit_colmap_amd.matching.match_exhaustive."""
import logging
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class LogConfig:
    """config.py:9-25"""

    level: int = logging.INFO
    format: str = "[%(asctime)s][%(filename)s:%(lineno)d][%(levelname)s] %(message)s"
    datefmt: str = "%H:%M:%S"

    def apply(self):
        logging.basicConfig(level=self.level, format=self.format, datefmt=self.datefmt, force=True)


@dataclass
class CameraConfig:
    """config.py:28-52"""

    model: str = "SIMPLE_PINHOLE"
    width: Optional[int] = None
    height: Optional[int] = None
    params: Optional[list[float]] = None

    def get_default_params(self, width: int, height: int) -> list[float]:
        if self.params is not None:
            return self.params
        from ..features.base_extractor import default_camera_params

        try:
            return default_camera_params(self.model, width, height)
        except ValueError:
            raise ValueError(f"Unsupported camera model: {self.model}")


@dataclass
class SiftMatchingOptions:
    """The fields of pycolmap's SiftMatchingOptions that the reference sets (config.py:72-94)."""

    max_ratio: float = 0.8
    max_distance: float = 0.7
    cross_check: bool = True
    use_gpu: bool = True
    num_threads: int = -1


@dataclass
class FeatureMatchingOptions:
    """Shape of pycolmap 3.13's FeatureMatchingOptions (config.py:70-81): `.sift` sub-options."""

    use_gpu: bool = True
    num_threads: int = -1
    sift: SiftMatchingOptions = field(default_factory=SiftMatchingOptions)


@dataclass
class MatchingConfig:
    """config.py:55-96"""

    use_gpu: bool = True
    max_ratio: float = 0.8
    max_distance: float = 0.7
    cross_check: bool = True
    num_threads: int = -1  # -1 means auto-detect

    def to_matching_options(self) -> FeatureMatchingOptions:
        opts = FeatureMatchingOptions(use_gpu=self.use_gpu, num_threads=self.num_threads)
        opts.sift.max_ratio = self.max_ratio
        opts.sift.max_distance = self.max_distance
        opts.sift.cross_check = self.cross_check
        opts.sift.use_gpu = self.use_gpu
        opts.sift.num_threads = self.num_threads
        return opts

    def _to_sift_options_legacy(self) -> SiftMatchingOptions:
        return SiftMatchingOptions(max_ratio=self.max_ratio, max_distance=self.max_distance,
                                   cross_check=self.cross_check, use_gpu=self.use_gpu,
                                   num_threads=self.num_threads)


@dataclass
class ReconstructionConfig:
    """config.py:99-112.  Incremental mapping is outside the hot path; the options
    object is only built when pycolmap is importable."""

    min_num_matches: int = 15
    multiple_models: bool = True

    def to_mapper_options(self):
        import pycolmap  # noqa: PLC0415 - optional, third party

        opts = pycolmap.IncrementalPipelineOptions()
        opts.min_num_matches = self.min_num_matches
        opts.multiple_models = self.multiple_models
        return opts


@dataclass
class ExtractorConfig:
    """config.py:115-120"""

    extractor_type: str = "vit"  # "vit" or "colmap_sift"
    vit_weights_path: Optional[str] = None


@dataclass
class Config:
    """config.py:123-198"""

    log: LogConfig = field(default_factory=LogConfig)
    camera: CameraConfig = field(default_factory=CameraConfig)
    extractor: ExtractorConfig = field(default_factory=ExtractorConfig)
    matching: MatchingConfig = field(default_factory=MatchingConfig)
    reconstruction: ReconstructionConfig = field(default_factory=ReconstructionConfig)
    do_matching: bool = True
    do_reconstruction: bool = True

    def __post_init__(self):
        self.log.apply()

    @classmethod
    def from_args(cls, args):
        config = cls()
        if hasattr(args, "camera_model"):
            config.camera.model = args.camera_model
        if hasattr(args, "extractor") and args.extractor:
            config.extractor.extractor_type = args.extractor
        elif hasattr(args, "use_colmap_sift") and args.use_colmap_sift:
            config.extractor.extractor_type = "colmap_sift"
        if hasattr(args, "vit_weights") and args.vit_weights:
            config.extractor.vit_weights_path = str(args.vit_weights)
        elif hasattr(args, "model") and args.model:
            config.extractor.vit_weights_path = str(args.model)
        if hasattr(args, "use_gpu"):
            config.matching.use_gpu = args.use_gpu
        if hasattr(args, "skip_matching"):
            config.do_matching = not args.skip_matching
        if hasattr(args, "skip_reconstruction"):
            config.do_reconstruction = not args.skip_reconstruction
        if hasattr(args, "verbose") and args.verbose:
            config.log.level = logging.DEBUG
            config.log.apply()
        return config

    def summary(self) -> str:
        lines = [
            "Configuration:",
            f"  Extractor: {self.extractor.extractor_type}",
            f"  Camera model: {self.camera.model}",
            f"  Matching: {'enabled' if self.do_matching else 'disabled'}",
            f"  Reconstruction: {'enabled' if self.do_reconstruction else 'disabled'}",
            f"  GPU matching: {'enabled' if self.matching.use_gpu else 'disabled'}",
            f"  Min matches: {self.reconstruction.min_num_matches}",
        ]
        return "\n".join(lines)
