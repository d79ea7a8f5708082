from .config import (CameraConfig, Config, ExtractorConfig, LogConfig, MatchingConfig, ReconstructionConfig)

__all__ = ["CameraConfig", "Config", "ExtractorConfig", "LogConfig", "MatchingConfig", "ReconstructionConfig"]
from .metrics import FeatureMetrics, MatchingMetrics, MetricsExtractor  # noqa: E402,F401

__all__ += ["FeatureMetrics", "MatchingMetrics", "MetricsExtractor"]
