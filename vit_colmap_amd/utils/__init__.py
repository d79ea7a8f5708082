from .config import (CameraConfig, Config, ExtractorConfig, LogConfig, MatchingConfig, ReconstructionConfig)

__all__ = ["CameraConfig", "Config", "ExtractorConfig", "LogConfig", "MatchingConfig", "ReconstructionConfig"]
