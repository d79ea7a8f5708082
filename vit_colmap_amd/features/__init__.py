"""Feature extraction modules (reference vit_colmap/features/__init__.py)."""
from .base_extractor import BaseExtractor
from .dummy_extractor import DummyExtractor

__all__ = ["BaseExtractor", "DummyExtractor"]
