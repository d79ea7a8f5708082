"""Trainable models (reference vit_colmap/model/__init__.py)."""
from .vit_feature_model import UpsampleBlock, ViTFeatureModel

__all__ = ["ViTFeatureModel", "UpsampleBlock"]
