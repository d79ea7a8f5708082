from .dinov2 import DINOV2_ARCHS, DinoV2, build_dinov2, load_dinov2_weights

__all__ = ["DINOV2_ARCHS", "DinoV2", "build_dinov2", "load_dinov2_weights"]
