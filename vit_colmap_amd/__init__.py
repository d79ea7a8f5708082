"""vit_colmap_amd — the vit-colmap feature-extraction + matching hot path, MI355X-native.

Sub-packages mirror the reference layout for the path only:
  features/   BaseExtractor, ViTExtractor, DummyExtractor        (reference vit_colmap/features)
  database/   ColmapDatabase over stdlib sqlite3                 (reference vit_colmap/database)
  matching/   exhaustive matcher on the HIP kernels              (replaces pycolmap.match_exhaustive)
  pipeline/   Pipeline dispatcher                                (reference vit_colmap/pipeline)
  utils/      Config dataclasses                                 (reference vit_colmap/utils/config.py)
  csrc/       HIP kernels + C ABI (include/vitcolmap_hip.h)
"""
__version__ = "0.1.0"
