from .colmap_db import Camera, ColmapDatabase, Image, SqliteColmapDatabase, pair_id_of, pair_id_to_image_ids

__all__ = ["Camera", "ColmapDatabase", "Image", "SqliteColmapDatabase", "pair_id_of", "pair_id_to_image_ids"]
