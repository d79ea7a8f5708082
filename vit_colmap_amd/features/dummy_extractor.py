"""Deterministic grid keypoints + position-seeded random 128-D uint8 descriptors.

Same behaviour as reference vit_colmap/features/dummy_extractor.py:8-117 (BASELINE config 1).
This is host-side test scaffolding in the reference too: there is nothing to put on the GPU.
The descriptor of a keypoint depends only on its grid cell, so the table is built once per
image size instead of constructing 300 RandomState objects per image.
"""
from pathlib import Path
from typing import Optional

import numpy as np

from ..utils import image_io
from .base_extractor import BaseExtractor, default_camera_params, list_images


class DummyExtractor(BaseExtractor):
    def __init__(self, step: int = 32, seed: int = 42):
        self.step = step
        self.seed = seed
        self._cache = {}

    def features_for(self, height: int, width: int):
        """(keypoints float32 (N, 2) [x, y], descriptors uint8 (N, 128)) — dummy_extractor.py:95-111."""
        key = (height, width)
        if key not in self._cache:
            step = self.step
            ys = np.arange(step // 2, height, step, dtype=np.float32)
            xs = np.arange(step // 2, width, step, dtype=np.float32)
            kpts = np.empty((len(ys) * len(xs), 2), np.float32)
            kpts[:, 0] = np.tile(xs, len(ys))
            kpts[:, 1] = np.repeat(ys, len(xs))
            desc = np.empty((len(kpts), 128), np.uint8)
            for r in range(len(kpts)):
                local_seed = self.seed + int(kpts[r, 0] / step) * 1000 + int(kpts[r, 1] / step)
                desc[r] = np.random.RandomState(local_seed).randint(0, 256, size=128, dtype=np.uint8)
            self._cache[key] = (kpts, desc)
        return self._cache[key]

    def extract(self, image_dir: Path, db_path: Path, camera_model: str,
                camera_params: Optional[list[float]] = None) -> None:
        from ..database.colmap_db import Camera, ColmapDatabase

        image_dir = Path(image_dir)
        image_files = list_images(image_dir) if image_dir.exists() else []
        if not image_files:  # dummy_extractor.py:46-55
            print(f"No images found in {image_dir}, generating 10 dummy images...")
            image_dir.mkdir(parents=True, exist_ok=True)
            for i in range(10):
                img = np.random.randint(0, 256, (480, 640, 3), dtype=np.uint8)
                img_path = image_dir / f"dummy_{i:03d}.png"
                image_io.imwrite(img_path, img)
                image_files.append(img_path)

        db = ColmapDatabase(str(db_path))
        first_img = image_io.imread(image_files[0])
        if first_img is None:  # dummy_extractor.py:59-61
            return
        height, width = first_img.shape[:2]
        if camera_params is None:
            camera_params = default_camera_params(camera_model, width, height)
        camera_id = db.db.write_camera(Camera(model=camera_model, width=width, height=height, params=camera_params))

        for img_file in image_files:
            img = image_io.imread(img_file)
            if img is None:
                continue
            image_id = db.add_image(img_file.name, camera_id=camera_id)
            h, w = img.shape[:2]
            kpts, desc = self.features_for(h, w)
            db.add_keypoints(image_id, kpts)
            db.add_descriptors(image_id, desc)
        db.commit()
