"""Extractor plugin API — identical contract to reference vit_colmap/features/base_extractor.py:6-16."""
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Optional


class BaseExtractor(ABC):
    @abstractmethod
    def extract(
        self,
        image_dir: Path,
        db_path: Path,
        camera_model: str,
        camera_params: Optional[list[float]] = None,
    ) -> None:
        """Process images in `image_dir` and write features into the COLMAP database at `db_path`."""
        raise NotImplementedError


IMAGE_EXTENSIONS = {".jpg", ".jpeg", ".png", ".bmp", ".tiff", ".tif"}  # vit_extractor.py:684


def list_images(image_dir: Path):
    """Sorted image files of a directory (vit_extractor.py:684-687, dummy_extractor.py:39-43)."""
    return sorted(f for f in Path(image_dir).iterdir() if f.suffix.lower() in IMAGE_EXTENSIONS)


def default_camera_params(camera_model: str, width: int, height: int) -> list:
    """f = max(w, h), principal point at the centre (vit_extractor.py:706-716)."""
    f = max(width, height)
    if camera_model == "SIMPLE_PINHOLE":
        return [f, width / 2.0, height / 2.0]
    if camera_model == "PINHOLE":
        return [f, f, width / 2.0, height / 2.0]
    raise ValueError(f"Unsupported camera model: {camera_model}")
