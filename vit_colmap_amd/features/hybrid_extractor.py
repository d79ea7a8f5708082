"""Hybrid extractor: keypoints from a classical detector, descriptors from the ViT token grid — MI355X implementation
of the descriptor half of the reference's `vit_colmap/features/hybrid_extractor.py` (class `ViTExtractor` there, never
selected by the reference's pipeline).  Same constructor arguments and the same
`_run_inference(image_bgr) -> (keypoints float32 (N, 2), descriptors uint8 (N, D))` contract.

What runs where
  host   keypoint DETECTION.  The reference uses OpenCV's SIFT / FAST / GFTT / ORB (hybrid_extractor.py:110-180); that
         stays on the host and stays OpenCV's: with cv2 importable `detector_type` selects the same detectors, without it a
         `keypoint_fn(image_bgr) -> (N, 2) float32` must be supplied (there is no cv2 in the build image).
  HIP    preprocessing, the DINOv2 forward (as ViTExtractor), and `_extract_descriptors_at_keypoints`
         (hybrid_extractor.py:224-294): bilinear sampling of the token grid at the sub-pixel keypoints, optional projection,
         RootSIFT normalisation, uint8 quantiser — csrc/select.hip `vc_describe_at`; no CPU fallback.
"""
from pathlib import Path
from typing import Callable, Optional

import numpy as np
import torch

from .. import _lib
from ..utils import image_io
from . import hip_select
from .base_extractor import BaseExtractor, default_camera_params, list_images
from .vit_extractor import PATCH, ViTExtractor


class HybridViTExtractor(BaseExtractor):
    def __init__(self, weights_path: Optional[str] = None, model_name: str = "dinov2_vitb14", num_keypoints: int = 2048,
                 descriptor_dim: int = 256, device: Optional[str] = None, detector_type: str = "sift", *,
                 keypoint_fn: Optional[Callable[[np.ndarray], np.ndarray]] = None, precision: str = "bf16",
                 projection=None, seed: int = 0):
        if detector_type not in ("sift", "fast", "gftt", "orb"):
            raise ValueError(f"Unknown detector type: {detector_type}")          # hybrid_extractor.py:130
        self.detector_type = detector_type
        self.num_keypoints = num_keypoints
        self.descriptor_dim = descriptor_dim
        self.keypoint_fn = keypoint_fn
        print(f"Initializing Hybrid extractor: {model_name}")
        print(f"Keypoint detector: {detector_type.upper()}")
        # the backbone, its preprocessing and the projection handling are ViTExtractor's
        self._vit = ViTExtractor(weights_path=weights_path, model_name=model_name, num_keypoints=num_keypoints,
                                 descriptor_dim=descriptor_dim, device=device, precision=precision, projection=projection, seed=seed)
        self.device = self._vit.device
        self.patch_size = PATCH
        if keypoint_fn is None:
            self.detector = self._create_detector()

    @property
    def descriptor_projection(self):
        return self._vit.descriptor_projection

    # ---- detection: OpenCV on the host, as in the reference ---------------------------------------------------------
    def _create_detector(self):
        try:
            import cv2  # noqa: PLC0415
        except ImportError:
            raise _lib.HipLibraryError("OpenCV is not importable: pass keypoint_fn=<callable image_bgr -> (N, 2) float32> "
                                       "(keypoint detection is host work outside the accelerated path)") from None
        if self.detector_type == "sift":
            return cv2.SIFT_create(nfeatures=self.num_keypoints)
        if self.detector_type == "fast":
            return cv2.FastFeatureDetector_create(threshold=10, nonmaxSuppression=True)
        if self.detector_type == "gftt":
            return cv2.goodFeaturesToTrack
        return cv2.ORB_create(nfeatures=self.num_keypoints)

    def _detect_keypoints(self, image_bgr: np.ndarray) -> np.ndarray:
        if self.keypoint_fn is not None:
            return np.asarray(self.keypoint_fn(image_bgr), np.float32).reshape(-1, 2)
        import cv2  # noqa: PLC0415

        gray = cv2.cvtColor(image_bgr, cv2.COLOR_BGR2GRAY)
        if self.detector_type == "gftt":
            corners = self.detector(gray, maxCorners=self.num_keypoints, qualityLevel=0.01, minDistance=7)
            return np.zeros((0, 2), np.float32) if corners is None else corners.reshape(-1, 2).astype(np.float32)
        kps = self.detector.detect(gray, None)
        kps = sorted(kps, key=lambda k: -k.response)[: self.num_keypoints]
        return np.array([k.pt for k in kps], np.float32).reshape(-1, 2)

    # ---- descriptors at the keypoints: HIP --------------------------------------------------------------------------------
    @torch.inference_mode()
    def describe_batch(self, images_bgr_np, keypoints_list):
        """Equal-size BGR uint8 arrays + their keypoints (N_i, 2) float32 in pixels -> list of uint8 (N_i, D)."""
        self._vit._require_gpu()
        h, w = images_bgr_np[0].shape[:2]
        h_new, w_new = (h // PATCH) * PATCH, (w // PATCH) * PATCH
        batch = torch.from_numpy(np.ascontiguousarray(np.stack(images_bgr_np))).to(self.device)
        tokens, hp, wp = self._vit._tokens(batch)
        kmax = max(max((len(k) for k in keypoints_list), default=0), 1)
        kp = np.zeros((len(images_bgr_np), kmax, 2), np.float32)
        cnt = np.zeros(len(images_bgr_np), np.int32)
        for i, k in enumerate(keypoints_list):
            cnt[i] = len(k)
            kp[i, : len(k)] = np.asarray(k, np.float32).reshape(-1, 2)
        C = tokens.shape[-1]
        proj = None
        if C > self.descriptor_dim:
            if self._vit.descriptor_projection is None:
                # the reference fits the projection on the first image's descriptors (hybrid_extractor.py:296-323); the
                # fit itself is ViTExtractor's (PCA when there are enough samples, seeded random projection otherwise)
                self._vit._ensure_projection(tokens, hp, wp, (w, h), (w_new, h_new))
            proj = self._vit.descriptor_projection
        u8 = hip_select.describe_at(tokens, hp, wp, torch.from_numpy(kp).to(self.device), torch.from_numpy(cnt).to(self.device),
                                    (w_new, h_new), (w, h), proj, rootsift=True).cpu().numpy()
        return [u8[i, : cnt[i]].copy() for i in range(len(images_bgr_np))]

    def _run_inference(self, image_bgr: np.ndarray):
        keypoints = self._detect_keypoints(image_bgr)
        if len(keypoints) == 0:
            print("Warning: No keypoints detected")
            D = min(self.descriptor_dim, self._vit.model.arch.dim)
            return keypoints, np.zeros((0, D), np.uint8)
        return keypoints, self.describe_batch([image_bgr], [keypoints])[0]

    def extract(self, image_dir: Path, db_path: Path, camera_model: str, camera_params: Optional[list] = None) -> None:
        """Same side effects as the reference's extract (hybrid_extractor.py:345-443): one camera, an image row per readable
        image before inference, keypoints + descriptors per image."""
        from ..database.colmap_db import Camera, ColmapDatabase

        image_files = list_images(Path(image_dir))
        if not image_files:
            raise ValueError(f"No images found in {image_dir}")
        db = ColmapDatabase(str(db_path))
        try:
            first = image_io.imread(image_files[0])
            if first is None:
                raise ValueError(f"Failed to read first image: {image_files[0]}")
            height, width = first.shape[:2]
            if camera_params is None:
                camera_params = default_camera_params(camera_model, width, height)
            cam = db.db.write_camera(Camera(model=camera_model, width=width, height=height, params=camera_params))
            for idx, f in enumerate(image_files):
                img = first if idx == 0 else image_io.imread(f)
                if img is None:
                    print(f"{f.name}: ⚠ failed to read image, skipping")
                    continue
                image_id = db.add_image(f.name, camera_id=cam)
                try:
                    kp, desc = self._run_inference(img)
                except _lib.HipLibraryError:
                    raise
                except Exception as e:  # noqa: BLE001 - one bad image never aborts the run
                    print(f"  ✗ Error during feature extraction of {f.name}: {e}")
                    continue
                if len(kp) == 0:
                    continue
                db.add_keypoints(image_id, kp)
                db.add_descriptors(image_id, desc)
            db.commit()
        finally:
            db.db.close()
