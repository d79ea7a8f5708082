"""Trainable ViT feature extractor for COLMAP databases — MI355X implementation of the reference's
`TrainableViTExtractor` (vit_colmap/features/trainable_vit_extractor.py:18-392): same constructor arguments, same
`_run_inference(image_bgr) -> (keypoints float32 (N, 6) = x, y, scale, orientation, score, 0; descriptors uint8 (N, D))`
contract and the same `extract(image_dir, db_path, camera_model, camera_params)` side effects.

What runs where
  host     file listing, image decode, SQLite writes (as in the reference)
  HIP      preprocessing (resize to a multiple of 14 / normalise / patchify: csrc/preprocess.hip), the DINOv2 backbone
           (vit/dinov2.py), and everything after the model — sigmoid, max-pool NMS, threshold, top-k, sub-pixel
           keypoints, descriptor gather + quantiser (csrc/heatmap.hip, `vc_heatmap_keypoints`); no CPU fallback
  PyTorch  the convolutional upsampler / trunk / heads (MIOpen), channels-last on the backbone's token grid

Throughput (1x MI355X, 640x480, batches of 8, `tools/bench_heatmap.py`): 830 images/s with the ViT-S backbone, 670 with ViT-B
(bf16; 307 / 274 with float32 heads); the convolutional heads are ~85 % of that time, the post-model HIP path 0.3 ms per 50
images.  The first batch of a new image size runs MIOpen's solver search for the seven convolution shapes (~30 s, cached by
MIOpen per user afterwards); `MIOPEN_FIND_MODE=2` skips the search at 5x lower head throughput.

Differences from the reference, deliberate: images are processed in batches of equal size; without `weights_path` the
reference downloads the pretrained backbone through torch.hub and leaves the heads at torch's default initialisation —
offline, backbone AND heads are seeded random and a warning is printed; checkpoints are read with
`torch.load(weights_only=True)` or safetensors (the reference unpickles)."""
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .. import _lib
from ..model import ViTFeatureModel
from ..utils import image_io
from . import hip_preprocess, hip_select
from .base_extractor import BaseExtractor, list_images

PATCH = 14


def _default_camera_params(camera_model: str, width: int, height: int) -> list:
    """trainable_vit_extractor.py:323-340 (two more models than ViTExtractor)."""
    f = max(width, height)
    if camera_model == "SIMPLE_PINHOLE":
        return [f, width / 2.0, height / 2.0]
    if camera_model == "PINHOLE":
        return [f, f, width / 2.0, height / 2.0]
    if camera_model == "SIMPLE_RADIAL":
        return [f, width / 2.0, height / 2.0, 0.0]
    if camera_model == "RADIAL":
        return [f, width / 2.0, height / 2.0, 0.0, 0.0]
    raise ValueError(f"Unsupported camera model: {camera_model}")


class TrainableViTExtractor(BaseExtractor):
    def __init__(
        self,
        weights_path: Optional[str] = None,
        model_name: str = "dinov2_vitb14",
        num_keypoints: int = 2048,
        descriptor_dim: int = 128,
        device: Optional[str] = None,
        score_threshold: float = 0.0,
        nms_radius: int = 4,
        *,
        precision: str = "bf16",     # "bf16": backbone and convolutional heads on the matrix cores; "fp32": the reference's precision
        batch_size: int = 32,        # images per device batch (4-6 GiB of head activations at 640 x 480; 8 leaves the backbone's persistent kernels half empty)
        seed: int = 0,
    ):
        self.weights_path = weights_path
        self.model_name = model_name
        self.num_keypoints = num_keypoints
        self.descriptor_dim = descriptor_dim
        self.score_threshold = score_threshold
        self.nms_radius = nms_radius
        self.batch_size = batch_size
        self.seed = seed
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision}")
        self.dtype = torch.bfloat16 if precision == "bf16" else torch.float32
        if device is None:  # trainable_vit_extractor.py:56-59
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        else:
            self.device = torch.device(device)
        print(f"Initializing Trainable ViT extractor: {model_name} on {self.device}")
        self.model = self._load_model()
        self.model.eval()
        self.model.backbone.fold_layerscale()
        if self.device.type == "cuda" and self.dtype == torch.bfloat16:
            self.model.to(self.device)
            self.model.backbone.prepare_hip()          # hand-written GEMM operands (ViT-S only; no-op otherwise)
        self.model.fold_batchnorm()                    # float32, before any cast
        self.model.to(self.device)
        self.model.to(dtype=self.dtype)                # heads too: MIOpen bf16 convolutions; their outputs are read back as float32
        if self.device.type == "cuda":
            self.model.upsampler.to(memory_format=torch.channels_last)
            self.model.trunk.to(memory_format=torch.channels_last)
            self.model.keypoint_head.to(memory_format=torch.channels_last)
            self.model.descriptor_head.to(memory_format=torch.channels_last)
            if self.dtype == torch.bfloat16:
                self.model.prepare_hip_heads()         # upsampler / trunk / heads on vc_conv_taps_bf16 instead of MIOpen
        self.patch_size = self.model.patch_size
        counts = self.model.count_parameters()
        print("✓ Model loaded successfully")
        print(f"  Total parameters: {counts['total']:,}")
        print(f"  Trainable parameters: {counts['trainable']:,}")
        print(f"  Frozen parameters: {counts['frozen']:,}")

    def _load_model(self) -> ViTFeatureModel:
        """trainable_vit_extractor.py:89-112."""
        model = ViTFeatureModel(backbone_name=self.model_name, descriptor_dim=self.descriptor_dim, freeze_backbone=True,
                                seed=self.seed)
        if self.weights_path is None:
            print("⚠ No weights_path given and torch.hub is unreachable offline: backbone and heads use seeded RANDOM "
                  f"weights (seed {self.seed}).")
            return model
        print(f"Loading custom weights from: {self.weights_path}")
        if str(self.weights_path).endswith(".safetensors"):
            from safetensors.torch import load_file

            checkpoint = load_file(str(self.weights_path))
        else:
            checkpoint = torch.load(str(self.weights_path), map_location="cpu", weights_only=True)
        if isinstance(checkpoint, dict) and "model_state_dict" in checkpoint:      # :101-106
            state_dict = checkpoint["model_state_dict"]
        elif isinstance(checkpoint, dict) and "state_dict" in checkpoint:
            state_dict = checkpoint["state_dict"]
        else:
            state_dict = checkpoint
        model.load_reference_state_dict(state_dict)
        print("✓ Custom weights loaded")
        return model

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise _lib.HipLibraryError(
                "TrainableViTExtractor needs an MI355X: preprocessing and keypoint selection are HIP-only (no CPU fallback)")

    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def head_maps(self, images_bgr: torch.Tensor):
        """uint8 (B, h, w, 3) on the GPU -> (keypoints map (B, 4, H/4, W/4), descriptor map (B, D, H/4, W/4)) float32."""
        B, h, w, _ = images_bgr.shape
        hp, wp = h // PATCH, w // PATCH
        layout = "patches_pad" if getattr(self.model.backbone, "_hip", None) else "patches"
        patches = hip_preprocess.preprocess(images_bgr, out_dtype=self.dtype, layout=layout)
        tokens = self.model.backbone.forward_patch_tokens(patches, hp, wp).contiguous()
        feats = ViTFeatureModel.tokens_to_grid(tokens, hp, wp)                 # channels-last view of the token grid
        out = self.model.forward_from_backbone_features(feats, target_size=((hp * PATCH) // 4, (wp * PATCH) // 4))
        return out["keypoints"], out["descriptors"]

    @torch.inference_mode()
    def extract_device(self, images_bgr: torch.Tensor):
        """Device-resident batch API: dict of GPU tensors (keypoints (B, K, 6), desc_u8 (B, K, D) zero padded, count (B,))."""
        self._require_gpu()
        B, h, w, _ = images_bgr.shape
        h_new, w_new = (h // PATCH) * PATCH, (w // PATCH) * PATCH
        if h_new == 0 or w_new == 0:
            raise ValueError(f"image {w}x{h} is smaller than one 14x14 patch")
        kp_map, d_map = self.head_maps(images_bgr)
        return hip_select.heatmap_keypoints(kp_map, d_map, self.num_keypoints, self.score_threshold, self.nms_radius,
                                            (w, h), (w_new, h_new))

    def _run_batch(self, images_bgr_np):
        self._require_gpu()
        batch = torch.from_numpy(np.ascontiguousarray(np.stack(images_bgr_np))).to(self.device, non_blocking=True)
        res = self.extract_device(batch)
        counts = res["count"].cpu().numpy()
        kps = res["keypoints"].cpu().numpy()
        desc = res["desc_u8"].cpu().numpy()
        return [(kps[i, : counts[i]].astype(np.float32).copy(), desc[i, : counts[i]].copy()) for i in range(len(images_bgr_np))]

    def _run_inference(self, image_bgr: np.ndarray):
        """Single image (trainable_vit_extractor.py:139-269): keypoints (N, 6) float32, descriptors (N, D) uint8;
        (0, 6) / (0, D) arrays when nothing passes the threshold (:195-200)."""
        return self._run_batch([image_bgr])[0]

    # ------------------------------------------------------------------------------------------
    def extract(self, image_dir: Path, db_path: Path, camera_model: str, camera_params: Optional[list[float]] = None):
        """trainable_vit_extractor.py:271-392."""
        from ..database.colmap_db import Camera, ColmapDatabase

        image_dir, db_path = Path(image_dir), Path(db_path)
        print(f"\n{'='*60}\nTrainable ViT Feature Extraction\n{'='*60}")
        print(f"Image directory: {image_dir}\nDatabase: {db_path}\nModel: {self.model_name}")
        print(f"Target keypoints per image: {self.num_keypoints}\nScore threshold: {self.score_threshold}")
        print(f"NMS radius: {self.nms_radius}\n{'='*60}\n")
        image_files = list_images(image_dir) + sorted(f for f in image_dir.iterdir() if f.suffix.lower() == ".ppm")  # :300
        image_files = sorted(image_files)
        if not image_files:
            raise ValueError(f"No images found in {image_dir}")
        print(f"Found {len(image_files)} images")
        db = ColmapDatabase(str(db_path))
        first_img = image_io.imread(image_files[0])
        if first_img is None:
            raise ValueError(f"Failed to read first image: {image_files[0]}")
        height, width = first_img.shape[:2]
        print(f"Image dimensions: {width}x{height}")
        if camera_params is None:
            camera_params = _default_camera_params(camera_model, width, height)
        print(f"Camera model: {camera_model}\nCamera params: {camera_params}")
        camera_id = db.db.write_camera(Camera(model=camera_model, width=width, height=height, params=camera_params))
        print(f"Camera ID: {camera_id}\n")

        pending = []  # (image_id, name, array): equal-size images, in file order

        def flush():
            if not pending:
                return
            try:
                results = self._run_batch([p[2] for p in pending])
            except _lib.HipLibraryError:
                raise
            except Exception:                          # isolate the failing image (:380-385)
                results = []
                for _, name, arr in pending:
                    try:
                        results.append(self._run_batch([arr])[0])
                    except Exception as e:  # noqa: BLE001
                        import traceback

                        print(f"  Error during feature extraction of {name}: {e}")
                        traceback.print_exc()
                        results.append(None)
            for (image_id, name, _), r in zip(pending, results):
                if r is None:
                    continue
                keypoints, descriptors = r
                print(f"  {name}: extracted {len(keypoints)} keypoints, descriptor shape {descriptors.shape}")
                if len(keypoints) == 0:
                    print("  Warning: No keypoints extracted")              # :371-373: nothing is written
                    continue
                scores = keypoints[:, 4]
                print(f"  Score range: [{scores.min():.3f}, {scores.max():.3f}]")
                db.add_keypoints(image_id, keypoints)
                db.add_descriptors(image_id, descriptors)
            pending.clear()

        # files are decoded ahead of the loop by a thread pool (the reference's loop is serial: imread -> inference -> write,
        # :340-390); results are consumed in file order, so image ids are those of the serial loop
        import os
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        workers = max(1, min(16, len(os.sched_getaffinity(0)) - 1 if hasattr(os, "sched_getaffinity") else 4))
        window = max(2 * self.batch_size, 2 * workers)
        pool = ThreadPoolExecutor(max_workers=workers)
        ahead = deque()
        files_iter = iter(image_files[1:])

        def refill():
            while len(ahead) < window:
                f = next(files_iter, None)
                if f is None:
                    return
                ahead.append(pool.submit(image_io.imread, f))

        try:
            refill()
            for idx, img_file in enumerate(image_files, start=1):
                if idx == 1:
                    img = first_img
                else:
                    img = ahead.popleft().result()
                    refill()
                if img is None:
                    print(f"[{idx}/{len(image_files)}] {img_file.name}: Warning: Failed to read image, skipping")
                    continue
                image_id = db.add_image(img_file.name, camera_id=camera_id)      # before inference (:358)
                if pending and (pending[0][2].shape != img.shape or len(pending) >= self.batch_size):
                    flush()
                pending.append((image_id, img_file.name, img))
            flush()
        finally:
            pool.shutdown(wait=False, cancel_futures=True)
        db.commit()
        print(f"\n{'='*60}\nFeature extraction complete!\n{'='*60}\n")
