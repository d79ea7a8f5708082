"""ViT feature extractor for COLMAP databases — MI355X implementation of the reference's
`ViTExtractor` (vit_colmap/features/vit_extractor.py:17-768), same constructor, same
`_run_inference(image_bgr) -> (keypoints float32 (N, 2), descriptors uint8 (N, D))` contract and
same `extract(image_dir, db_path, camera_model, camera_params)` side effects.

What runs where
  host     file listing, image decode, SQLite writes (as in the reference)
  HIP      preprocessing (resize / normalise / patchify), structure tensor, score map,
           binning + top-k + NMS, descriptor gather / projection / normalise / quantise
           — csrc/*.hip through the C ABI; there is no CPU fallback for any of it
  HIP      the whole DINOv2 ViT-S forward in bf16 (csrc/gemm.hip, csrc/attention.hip, driven by vit/dinov2.py)
  PyTorch  the GEMMs of the wider backbones (ViT-B/L/g: hipBLASLt) around the same HIP attention / LayerNorm kernels

Differences from the reference, all deliberate:
  * images are processed in batches of equal size instead of one at a time;
  * `weights_path` loads a DINOv2 state dict (the reference raises NotImplementedError,
    vit_extractor.py:88-92); with no path the reference downloads pretrained weights through
    torch.hub, which is impossible offline, so seeded random weights are used and a warning printed;
  * the PCA / random projection (vit_extractor.py:588-653) can be supplied (`projection=`) so runs
    are reproducible; if it is not, it is fitted on the first image as the reference does, but
    with a seeded generator for the random branch.
"""
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .. import _lib
from ..utils import image_io
from ..vit import build_dinov2, load_dinov2_weights
from . import hip_preprocess, hip_select
from .base_extractor import BaseExtractor, default_camera_params, list_images

PATCH = 14


class ViTExtractor(BaseExtractor):
    def __init__(
        self,
        weights_path: str | None = None,
        model_name: str = "dinov2_vitb14",
        num_keypoints: int = 2048,
        descriptor_dim: int = 128,
        device: str | None = None,
        detection_method: str = "harris",  # "harris", "dog", or "combined"
        *,
        precision: str = "bf16",            # "bf16" (MFMA) or "fp32" (reference precision)
        batch_size: int = 50,
        projection: "np.ndarray | torch.Tensor | None" = None,
        seed: int = 0,
        tune_gemm: bool = True,
    ):
        self.weights_path = weights_path
        self.model_name = model_name
        self.num_keypoints = num_keypoints
        self.descriptor_dim = descriptor_dim
        self.detection_method = detection_method
        self.batch_size = batch_size
        self.seed = seed
        if detection_method not in hip_select.METHODS:
            raise ValueError(f"Unknown detection method: {detection_method}")
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision}")
        self.dtype = torch.bfloat16 if precision == "bf16" else torch.float32

        if device is None:  # vit_extractor.py:55-58
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        else:
            self.device = torch.device(device)
        print(f"Initializing ViT extractor: {model_name} on {self.device}")

        self.model = self._load_model()
        self.model.eval()
        self.model.fold_layerscale()
        if self.device.type == "cuda" and self.dtype == torch.bfloat16:
            self.model.to(device=self.device)
            self.model.prepare_hip()      # GEMM operands from the float32 parameters (csrc/gemm.hip), ViT-S only
        self.model.to(device=self.device, dtype=self.dtype)
        self.patch_size = PATCH
        self.descriptor_projection = None  # vit_extractor.py:82
        if projection is not None:
            self.set_projection(projection)
        self.timings = {"decode_s": 0.0, "gpu_s": 0.0, "db_s": 0.0, "images": 0}
        # Wider backbones (ViT-B/L/g) still run their GEMMs on hipBLASLt, whose default heuristic is poor for these
        # shapes; TunableOp times the candidate kernels once per new GEMM shape (~1 s each, first batch only).
        # It is used ONLY for bf16 on models the hand-written GEMMs do not cover, only around the ViT forward
        # (`_tokens`), never process-wide and never for float32: round 1 recorded a SIGABRT inside an unrelated
        # float32 batched matmul while it was enabled process-wide ("TunableOp").  Results are kept in
        # memory; a CSV is written only when VITCOLMAP_TUNABLEOP_CSV names one.
        self.tune_gemm = (bool(tune_gemm) and self.device.type == "cuda" and self.dtype == torch.bfloat16
                          and not getattr(self.model, "_hip", None))
        if self.tune_gemm:
            import os
            import torch.cuda.tunable as tunable

            tunable.set_max_tuning_duration(200)
            tunable.set_max_tuning_iterations(20)
            csv = os.environ.get("VITCOLMAP_TUNABLEOP_CSV")
            if csv:
                tunable.set_filename(csv, True)   # one file per device
            else:
                tunable.set_filename(os.devnull, False)   # this torch has no "do not write" switch: results stay in memory
        print("✓ ViT model ready")

    # ------------------------------------------------------------------------------------------
    def _load_model(self):
        model = build_dinov2(self.model_name)  # ValueError for non-DINOv2 names (vit_extractor.py:100-104)
        if self.weights_path is not None:
            print(f"Loading custom weights from: {self.weights_path}")
            return load_dinov2_weights(model, self.weights_path)
        print("⚠ No weights_path given and torch.hub is unreachable offline: using seeded RANDOM weights "
              f"(seed {self.seed}). Pass a DINOv2 state dict for real features.")
        return model.init_random(self.seed)

    def set_projection(self, projection):
        p = torch.as_tensor(np.asarray(projection) if not torch.is_tensor(projection) else projection)
        p = p.to(device=self.device, dtype=torch.float32).contiguous()
        if p.dim() != 2 or p.shape[1] != self.descriptor_dim:
            raise ValueError(f"projection must have shape (C, {self.descriptor_dim}), got {tuple(p.shape)}")
        self.descriptor_projection = p

    def sync_projection(self, image_dir):
        """Multi-GPU runs: fit the projection on rank 0 (first image, as the reference does) and broadcast it, so
        that every rank projects with the same matrix."""
        from .. import dist as vd

        if not vd.is_distributed():
            return
        rank, _ = vd.rank_world()
        if rank == 0 and self.descriptor_projection is None:
            files = list_images(Path(image_dir))
            first = image_io.imread(files[0]) if files else None
            if first is not None:
                self._run_batch([first])
        have = vd.broadcast_array(np.array([self.descriptor_projection is not None], np.int32), 0, str(self.device))
        if int(have[0]):
            p = vd.broadcast_array(self.descriptor_projection.cpu().numpy() if rank == 0 else None, 0, str(self.device))
            self.set_projection(p)

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise _lib.HipLibraryError(
                "ViTExtractor needs an MI355X: the selection / descriptor path is HIP-only (no CPU fallback)")

    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def _tokens(self, images_bgr: torch.Tensor):
        """uint8 (B, h, w, 3) on the GPU -> patch tokens (B, Hp*Wp, C), Hp, Wp."""
        B, h, w, _ = images_bgr.shape
        hp, wp = h // PATCH, w // PATCH
        if getattr(self.model, "_hip", None):
            # ViT-S / B / L bf16: every GEMM of the forward is hand-written (csrc/gemm.hip); no hipBLASLt, no TunableOp
            patches = hip_preprocess.preprocess(images_bgr, out_dtype=self.dtype, layout="patches_pad")
            return self.model.forward_patch_tokens(patches, hp, wp).contiguous(), hp, wp
        patches = hip_preprocess.preprocess(images_bgr, out_dtype=self.dtype, layout="patches")
        if self.tune_gemm:
            import torch.cuda.tunable as tunable

            was = tunable.is_enabled()
            tunable.enable(True)
            try:
                tokens = self.model.forward_patch_tokens(patches, hp, wp).contiguous()
            finally:
                tunable.enable(was)
        else:
            tokens = self.model.forward_patch_tokens(patches, hp, wp).contiguous()
        return tokens, hp, wp

    @torch.inference_mode()
    def _ensure_projection(self, tokens, hp, wp, original_wh, resized_wh):
        """vit_extractor.py:601-648: fitted once, on the first image that needs it."""
        C = tokens.shape[-1]
        if C <= self.descriptor_dim or self.descriptor_projection is not None:
            return
        first = hip_select.dense_to_sparse(tokens[:1], hp, wp, original_wh, resized_wh, self.num_keypoints,
                                           self.detection_method, None, want_f32=False)
        m = int(first["count"][0].item())
        yx = first["yx"][0, :m].long()
        desc = tokens[0].float()[yx[:, 0] * wp + yx[:, 1]]        # integer grid points: plain gather
        if m > self.descriptor_dim:
            centred = desc - desc.mean(dim=0, keepdim=True)
            _, S, Vh = torch.linalg.svd(centred, full_matrices=False)
            self.descriptor_projection = Vh.T[:, : self.descriptor_dim].contiguous()
            print(f"Initialized PCA projection: {C} -> {self.descriptor_dim}")
            print(f"  Variance explained: {(S[:self.descriptor_dim].sum() / S.sum()).item():.2%}")
        else:
            g = torch.Generator(device="cpu").manual_seed(self.seed)
            p = torch.randn(C, self.descriptor_dim, generator=g, dtype=torch.float32) / np.sqrt(C)
            self.descriptor_projection = p.to(self.device)
            print(f"Initialized random projection (insufficient samples): {C} -> {self.descriptor_dim}")

    @torch.inference_mode()
    def extract_device(self, images_bgr: torch.Tensor):
        """Device-resident batch API: uint8 (B, h, w, 3) already in HBM -> dict of GPU tensors
        (keypoints (B, K, 2) float32, desc_u8 (B, K, D) uint8 zero-padded, count (B,) int32).
        This is what bench.py and the multi-GPU path call; nothing is copied to the host."""
        self._require_gpu()
        B, h, w, _ = images_bgr.shape
        h_new, w_new = (h // PATCH) * PATCH, (w // PATCH) * PATCH
        tokens, hp, wp = self._tokens(images_bgr)
        self._ensure_projection(tokens, hp, wp, (w, h), (w_new, h_new))
        proj = self.descriptor_projection if tokens.shape[-1] > self.descriptor_dim else None
        return hip_select.dense_to_sparse(tokens, hp, wp, (w, h), (w_new, h_new), self.num_keypoints,
                                          self.detection_method, proj)

    @torch.inference_mode()
    def _run_batch(self, images_bgr_np):
        """list of equal-size BGR uint8 arrays -> list of (keypoints (N, 2) float32, descriptors (N, D) uint8)."""
        self._require_gpu()
        h, w = images_bgr_np[0].shape[:2]
        h_new, w_new = (h // PATCH) * PATCH, (w // PATCH) * PATCH
        if h_new == 0 or w_new == 0:
            raise ValueError(f"image {w}x{h} is smaller than one 14x14 patch")
        batch = torch.from_numpy(np.ascontiguousarray(np.stack(images_bgr_np))).to(self.device, non_blocking=True)
        tokens, hp, wp = self._tokens(batch)
        self._ensure_projection(tokens, hp, wp, (w, h), (w_new, h_new))
        proj = self.descriptor_projection if tokens.shape[-1] > self.descriptor_dim else None
        res = hip_select.dense_to_sparse(tokens, hp, wp, (w, h), (w_new, h_new), self.num_keypoints,
                                         self.detection_method, proj)
        counts = res["count"].cpu().numpy()
        kps = res["keypoints"].cpu().numpy()
        desc = res["desc_u8"].cpu().numpy()
        return [(kps[i, : counts[i]].astype(np.float32).copy(), desc[i, : counts[i]].copy())
                for i in range(len(images_bgr_np))]

    def _run_inference(self, image_bgr: np.ndarray):
        """Single image (vit_extractor.py:106-166): keypoints (N, 2) float32 (x, y) in original-image
        pixels, descriptors (N, D) uint8."""
        return self._run_batch([image_bgr])[0]

    # ------------------------------------------------------------------------------------------
    def extract(
        self,
        image_dir: Path,
        db_path: Path,
        camera_model: str,
        camera_params: Optional[list[float]] = None,
    ):
        """vit_extractor.py:655-768."""
        import time

        from ..database.colmap_db import Camera, ColmapDatabase

        image_dir, db_path = Path(image_dir), Path(db_path)
        print(f"\n{'='*60}\nViT Feature Extraction\n{'='*60}")
        print(f"Image directory: {image_dir}\nDatabase: {db_path}\nModel: {self.model_name}")
        print(f"Target keypoints per image: {self.num_keypoints}\n{'='*60}\n")

        image_files = list_images(image_dir)
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
            camera_params = default_camera_params(camera_model, width, height)  # ValueError if unsupported
        print(f"Camera model: {camera_model}\nCamera params: {camera_params}")
        camera_id = db.db.write_camera(Camera(model=camera_model, width=width, height=height, params=camera_params))
        print(f"Camera ID: {camera_id}\n")

        # ---- batches of equal-size images, in file order ---------------------------------------
        pending = []  # (image_id, name, array)

        def flush():
            if not pending:
                return
            t0 = time.perf_counter()
            try:
                results = self._run_batch([p[2] for p in pending])
            except _lib.HipLibraryError:
                raise                          # a missing GPU / library is not a per-image problem
            except Exception:                  # isolate the failing image (vit_extractor.py:757-762)
                results = []
                for _, name, arr in pending:
                    try:
                        results.append(self._run_batch([arr])[0])
                    except Exception as e:  # noqa: BLE001
                        import traceback

                        print(f"  ✗ Error during feature extraction of {name}: {e}")
                        traceback.print_exc()
                        results.append(None)
            if self.device.type == "cuda":
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            for (image_id, name, _), r in zip(pending, results):
                if r is None:
                    continue
                keypoints, descriptors = r
                print(f"  {name}: {len(keypoints)} keypoints, descriptors {descriptors.shape}")
                if len(keypoints) == 0:
                    print("  ⚠ Warning: No keypoints extracted")
                    continue
                db.add_keypoints(image_id, keypoints)
                db.add_descriptors(image_id, descriptors)
            self.timings["gpu_s"] += t1 - t0
            self.timings["db_s"] += time.perf_counter() - t1
            self.timings["images"] += len(pending)
            pending.clear()

        for idx, img_file in enumerate(image_files, start=1):
            t0 = time.perf_counter()
            img = first_img if idx == 1 else image_io.imread(img_file)
            self.timings["decode_s"] += time.perf_counter() - t0
            if img is None:
                print(f"[{idx}/{len(image_files)}] {img_file.name}: ⚠ failed to read image, skipping")
                continue
            image_id = db.add_image(img_file.name, camera_id=camera_id)  # before inference (:739)
            if pending and (pending[0][2].shape != img.shape or len(pending) >= self.batch_size):
                flush()
            pending.append((image_id, img_file.name, img))
        flush()
        db.commit()
        print(f"\n{'='*60}\n✓ Feature extraction complete!\n{'='*60}\n")
