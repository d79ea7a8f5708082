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
import os
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

    # Batch shards on HIP streams (see vit/dinov2.py `_run_sharded`): at this level the WHOLE device path of a shard —
    # preprocessing, patch embedding, the block stack, the final norm, selection and descriptors — runs on the shard's
    # stream, so that the small-grid tail of one shard (selection: one workgroup per image) and the head of the next
    # step's other shard overlap with the full-chip ViT kernels of its sibling.  VITCOLMAP_VIT_SHARDS / `batch_shards`
    # as for the model (default 2, batches of >= 16 images; 1 = single stream).
    batch_shards = None

    def _shard_bounds(self, B, pipelined=False):
        # default: two shards for a call that stands alone; a call that pipelines with its neighbours (input_ready) runs its
        # batch whole — full-size kernels — on one of two alternating streams (measured: 6.80 vs 7.00 ms per 50 images)
        k = self.batch_shards if self.batch_shards is not None else int(os.environ.get("VITCOLMAP_VIT_SHARDS", "1" if pipelined else "2"))
        if k <= 1 or B < 8 * k or not getattr(self.model, "_hip", None):
            return None
        return [B * i // k for i in range(k + 1)]

    def _extract_one(self, images_bgr, hw):
        h, w, h_new, w_new = hw
        tokens, hp, wp = self._tokens(images_bgr)
        self._ensure_projection(tokens, hp, wp, (w, h), (w_new, h_new))
        proj = self.descriptor_projection if tokens.shape[-1] > self.descriptor_dim else None
        return hip_select.dense_to_sparse(tokens, hp, wp, (w, h), (w_new, h_new), self.num_keypoints,
                                          self.detection_method, proj)

    @torch.inference_mode()
    def extract_device(self, images_bgr: torch.Tensor, input_ready=None):
        """Device-resident batch API: uint8 (B, h, w, 3) already in HBM -> dict of GPU tensors
        (keypoints (B, K, 2) float32, desc_u8 (B, K, D) uint8 zero-padded, count (B,) int32).
        This is what bench.py and the multi-GPU path call; nothing is copied to the host.
        input_ready: a torch.cuda.Event after which `images_bgr` is valid.  Without it the shards start behind everything
        the caller's stream has been given so far (the only safe assumption); with it they only wait for that event and
        their own streams, so consecutive batches pipeline: batch k + 1's ViT kernels start while the caller's stream still
        holds batch k's concatenation / matching (results are always joined into the caller's stream before they are read)."""
        self._require_gpu()
        B, h, w, _ = images_bgr.shape
        hw = (h, w, (h // PATCH) * PATCH, (w // PATCH) * PATCH)
        bounds = self._shard_bounds(B, pipelined=input_ready is not None)
        needs_fit = self.model.arch.dim > self.descriptor_dim and self.descriptor_projection is None
        if needs_fit or (bounds is None and (input_ready is None or not getattr(self.model, "_hip", None))):
            return self._extract_one(images_bgr, hw)   # (the projection is fitted once, on the first image: vit_extractor.py:601-648)
        if bounds is None:
            bounds = [0, B]                   # one shard, but consecutive batches still alternate between two streams
        dev = images_bgr.device
        n_sh = len(bounds) - 1
        n_streams = max(n_sh, int(os.environ.get("VITCOLMAP_VIT_STREAMS", "2")))
        if getattr(self, "_shard_streams", None) is None or self._shard_streams[0] != (dev, n_streams):
            self._shard_streams = ((dev, n_streams), [torch.cuda.Stream(device=dev) for _ in range(n_streams)])
            self._shard_turn = 0
        cur = torch.cuda.current_stream(dev)
        side = self._shard_streams[1]
        if input_ready is None:               # shard 0 on the caller's stream, the others behind its present position
            streams = [cur] + side[1:n_sh]
            input_ready = torch.cuda.Event()
            input_ready.record(cur)
        else:                                 # every shard on a stream of its own, consecutive calls rotate through them
            streams = [side[(self._shard_turn + i) % n_streams] for i in range(n_sh)]
            self._shard_turn = (self._shard_turn + n_sh) % n_streams
        inner = self.model.batch_shards
        self.model.batch_shards = 1           # the shards are cut here, not inside the block loop
        parts = []
        try:
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    if s is not cur:
                        s.wait_event(input_ready)
                    parts.append(self._extract_one(images_bgr[bounds[i]:bounds[i + 1]], hw))
        finally:
            self.model.batch_shards = inner
        for s, part in zip(streams, parts):   # join before anything of the side streams is read on the caller's stream
            if s is cur:
                continue
            done = torch.cuda.Event()
            done.record(s)
            cur.wait_event(done)
            for t in part.values():           # allocated on s, read on cur: the allocator must not hand the block back to s early
                t.record_stream(cur)
        if len(parts) == 1:                   # (torch.cat of one tensor is a copy: six of them per batch, 10 MB the largest)
            return parts[0]
        return {k: torch.cat([p[k] for p in parts]) for k in parts[0]}

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

    # ---- asynchronous form of _run_batch: what extract() pipelines --------------------------------------------
    def _staging(self, slot, shape):
        """Pinned host staging buffer `slot` (0 / 1) for a uint8 batch of `shape` (grown, never shrunk)."""
        bufs = self.__dict__.setdefault("_pinned_in", {})
        n = int(np.prod(shape))
        if slot not in bufs or bufs[slot].numel() < n:
            bufs[slot] = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        return bufs[slot][:n].view(shape)

    @torch.inference_mode()
    def _launch_batch(self, images_bgr_np):
        """Enqueue upload + extraction + read-back of one batch of equal-size frames on the extractor's I/O stream and
        return a handle at once (no host wait)."""
        self._require_gpu()
        h, w = images_bgr_np[0].shape[:2]
        if (h // PATCH) * PATCH == 0 or (w // PATCH) * PATCH == 0:
            raise ValueError(f"image {w}x{h} is smaller than one 14x14 patch")
        if getattr(self, "_io_stream", None) is None:
            self._io_stream = torch.cuda.Stream(device=self.device)
            self._io_slot = 0
        self._io_slot ^= 1
        n = len(images_bgr_np)
        host = self._staging(self._io_slot, (n, h, w, 3))
        hv = host.numpy()
        for i, im in enumerate(images_bgr_np):
            hv[i] = im
        with torch.cuda.stream(self._io_stream):
            batch = host.to(self.device, non_blocking=True)
            res = self.extract_device(batch)
            out = {}
            for k in ("count", "keypoints", "desc_u8"):
                dst = torch.empty(res[k].shape, dtype=res[k].dtype, pin_memory=True)
                dst.copy_(res[k], non_blocking=True)
                out[k] = dst
            done = torch.cuda.Event()
            done.record(self._io_stream)
        return dict(n=n, out=out, done=done, keep=(batch, res))   # (device tensors stay referenced until the event)

    def _finish_batch(self, handle):
        """Wait for a launched batch -> list of (keypoints (N, 2) float32, descriptors (N, D) uint8)."""
        handle["done"].synchronize()
        counts = handle["out"]["count"].numpy()
        kps, desc = handle["out"]["keypoints"].numpy(), handle["out"]["desc_u8"].numpy()
        handle["keep"] = None
        return [(kps[i, : counts[i]].astype(np.float32).copy(), desc[i, : counts[i]].copy()) for i in range(handle["n"])]

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

        # ---- batches of equal-size images, in file order, three overlapped stages -----------------------------------
        # The reference's loop is serial: imread -> inference -> two DB writes per image, the GPU idle during the host
        # work (vit_extractor.py:729-755).  Here:
        #   * files are decoded by a small thread pool running ahead of the loop (PIL / OpenCV release the GIL), results
        #     consumed in file order so image ids stay those of the serial loop;
        #   * a batch is uploaded from a pinned staging buffer and runs on its own stream without a host wait; its
        #     keypoints / descriptors come back by asynchronous copies into pinned buffers behind an event;
        #   * the rows of batch k-1 are written while batch k runs (two staging buffers, one batch in flight).
        # `add_image` still precedes inference, and a failing batch is re-run image by image so that one bad image never
        # takes its neighbours with it (vit_extractor.py:739-762).
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        pending = []      # (image_id, name, array): the batch being collected
        in_flight = []    # at most one launched batch: (items, handle)

        def write_rows(items, results):
            t1 = time.perf_counter()
            for (image_id, name, _), r in zip(items, results):
                if r is None:
                    continue
                keypoints, descriptors = r
                print(f"  {name}: {len(keypoints)} keypoints, descriptors {descriptors.shape}")
                if len(keypoints) == 0:
                    print("  ⚠ Warning: No keypoints extracted")
                    continue
                db.add_keypoints(image_id, keypoints)
                db.add_descriptors(image_id, descriptors)
            self.timings["db_s"] += time.perf_counter() - t1
            self.timings["images"] += len(items)

        def one_by_one(items):                 # isolate the failing image (vit_extractor.py:757-762)
            results = []
            for _, name, arr in items:
                try:
                    results.append(self._run_batch([arr])[0])
                except _lib.HipLibraryError:
                    raise
                except Exception as e:  # noqa: BLE001
                    import traceback

                    print(f"  ✗ Error during feature extraction of {name}: {e}")
                    traceback.print_exc()
                    results.append(None)
            return results

        def finish():
            if not in_flight:
                return
            items, handle = in_flight.pop()
            t0 = time.perf_counter()
            try:
                results = self._finish_batch(handle)
            except _lib.HipLibraryError:
                raise
            except Exception:  # noqa: BLE001
                results = one_by_one(items)
            self.timings["gpu_s"] += time.perf_counter() - t0      # host time spent WAITING for the GPU
            write_rows(items, results)

        def flush():
            if not pending:
                return
            items = list(pending)
            pending.clear()
            t0 = time.perf_counter()
            try:
                handle = self._launch_batch([p[2] for p in items])   # returns without waiting for the GPU
            except _lib.HipLibraryError:
                raise                          # a missing GPU / library is not a per-image problem
            except Exception:  # noqa: BLE001
                finish()
                self.timings["gpu_s"] += time.perf_counter() - t0
                write_rows(items, one_by_one(items))
                return
            self.timings["gpu_s"] += time.perf_counter() - t0
            finish()                           # the PREVIOUS batch: its rows are written while this one runs
            in_flight.append((items, handle))

        def decode(path):
            t0 = time.perf_counter()
            img = image_io.imread(path)
            return img, time.perf_counter() - t0

        n_files = len(image_files)
        workers = max(1, min(16, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)) - 1))
        ahead = 2 * self.batch_size
        with ThreadPoolExecutor(workers) as pool:
            futures = deque()
            nxt = 1                            # index (0-based) of the next file to hand to the pool; file 0 is decoded
            while nxt < n_files and len(futures) < ahead:
                futures.append(pool.submit(decode, image_files[nxt]))
                nxt += 1
            for idx, img_file in enumerate(image_files, start=1):
                if idx == 1:
                    img = first_img
                else:
                    img, dt = futures.popleft().result()
                    self.timings["decode_s"] += dt
                    if nxt < n_files:
                        futures.append(pool.submit(decode, image_files[nxt]))
                        nxt += 1
                if img is None:
                    print(f"[{idx}/{n_files}] {img_file.name}: ⚠ failed to read image, skipping")
                    continue
                image_id = db.add_image(img_file.name, camera_id=camera_id)  # before inference (:739)
                if pending and (pending[0][2].shape != img.shape or len(pending) >= self.batch_size):
                    flush()
                pending.append((image_id, img_file.name, img))
            flush()
            finish()
        db.commit()
        print(f"\n{'='*60}\n✓ Feature extraction complete!\n{'='*60}\n")
