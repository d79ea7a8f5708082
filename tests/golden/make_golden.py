#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own post-ViT functions on CPU.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

What it does
------------
* Imports `/root/reference/vit_colmap/features/vit_extractor.py` with inert `sys.modules`
  entries for `cv2`, `torchvision`, `torchvision.transforms` (absent in this image; none of the
  functions called below touch them — SURVEY.md §8c).
* Builds a `ViTExtractor` with `__new__` (no model load: `torch.hub` is unreachable offline) and
  sets the attributes the post-ViT methods read.
* Feeds seeded synthetic feature maps (regenerated from numpy seeds by `tests/golden/cases.py`,
  never stored) through `_harris_response`, `_dog_response`, `_compute_distinctiveness`,
  `_spatial_binning_selection`, `_apply_nms`, `_extract_descriptors_interp`,
  `_reduce_descriptor_dim` (with a *stored* projection), and the full `_dense_to_sparse`.
* Runs the reference's `DummyExtractor.extract` with recording doubles at its two I/O
  boundaries (`cv2.imread` returns a blank 640x480 frame, `pycolmap.Database` records writes),
  so the Dummy keypoints/descriptors in the fixture are computed by the reference's own code.
* Writes `tests/golden/select_<case>.npz` (data only: inputs are seeds, outputs are arrays).

Ties: `torch.topk` / `torch.argsort` leave tie order unspecified, so every case asserts that the
candidate scores it orders are pairwise distinct; a case with a tie is rejected here rather than
committed.
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cases import CASES, make_feature_map, make_projection  # noqa: E402

REF = "/root/reference"
HEAD_ROWS = 16  # float descriptors: first rows only (fixtures stay small); uint8 is stored whole


def load_reference():
    for name in ("cv2", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    import importlib

    return importlib.import_module("vit_colmap.features.vit_extractor").ViTExtractor


def bare_extractor(cls, num_keypoints, descriptor_dim, method):
    ex = cls.__new__(cls)
    ex.num_keypoints = num_keypoints
    ex.descriptor_dim = descriptor_dim
    ex.detection_method = method
    ex.device = torch.device("cpu")
    ex.patch_size = 14
    ex.descriptor_projection = None
    return ex


def assert_distinct(x, what):
    x = np.asarray(x)
    if len(np.unique(x)) != len(x):
        raise SystemExit(f"tie among {what}: case must be changed")


def main():
    torch.set_num_threads(1)
    torch.manual_seed(0)
    cls = load_reference()
    for case in CASES:
        name = case["name"]
        C, H, W = case["C"], case["H"], case["W"]
        fmap_np = make_feature_map(case)
        fmap = torch.from_numpy(fmap_np)
        ex = bare_extractor(cls, case["num_keypoints"], case["descriptor_dim"], case["method"])
        out = {}
        with torch.no_grad():
            harris = ex._harris_response(fmap)
            dog = ex._dog_response(fmap)
            combined = ex._compute_distinctiveness(fmap, method="combined")
            score = ex._compute_distinctiveness(fmap, method=case["method"])
            out["harris"] = harris.numpy()
            out["dog"] = dog.numpy()
            out["combined"] = combined.numpy()
            out["score"] = score.numpy()

            coords, scores = ex._spatial_binning_selection(score, case["num_keypoints"], bin_size=16)
            assert_distinct(scores.numpy(), f"{name}: binned candidate scores")
            out["bin_coords"] = coords.numpy().astype(np.int64)
            out["bin_scores"] = scores.numpy()

            tk_coords, tk_scores = ex._simple_topk_selection(score, min(case["num_keypoints"], 64))
            assert_distinct(tk_scores.numpy(), f"{name}: simple top-k scores")
            out["topk_coords"] = tk_coords.numpy().astype(np.int64)
            out["topk_scores"] = tk_scores.numpy()

            kept, kept_scores = ex._apply_nms(coords, scores, nms_radius=1.5)
            out["nms_coords"] = kept.numpy().astype(np.int64)
            out["nms_scores"] = kept_scores.numpy()

            desc = ex._extract_descriptors_interp(fmap, kept)
            out["desc_gather_head"] = desc.numpy()[:HEAD_ROWS]

            if C > case["descriptor_dim"]:
                proj = make_projection(case)
                ex.descriptor_projection = torch.from_numpy(proj)

            w_r, h_r = W * 14, H * 14
            w_o, h_o = case["orig_wh"]
            kp, du8 = ex._dense_to_sparse(
                fmap.unsqueeze(0),
                original_size=(w_o, h_o),
                resized_size=(w_r, h_r),
                feature_grid_size=(H, W),
            )
            out["keypoints"] = kp
            out["desc_u8"] = du8
            # float descriptors right before quantisation (vit_extractor.py:239-243)
            d = desc
            if C > case["descriptor_dim"]:
                d = ex._reduce_descriptor_dim(d)
            d = torch.nn.functional.normalize(d, p=2, dim=1)
            out["desc_f32_head"] = d.numpy()[:HEAD_ROWS]
            out["desc_u8_sha256"] = np.frombuffer(
                hashlib.sha256(np.ascontiguousarray(du8).tobytes()).digest(), dtype=np.uint8
            )
        path = os.path.join(HERE, f"select_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: C={C} HxW={H}x{W} binned={len(coords)} kept={len(kept)} -> {path} "
              f"({os.path.getsize(path)/1024:.0f} KiB)")

    # ---- DummyExtractor (dummy_extractor.py:19-117), the reference's own code executed ------
    # cv2 / pycolmap are absent here, so the two I/O boundaries are replaced by recording
    # doubles: `cv2.imread` hands back a blank 640x480 frame (the extractor only reads its
    # shape, dummy_extractor.py:93) and `pycolmap.Database` records what is written to it.
    # Every keypoint / descriptor value below is computed by the reference's code.
    import tempfile
    from pathlib import Path

    recorded = {"keypoints": {}, "descriptors": {}, "images": [], "cameras": []}

    class _RecDB:
        @staticmethod
        def open(path):
            return _RecDB()

        def write_camera(self, cam):
            recorded["cameras"].append(cam)
            return len(recorded["cameras"])

        def write_image(self, img):
            recorded["images"].append(img.name)
            return len(recorded["images"])

        def write_keypoints(self, image_id, k):
            recorded["keypoints"][image_id] = np.array(k)

        def write_descriptors(self, image_id, d):
            recorded["descriptors"][image_id] = np.array(d)

    pyc = types.ModuleType("pycolmap")
    pyc.Database = _RecDB
    pyc.Camera = lambda **kw: types.SimpleNamespace(**kw)
    pyc.Image = lambda **kw: types.SimpleNamespace(**kw)
    sys.modules["pycolmap"] = pyc
    sys.modules["cv2"].imread = lambda path: np.zeros((480, 640, 3), np.uint8)
    import importlib

    dummy_mod = importlib.import_module("vit_colmap.features.dummy_extractor")
    with tempfile.TemporaryDirectory() as td:
        for i in range(2):
            (Path(td) / f"image_{i:03d}.png").write_bytes(b"")
        dummy_mod.DummyExtractor(step=32).extract(Path(td), Path(td) / "db.db", "PINHOLE")
    assert recorded["images"] == ["image_000.png", "image_001.png"]
    assert np.array_equal(recorded["descriptors"][1], recorded["descriptors"][2])
    cam = recorded["cameras"][0]
    np.savez_compressed(
        os.path.join(HERE, "dummy_640x480.npz"),
        keypoints=recorded["keypoints"][1],
        descriptors=recorded["descriptors"][1],
        camera_params=np.array(cam.params, dtype=np.float64),
        camera_wh=np.array([cam.width, cam.height]),
    )
    print("dummy:", recorded["keypoints"][1].shape, recorded["descriptors"][1].shape, cam)


if __name__ == "__main__":
    main()
