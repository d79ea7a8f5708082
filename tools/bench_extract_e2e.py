#!/usr/bin/env python3
"""Developer tool: files -> database throughput of `ViTExtractor.extract` (the plugin entry) for several directory sizes:
how much of bench.py's `extract_e2e_images_per_s` (100 files) is start-up and drain, and what the steady state is.
usage: python tools/bench_extract_e2e.py [n_files ...]   (default 100 300 1000)"""
import io, os, sys, shutil, tempfile, time, contextlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from vit_colmap_amd.features.vit_extractor import ViTExtractor
from vit_colmap_amd.utils import image_io

sizes = [int(a) for a in sys.argv[1:]] or [100, 300, 1000]
ex = ViTExtractor(model_name="dinov2_vits14", num_keypoints=512, descriptor_dim=384, device="cuda:0", precision="bf16", seed=0)
tmp = tempfile.mkdtemp(prefix="vc_e2e_")
try:
    n_max = max(sizes)
    fr = bench.synthetic_frames(0, min(n_max, 200))
    for n in sizes:
        d = os.path.join(tmp, f"images_{n}")
        os.makedirs(d)
        for k in range(n):
            image_io.imwrite(os.path.join(d, f"img_{k:04d}.png"), fr[k % len(fr)])
        best = None
        for rep in range(3):
            ex.timings = {k: 0 if k == "images" else 0.0 for k in ex.timings}
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                ex.extract(d, os.path.join(tmp, f"e2e_{n}_{rep}.db"), "SIMPLE_PINHOLE")
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best[0]):
                best = (dt, dict(ex.timings))
        print(f"{n} files: {best[0]:.3f} s = {n / best[0]:.0f} images/s; summed host seconds {best[1]}", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
