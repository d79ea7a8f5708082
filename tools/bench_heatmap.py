#!/usr/bin/env python3
"""Developer tool: timing of the trainable extractor's post-model path (csrc/heatmap.hip) and of the whole extractor."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.features import hip_select as hs

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

B, H, W, D = int(os.environ.get("B", 50)), 119, 157, 128
g = torch.Generator(device="cuda").manual_seed(0)
kp = torch.randn(B, 4, H, W, device="cuda", generator=g)
kp[:, 0] = torch.nn.functional.avg_pool2d(kp[:, :1] * 3, 5, 1, 2)[:, 0]
d = torch.nn.functional.normalize(torch.randn(B, D, H, W, device="cuda", generator=g), dim=1)
dcl = d.contiguous(memory_format=torch.channels_last)
for (k, thr, r) in ((2048, 0.0, 4), (512, 0.0, 4), (20480, 0.4, 1)) if os.environ.get("SEL", "1") == "1" else ():
    for name, dm in (("NCHW", d), ("NHWC", dcl)):
        t = timeit(lambda: hs.heatmap_keypoints(kp, dm, k, thr, r, (640, 480), (630, 476)))
        res = hs.heatmap_keypoints(kp, dm, k, thr, r, (640, 480), (630, 476))
        cells = B * H * W
        print(f"B={B} {H}x{W} k={k} r={r} desc {name}: {t*1e3:8.1f} us  ({cells*4*7/t/1e6:7.1f} GB/s of 7x4 B/cell; mean count {res['count'].float().mean():.0f})", flush=True)
if os.environ.get("E2E", "1") == "1":
    from vit_colmap_amd.features.trainable_vit_extractor import TrainableViTExtractor
    so, sys.stdout = sys.stdout, open(os.devnull, "w")
    for model in ("dinov2_vits14", "dinov2_vitb14"):
        ex = TrainableViTExtractor(model_name=model, num_keypoints=2048, device="cuda")
        frames = torch.randint(0, 255, (8, 480, 640, 3), dtype=torch.uint8, device="cuda")
        t0 = time.perf_counter()
        ex.extract_device(frames); torch.cuda.synchronize()
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(3): ex.extract_device(frames)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"TrainableViTExtractor {model}: {dt*1e3:.1f} ms per batch of 8 = {8/dt:.0f} images/s (first call {first:.1f} s)", file=so, flush=True)
    sys.stdout = so
