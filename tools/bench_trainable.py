#!/usr/bin/env python3
"""Developer tool: TrainableViTExtractor.extract_device throughput against the batch size."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.features.trainable_vit_extractor import TrainableViTExtractor
for model in ("dinov2_vits14", "dinov2_vitb14"):
    so, sys.stdout = sys.stdout, open(os.devnull, "w")
    ex = TrainableViTExtractor(model_name=model, num_keypoints=2048, device="cuda")
    sys.stdout = so
    for B in (8, 16, 32, 48):
        frames = torch.randint(0, 255, (B, 480, 640, 3), dtype=torch.uint8, device="cuda")
        for _ in range(2): ex.extract_device(frames)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ex.extract_device(frames)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{model} B={B}: {dt*1e3:.1f} ms per batch = {B/dt:.0f} images/s, peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
