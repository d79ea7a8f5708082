#!/usr/bin/env python3
"""Developer tool: ViT-B/14 token path (50 x 640x480) on the hand-written GEMMs vs the F.linear (hipBLASLt) path."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.features.vit_extractor import ViTExtractor
name = sys.argv[1] if len(sys.argv) > 1 else "dinov2_vitb14"
ex = ViTExtractor(model_name=name, num_keypoints=2048, descriptor_dim=128)
frames = torch.randint(0, 255, (50, 480, 640, 3), dtype=torch.uint8, device="cuda")
def timeit(label):
    for _ in range(3): ex._tokens(frames)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ex._tokens(frames)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    flop = 3.48e11 if "vitb" in name else 1.09e11
    print(f"{label}: {ms:.2f} ms / 50 images = {50/ms*1e3:.0f} images/s, {flop*50/ms/1e9:.0f} TFLOP/s")
timeit("hand-written GEMMs")
hip = ex.model._hip
ex.model._hip = False
timeit("F.linear (hipBLASLt default heuristic)")
