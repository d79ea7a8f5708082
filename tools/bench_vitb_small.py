#!/usr/bin/env python3
"""Developer tool: ViT-B/14 token path at small batches (1, 2, 4, 8 images of 640 x 480): where the 256 x 256 GEMM tile stops paying
(`VITCOLMAP_GEMM256_MIN_FILL=<percent of CUs the large tiles must occupy>`, 0 = always the large tile)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.features.vit_extractor import ViTExtractor
so, sys.stdout = sys.stdout, open(os.devnull, "w")
ex = ViTExtractor(model_name=sys.argv[1] if len(sys.argv) > 1 else "dinov2_vitb14", num_keypoints=2048, descriptor_dim=128)
sys.stdout = so
for B in (1, 2, 4, 8, 16):
    frames = torch.randint(0, 255, (B, 480, 640, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3): ex._tokens(frames)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ex._tokens(frames)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"B={B}: {ms:.2f} ms = {B/ms*1e3:.0f} images/s", flush=True)
