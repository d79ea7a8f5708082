#!/usr/bin/env python3
"""Developer tool: ViT-B/14 token path (50 x 640x480) in a loop for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.features.vit_extractor import ViTExtractor
so, sys.stdout = sys.stdout, open(os.devnull, "w")
ex = ViTExtractor(model_name="dinov2_vitb14", num_keypoints=2048, descriptor_dim=128)
sys.stdout = so
frames = torch.randint(0, 255, (50, 480, 640, 3), dtype=torch.uint8, device="cuda")
for _ in range(12):
    ex._tokens(frames)
torch.cuda.synchronize()
print("done")
