#!/usr/bin/env python3
"""Developer tool: the trainable extractor's device path in a loop, for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.features.trainable_vit_extractor import TrainableViTExtractor
so, sys.stdout = sys.stdout, open(os.devnull, "w")
ex = TrainableViTExtractor(model_name=sys.argv[1] if len(sys.argv) > 1 else "dinov2_vits14", num_keypoints=2048, device="cuda")
sys.stdout = so
frames = torch.randint(0, 255, (8, 480, 640, 3), dtype=torch.uint8, device="cuda")
for _ in range(20):
    ex.extract_device(frames)
torch.cuda.synchronize()
print("done")
