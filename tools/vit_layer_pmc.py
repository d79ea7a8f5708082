#!/usr/bin/env python3
"""Developer tool: run the ViT-S bf16 forward (50 images of 34x45 patches, random weights) a few times, for PMC collection
(rocprofv3 --kernel-trace --pmc ... -- python3 tools/vit_layer_pmc.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit import build_dinov2

m = build_dinov2("dinov2_vits14").init_random(seed=1).eval().fold_layerscale().to("cuda")
m.prepare_hip()
m.to(torch.bfloat16)
g = torch.Generator(device="cuda").manual_seed(0)
patches = torch.zeros(50, 34 * 45, 640, device="cuda", dtype=torch.bfloat16)
patches[..., :588] = torch.randn(50, 34 * 45, 588, device="cuda", generator=g).to(torch.bfloat16)
with torch.inference_mode():
    for _ in range(int(os.environ.get("ITERS", "3"))):
        out = m.forward_patch_tokens(patches, 34, 45)
torch.cuda.synchronize()
print("ok", tuple(out.shape), float(out.float().abs().mean()))
