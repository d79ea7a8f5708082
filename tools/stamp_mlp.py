#!/usr/bin/env python3
"""Developer tool: per-wave cycle totals of a VC_MLP_STAMP build of the fused MLP kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit.hip_ops import FusedMlp
M, K, Hd = 50 * 1531, 384, 1536
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
mlp = FusedMlp(torch.randn(Hd, K, device="cuda") / K ** 0.5, torch.zeros(Hd, device="cuda"), torch.ones(K, device="cuda"),
               torch.zeros(K, device="cuda"), torch.randn(K, Hd, device="cuda") / Hd ** 0.5 * 0.1, torch.zeros(K, device="cuda"))
for _ in range(3):
    x.normal_()
    mlp(x)
torch.cuda.synchronize()
d = x.view(-1)[: 256 * 8 * 8 * 2].view(torch.int32).cpu().numpy().astype(np.int64).reshape(256, 8, 8)
for role, sl, names in (("A waves (fc1 + GELU)", slice(0, 4), ["wait+barrier", "issue W1 pieces", "x load + LN", "fc1 MFMAs", "GELU + hand-off", "-", "-", "total"]),
                        ("B waves (fc2)", slice(4, 8), ["wait+barrier", "issue W2 pieces", "-", "fc2 MFMAs (+G read)", "-", "tile epilogue", "-", "total"])):
    print(role)
    for i, nme in enumerate(names):
        if nme == "-": continue
        v = d[:, sl, i].astype(np.float64)
        print(f"    {nme:22s} mean {v.mean():9.0f} cycles  (per stage {v.mean()/144:6.0f})")
