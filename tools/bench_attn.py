#!/usr/bin/env python3
"""Developer tool: time the HIP attention kernel alone at the bench shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit.hip_ops import attention
B, N, C, H = 50, 1531, 384, 6
qkv = torch.randn(B, N, 3 * C, device="cuda", dtype=torch.bfloat16)
for _ in range(3): attention(qkv, H)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): attention(qkv, H)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print(f"HIP attention: {t*1e3:.1f} us  {4.0*B*H*N*N*64/t/1e9:.0f} TFLOP/s")
