#!/usr/bin/env python3
"""Developer tool: per-phase cycle totals of a -DVC_ATTN_STAMP build of the attention kernel (bench shape).
usage: FILE=attention.hip tools/build_variants.sh attnstamp:"-DVC_ATTN_STAMP"; VITCOLMAP_HIP_LIB=tools/exp/lib_attnstamp.so python tools/stamp_attn.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit.hip_ops import attention, Q_PRESCALE
B, N, C, H = 50, 1531, 384, 6
qkv = torch.randn(B, N, 3, C, device="cuda") * 0.7
qkv[:, :, 0] *= Q_PRESCALE
qkv = qkv.reshape(B, N, 3 * C).to(torch.bfloat16).contiguous()
for _ in range(3):
    out = attention(qkv, H, True)
torch.cuda.synchronize()
o = out.view(torch.int32).reshape(B, N, H, 32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
rows = np.minimum(np.arange(0, 1536, 64), N - 1)          # first query row of every wave (QT = 2: 64 rows per wave)
d = o[:, rows][..., :8].reshape(-1, 8)
d = d[d[:, 7] == 0x5354414D]
blocks = d[:, 6].mean()
names = ["wait + barrier", "LDS-DMA issue", "acc init + QK^T MFMAs", "softmax", "PV MFMAs", "total"]
print(f"{len(d)} waves, {blocks:.0f} key blocks each; cycles per block and wave (mean, p10, p90); 32 MFMAs = 1024 pipe cycles per block")
for i, n in enumerate(names):
    v = d[:, i] / d[:, 6]
    print(f"  {n:24s} {v.mean():8.0f} {np.percentile(v,10):8.0f} {np.percentile(v,90):8.0f}")
