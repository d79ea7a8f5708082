#!/usr/bin/env python3
"""Developer tool: time the HIP attention kernel alone at the bench shape (standard and q_prescaled modes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit.hip_ops import attention, Q_PRESCALE
B, N, C, H = 50, 1531, 384, 6
qkv = torch.randn(B, N, 3 * C, device="cuda", dtype=torch.bfloat16)
qs = qkv.clone().reshape(B, N, 3, C)
qs[:, :, 0] = (qs[:, :, 0].float() * Q_PRESCALE).to(torch.bfloat16)
qs = qs.reshape(B, N, 3 * C).contiguous()
for name, t_in, pre in (("standard", qkv, False), ("q_prescaled (lazy max)", qs, True), ("standard", qkv, False), ("q_prescaled (lazy max)", qs, True)):
    for _ in range(3): attention(t_in, H, pre)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): attention(t_in, H, pre)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"HIP attention {name}: {t*1e3:.1f} us  {4.0*B*H*N*N*64/t/1e9:.0f} TFLOP/s")
