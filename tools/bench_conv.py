#!/usr/bin/env python3
"""Developer tool: vc_conv_taps_bf16 vs torch (MIOpen) bf16 channels-last convolutions on the trainable extractor's head
shapes (batch 8 of 640 x 480: token grid 34 x 45, upsampled 68 x 90 and 136 x 180, heads at 120 x 160)."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit.hip_ops import conv_rows, conv_taps

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

B = int(os.environ.get("B", 8))
tot_h = tot_t = 0.0
for name, (H, W, C, N, kh, kw) in {"up1.conv 3x3": (68, 90, 512, 512, 3, 3), "up2.conv 3x3": (136, 180, 512, 512, 3, 3),
                                    "trunk 3x3": (120, 160, 512, 256, 3, 3), "heads 3x3 (64+128 -> 256)": (120, 160, 256, 256, 3, 3),
                                    "up1.deconv class 2x2": (34, 45, 384, 512, 2, 2), "up2.deconv class 2x2": (68, 90, 512, 512, 2, 2)}.items():
    xr = conv_rows(B, H, W, C, "cuda"); xr.normal_()
    w = (torch.randn(N, kh * kw * C, device="cuda") / (kh * kw * C) ** 0.5).to(torch.bfloat16)
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(B * H * W, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: conv_taps(xr, w, b, B, H, W, kh, kw, -1, -1, 1, out=out))
    img = xr[: B * H * W].reshape(B, H, W, C).permute(0, 3, 1, 2)      # channels-last view
    wt = w.reshape(N, kh, kw, C).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    tt = timeit(lambda: F.gelu(F.conv2d(img, wt, b, padding=1 if kh == 3 else 0)))
    fl = 2.0 * B * H * W * kh * kw * C * N
    mult = 4 if "deconv" in name else 1
    tot_h += t * mult; tot_t += tt * mult
    print(f"{name:28s} B={B}: hip {t*1e3:8.1f} us {fl/t/1e9:6.0f} TF/s | torch conv2d + gelu {tt*1e3:8.1f} us {fl/tt/1e9:6.0f} TF/s", flush=True)
print(f"sum (deconv classes x 4): hip {tot_h:.2f} ms, torch {tot_t:.2f} ms per {B} images")
