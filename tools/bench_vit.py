#!/usr/bin/env python3
"""Developer tool: time the pieces of the ViT-S bf16 forward at the bench shape (B=50, N=1531)."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

B, N, C, H = 50, 1531, 384, 6
dev = "cuda"
x = torch.randn(B, N, C, device=dev, dtype=torch.bfloat16)
qkv = torch.randn(B, N, 3, H, C // H, device=dev, dtype=torch.bfloat16).permute(2, 0, 3, 1, 4)
q, k, v = qkv[0], qkv[1], qkv[2]
flops_attn = 4.0 * B * H * N * N * (C // H)
print("sdpa backends available:", [n for n in dir(torch.backends.cuda) if "sdp" in n or "fa_" in n])
t = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
print(f"sdpa default: {t*1e3:.1f} us  {flops_attn/t/1e9:.0f} TFLOP/s")
try:
    torch.backends.cuda.preferred_rocm_fa_library("ck")
    t = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
    print(f"sdpa ck: {t*1e3:.1f} us  {flops_attn/t/1e9:.0f} TFLOP/s")
    torch.backends.cuda.preferred_rocm_fa_library("aotriton")
except Exception as e:
    print("ck fa not available:", str(e)[:200])
qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
t = timeit(lambda: F.scaled_dot_product_attention(qc, kc, vc))
print(f"sdpa contiguous q,k,v: {t*1e3:.1f} us  {flops_attn/t/1e9:.0f} TFLOP/s")
from torch.nn.attention import SDPBackend, sdpa_kernel
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
    try:
        with sdpa_kernel(be):
            t = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
        print(f"sdpa {be}: {t*1e3:.1f} us  {flops_attn/t/1e9:.0f} TFLOP/s")
    except Exception as e:
        print(be, "failed:", str(e)[:100])
for name, (K, Nn) in {"qkv": (384, 1152), "proj": (384, 384), "fc1": (384, 1536), "fc2": (1536, 384)}.items():
    a = torch.randn(B * N, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(Nn, K, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(Nn, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: F.linear(a, w, bias))
    fl = 2.0 * B * N * K * Nn
    byt = 2.0 * (B * N * K + K * Nn + B * N * Nn)
    print(f"gemm {name} ({B*N}x{K}x{Nn}): {t*1e3:.1f} us  {fl/t/1e9:.0f} TFLOP/s  {byt/t/1e6:.0f} GB/s")
if os.environ.get("TUNE") == "1":
    import torch.cuda.tunable as tn
    tn.enable(True); tn.set_max_tuning_duration(200); tn.set_max_tuning_iterations(20)
    for name, (K, Nn) in {"qkv": (384, 1152), "proj": (384, 384), "fc1": (384, 1536), "fc2": (1536, 384)}.items():
        a = torch.randn(B * N, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(Nn, K, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(Nn, device=dev, dtype=torch.bfloat16)
        t0 = time.time(); F.linear(a, w, bias); torch.cuda.synchronize(); tt = time.time() - t0
        t = timeit(lambda: F.linear(a, w, bias))
        fl = 2.0 * B * N * K * Nn
        print(f"TUNED gemm {name}: {t*1e3:.1f} us  {fl/t/1e9:.0f} TFLOP/s (tuning took {tt:.1f}s)")
h = torch.randn(B * N, 1536, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: F.gelu(h)); print(f"gelu: {t*1e3:.1f} us")
try:
    from vit_colmap_amd.vit.hip_ops import attention
    qkv_c = torch.randn(B, N, 3 * C, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: attention(qkv_c, H))
    print(f"HIP attention: {t*1e3:.1f} us  {flops_attn/t/1e9:.0f} TFLOP/s")
except Exception as e:
    print("hip attention failed:", e)
