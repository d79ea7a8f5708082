#!/usr/bin/env python3
"""Developer tool: fused MLP kernel vs the two-kernel path (xs fc1 + table GELU, staged fc2) at the bench shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit.hip_ops import FusedMlp, XsLinear, linear, gelu_table, EPI_GELU, EPI_RESIDUAL

def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

M, K, Hd = int(os.environ.get("M", 50 * 1531)), 384, 1536
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w1 = torch.randn(Hd, K, device="cuda") / K ** 0.5
b1 = torch.randn(Hd, device="cuda") * 0.1
w2 = torch.randn(K, Hd, device="cuda") / Hd ** 0.5 * 0.1
b2 = torch.randn(K, device="cuda") * 0.1
gam, bet = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
mlp = FusedMlp(w1, b1, gam, bet, w2, b2)
fc1 = XsLinear(w1, b1, gam, bet)
tab = gelu_table("cuda")
w2b, b2b = w2.to(torch.bfloat16), b2.to(torch.bfloat16)
fl = 4.0 * M * K * Hd
for _ in range(2):
    t = timeit(lambda: mlp(x))
    print(f"fused MLP: {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s", flush=True)
    def two():
        h = fc1(x, EPI_GELU, gelu_table=tab)
        linear(h, w2b, b2b, EPI_RESIDUAL, residual=x, out=x)
    t = timeit(two)
    print(f"fc1 + fc2: {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s", flush=True)
