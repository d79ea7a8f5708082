#!/usr/bin/env python3
"""Developer tool: csrc/gemm.hip vs torch F.linear (hipBLASLt) on the ViT-S shapes at B=50, N=1531."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit.hip_ops import linear, XsLinear, gelu_table

def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

M = int(os.environ.get("M", 50 * 1531))
for name, (K, N, epi) in {"qkv": (384, 1152, 0), "proj": (384, 384, 2), "fc1": (384, 1536, 1), "fc2": (1536, 384, 2)}.items():
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 2 else None
    fl = 2.0 * M * K * N
    if os.environ.get("XSONLY") == "1":
        for ln in (False, True):
            if K != 384: continue
            xs = XsLinear(w.float(), b.float(), torch.ones(K, device="cuda") if ln else None, torch.zeros(K, device="cuda") if ln else None)
            o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            tx = timeit(lambda: xs(a, epi, r, out=o))
            print(f"{name:5s} xs ln={int(ln)}: {tx*1e3:7.1f} us {fl/tx/1e9:6.0f} TF/s", flush=True)
            if epi == 1:
                tab = gelu_table("cuda")
                tx = timeit(lambda: xs(a, epi, r, out=o, gelu_table=tab))
                print(f"{name:5s} xs ln={int(ln)} GELU table: {tx*1e3:7.1f} us {fl/tx/1e9:6.0f} TF/s", flush=True)
        continue
    t_ref = timeit(lambda: F.linear(a, w, b))
    if epi == 1:
        t_full = timeit(lambda: F.gelu(F.linear(a, w, b)))
    elif epi == 2:
        t_full = timeit(lambda: r + F.linear(a, w, b))
    else:
        t_full = t_ref
    t = timeit(lambda: linear(a, w, b, epi, r))
    if K == 384:
        for ln in (False, True):
            xs = XsLinear(w.float(), b.float(), torch.ones(K, device="cuda") if ln else None, torch.zeros(K, device="cuda") if ln else None)
            o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            tx = timeit(lambda: xs(a, epi, r, out=o))
            print(f"{name:5s} xs ln={int(ln)}: {tx*1e3:7.1f} us {fl/tx/1e9:6.0f} TF/s", flush=True)
    print(f"{name:5s} M={M} K={K} N={N} epi={epi}: hip {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s | torch gemm {t_ref*1e3:7.1f} us "
          f"{fl/t_ref/1e9:6.0f} TF/s, with unfused epilogue {t_full*1e3:7.1f} us", flush=True)
