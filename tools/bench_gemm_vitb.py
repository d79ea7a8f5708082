#!/usr/bin/env python3
"""Developer tool: the four Linear shapes of ViT-B/14 at 50 x 1531 rows: csrc/gemm.hip with the 256 x 256 tile against
the 128 x 128 tile (VITCOLMAP_GEMM_TILE=128, read once per process: run twice) and F.linear (hipBLASLt default heuristic)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit.hip_ops import linear


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


M = int(os.environ.get("M", 50 * 1531))
dim = int(os.environ.get("DIM", 768))
tot_h = tot_l = 0.0
for name, (K, N, epi) in {"qkv": (dim, 3 * dim, 0), "proj": (dim, dim, 2), "fc1": (dim, 4 * dim, 1), "fc2": (4 * dim, dim, 2)}.items():
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 2 else None
    fl = 2.0 * M * K * N
    t = timeit(lambda: linear(a, w, b, epi, r))
    t_lib = timeit(lambda: F.linear(a, w, b))
    tot_h += t; tot_l += t_lib
    print(f"{name:5s} {M}x{K}x{N}: hand-written (tile {os.environ.get('VITCOLMAP_GEMM_TILE', '256')}) {t*1e3:7.1f} us {fl/t/1e9:5.0f} TF/s | "
          f"F.linear alone (no epilogue) {t_lib*1e3:7.1f} us {fl/t_lib/1e9:5.0f} TF/s", flush=True)
print(f"sum per layer: hand-written {tot_h*1e3:.0f} us, F.linear {tot_l*1e3:.0f} us")
