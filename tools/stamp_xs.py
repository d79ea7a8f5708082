#!/usr/bin/env python3
"""Developer tool: read the per-wave cycle stamps of a VC_XS_STAMP build of the x-stationary GEMM
(VITCOLMAP_HIP_LIB=tools/exp/lib_stamp.so python tools/stamp_xs.py)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit.hip_ops import XsLinear
M, K = 50 * 1531, 384
names = ["top wait (vmcnt)", "barrier", "epilogue (waves 4-7)", "issue + x reload", "res load + bias + MFMA", "epilogue (waves 0-3)",
         "tail", "kernel total"]
for name, (N, epi, ln) in {"qkv": (1152, 0, True), "proj": (384, 2, False), "fc1": (1536, 1, True)}.items():
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 2 else None
    xs = XsLinear(w, b, torch.ones(K, device="cuda") if ln else None, torch.zeros(K, device="cuda") if ln else None)
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        xs(a, epi, r, out=o)
    torch.cuda.synchronize()
    d = o.view(-1)[: 256 * 8 * 8 * 2].view(torch.int32).cpu().numpy().astype(np.int64).reshape(256, 8, 8)
    print(f"== {name} (N={N}, epi={epi}, ln={ln})")
    if os.environ.get("STEADY") == "1":   # -DVC_XS_STAMP -DVC_XS_STAMP_STEADY build: totals from iteration 2 on, slot 6 = stages counted
        st = d[:, :, 6].astype(np.float64)
        for i, nme in enumerate(["vmcnt wait", "barrier", "issue (+ x reload at tile switches)", "bias + MFMAs (+ epilogue slices)"]):
            v = d[:, :, i].astype(np.float64) / np.maximum(st, 1)
            print(f"    per stage and wave: {nme:38s} {v.mean():7.0f} cycles")
        print(f"    per stage and wave: {'sum':38s} {(d[:, :, :4].sum(-1) / np.maximum(st, 1)).mean():7.0f} cycles")
        continue
    for half, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        print(" ", half)
        for i, nme in enumerate(names):
            v = d[:, sl, i].astype(np.float64)
            print(f"    {nme:26s} mean {v.mean():9.0f} cycles   p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
