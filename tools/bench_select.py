#!/usr/bin/env python3
"""Developer tool: the selection kernels alone (structure tensor, score map, select, describe) on the bench's token grid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.features import hip_select as hs

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

B, H, W, C = 50, 34, 45, 384
g = torch.Generator(device="cuda").manual_seed(0)
tokens = torch.randn(B, H * W, C, device="cuda", generator=g).to(torch.bfloat16)
score = torch.rand(B, H, W, device="cuda", generator=g)
score = torch.nn.functional.avg_pool2d(score[:, None], 3, 1, 1)[:, 0].contiguous()      # smooth, like a real score map
print(f"select_keypoints (50 maps of {H}x{W}, target 512): {timeit(lambda: hs.select_keypoints(score, 512)):.1f} us")
print(f"dense_to_sparse  (whole tail from tokens):          {timeit(lambda: hs.dense_to_sparse(tokens, H, W, (640, 480), (630, 476), 512, 'combined', None)):.1f} us")
