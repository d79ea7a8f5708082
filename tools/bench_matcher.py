#!/usr/bin/env python3
"""Micro-benchmark of the fused pair kernel (developer tool; bench.py is the judged entry)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util_data import image_set  # noqa: E402
from vit_colmap_amd.matching import exhaustive_pairs, match_pairs, prepare_descriptors  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=50)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--d", type=int, default=384)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--kind", default="vit")
    ap.add_argument("--pairs", default="exhaustive", help="exhaustive | sameb | samea | sorted_b")
    ap.add_argument("--limit", type=int, default=0, help="keep only the first LIMIT pairs of the list")
    a = ap.parse_args()
    desc, counts = image_set(1, a.images, a.n, a.d, kind=a.kind)
    dd, dc = torch.from_numpy(desc).cuda(), torch.from_numpy(counts).cuda()
    pairs = exhaustive_pairs(a.images, "cuda")
    if a.pairs == "sameb":      # every pair streams the same image b: B is always L2-hot
        pairs[:, 1] = 0
    elif a.pairs == "samea":
        pairs[:, 0] = 0
    elif a.pairs == "sorted_b":  # concurrent workgroups of one XCD (block id % 8) share b
        pn = pairs.cpu().numpy()
        order = np.lexsort((pn[:, 0], pn[:, 1]))
        pn = pn[order]
        n = len(pn)
        chunk = (n + 7) // 8
        idx = np.arange(n)
        src = (idx % 8) * chunk + idx // 8
        src = np.minimum(src, n - 1)
        pairs = torch.from_numpy(np.ascontiguousarray(pn[src])).cuda()
    if a.limit:
        pairs = pairs[: a.limit].contiguous()
    P = pairs.shape[0]
    prepared = prepare_descriptors(dd, dc)
    m = torch.empty((P, a.n, 2), dtype=torch.int32, device="cuda")
    c = torch.empty((P,), dtype=torch.int32, device="cuda")
    for _ in range(3):
        match_pairs(prepared, dc, a.images, a.n, a.d, pairs, out_matches=m, out_counts=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        match_pairs(prepared, dc, a.images, a.n, a.d, pairs, out_matches=m, out_counts=c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    bytes_pair = 2 * a.n * a.d + 2 * a.n * 12
    ops_pair = 2.0 * a.n * a.n * a.d
    print(f"images={a.images} N={a.n} D={a.d} pairs={P}: {ms:.3f} ms/launch, {P/ms*1e3:,.0f} pairs/s, "
          f"{P*bytes_pair/ms/1e6:.1f} GB/s algorithmic ({P*bytes_pair/ms/1e6/8000*100:.1f}% of 8 TB/s), "
          f"{P*ops_pair/ms/1e9:.1f} Tops int8 ({P*ops_pair/ms/1e9/5000*100:.1f}% of 5 Pops), matches={int(c.sum())}")
    e0.record()
    for _ in range(a.iters):
        prepare_descriptors(dd, dc)
    e1.record()
    torch.cuda.synchronize()
    print(f"prepare: {e0.elapsed_time(e1)/a.iters:.3f} ms for {a.images} images")


if __name__ == "__main__":
    main()
