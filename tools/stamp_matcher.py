#!/usr/bin/env python3
"""Developer tool: read the per-wave cycle stamps of a -DVC_EXP_STAMP build of the persistent pair kernel
(pair2_kernel writes, per workgroup, eight totals per wave into the tail of its first pair's match block).
usage: VITCOLMAP_HIP_LIB=tools/exp/lib_stamp.so python tools/stamp_matcher.py [vit|scene] [n_images]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util_data import image_set
from vit_colmap_amd.matching import exhaustive_pairs, match_pairs, prepare_descriptors
kind = sys.argv[1] if len(sys.argv) > 1 else "vit"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 200
desc, counts = image_set(1, n_img, 512, 384, kind=kind)
dd, dc = torch.from_numpy(desc).cuda(), torch.from_numpy(counts).cuda()
pairs = exhaustive_pairs(n_img, "cuda")
P = len(pairs)
prepared = prepare_descriptors(dd, dc)
for _ in range(3):
    m, c = match_pairs(prepared, dc, n_img, 512, 384, pairs)
torch.cuda.synchronize()
m = m.cpu().numpy().view(np.uint32)
G = min(P, 256)
cid = (np.arange(G) % 8) * (G // 8) + np.arange(G) // 8 if G % 8 == 0 else np.arange(G)
lo = (cid.astype(np.int64) * P) // G
dbg = m[lo, 512 - 64:, :].reshape(G, 128)[:, :64].reshape(G, 8, 8).astype(np.float64)
npairs = dbg[:, :, 7]
names = ["wait+barrier", "mfma phase", "epilogue", "pair/pass init", "row reduce", "finalise", "total"]
print(f"{kind}, {n_img} images, {P} pairs over {G} workgroups; cycles PER PAIR (mean over workgroups)")
for half, sl in (("early waves 0-3", slice(0, 4)), ("late waves 4-7", slice(4, 8))):
    print(half)
    for i, nme in enumerate(names):
        v = dbg[:, sl, i] / npairs[:, sl]
        print(f"  {nme:16s} mean {v.mean():9.0f}   p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
