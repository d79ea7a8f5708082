#!/usr/bin/env python3
"""Developer tool: read the per-wave cycle stamps of a VC_EXP_STAMP build of the pair kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util_data import image_set
from vit_colmap_amd.matching import exhaustive_pairs, match_pairs, prepare_descriptors
kind = sys.argv[1] if len(sys.argv) > 1 else "vit"
n_img = 200
desc, counts = image_set(1, n_img, 512, 384, kind=kind)
dd, dc = torch.from_numpy(desc).cuda(), torch.from_numpy(counts).cuda()
pairs = exhaustive_pairs(n_img, "cuda")
prepared = prepare_descriptors(dd, dc)
for _ in range(3):
    m, c = match_pairs(prepared, dc, n_img, 512, 384, pairs)
torch.cuda.synchronize()
m = m.cpu().numpy().view(np.uint32)
dbg = m[:, 512 - 64:, :].reshape(len(pairs), 128)[:, :64].reshape(len(pairs), 8, 8)
names = ["wait+barrier+stage", "mfma", "epilogue", "loop total", "prologue", "kernel total"]
for half, sl in (("early waves 0-3", slice(0, 4)), ("late waves 4-7", slice(4, 8))):
    print(half)
    for i, nme in enumerate(names):
        v = dbg[:, sl, i].astype(np.float64)
        print(f"  {nme:20s} mean {v.mean():9.0f} cycles   p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
