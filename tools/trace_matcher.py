#!/usr/bin/env python3
"""Developer tool: per-tile timeline of one workgroup of a -DVC_EXP_STAMP -DVC_EXP_TRACE build of pair2_kernel (data without
matches, so the workgroup's match blocks are free to hold the trace: 8 words per wave and column tile).
usage: tools/build_variants.sh trace:"-DVC_EXP_STAMP -DVC_EXP_TRACE"; VITCOLMAP_HIP_LIB=tools/exp/lib_trace.so python tools/trace_matcher.py [wg] [first_tile] [n_tiles]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util_data import image_set
from vit_colmap_amd.matching import exhaustive_pairs, match_pairs, prepare_descriptors
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
first = int(sys.argv[2]) if len(sys.argv) > 2 else 160
count = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n_img = 200
desc, counts = image_set(1, n_img, 512, 384, kind="vit")
dd, dc = torch.from_numpy(desc).cuda(), torch.from_numpy(counts).cuda()
pairs = exhaustive_pairs(n_img, "cuda")
P = len(pairs)
prepared = prepare_descriptors(dd, dc)
for _ in range(3):
    m, c = match_pairs(prepared, dc, n_img, 512, 384, pairs)
torch.cuda.synchronize()
if int(c.sum()) != 0: print("WARNING: matches were written over parts of the trace (use data without matches)")
m = m.cpu().numpy().view(np.uint32)
G = 256
lo, hi = (wg * P) // G, ((wg + 1) * P) // G
tr = m[lo:hi].reshape(-1, 8, 8).astype(np.int64)          # (tile, wave, word)
n_tiles = tr.shape[0]
t = tr[16:n_tiles - 1]
per_tile = np.diff(t[:, :, 1].min(axis=1))
print(f"workgroup range {wg}: {hi - lo} pairs, {n_tiles} tiles; cycles per tile (release to release): mean {per_tile.mean():.0f}, "
      f"p10 {np.percentile(per_tile, 10):.0f}, p50 {np.percentile(per_tile, 50):.0f}, p90 {np.percentile(per_tile, 90):.0f}")
cuts = t[:, :, 5]
print(f"tiles cut short by the early-out, per wave: {(cuts == 3).mean(axis=0).round(3)}; tiles with every wave cut: {(cuts == 3).all(axis=1).mean():.3f}")
last = t[:, :, 0].argmax(axis=1)
print("last wave to arrive at the barrier, share per wave:", np.bincount(last, minlength=8) / len(last))
print("mean arrival behind the first wave, per wave:", (t[:, :, 0] - t[:, :, 0].min(axis=1, keepdims=True)).mean(axis=0).round(0))
if t[:, :, 6].any():
    print("cycles in the vmcnt wait (arrival -> data landed), mean per wave:", (t[:, :, 6] - t[:, :, 0]).mean(axis=0).round(0))
    print("cycles from data landed to release, mean per wave:", (t[:, :, 1] - t[:, :, 6]).mean(axis=0).round(0))
print("\nper tile: wave: [arrive release | mfma begin..end | epilogue end] relative to the tile's first release; * = cut")
for i in range(first, min(first + count, n_tiles - 1)):
    r0 = tr[i, :, 1].min()
    row = []
    for w in range(8):
        a, r, mb, me, ee, cut = tr[i, w, :6]
        row.append(f"w{w}[{a - r0:5d} {r - r0:4d}|{mb - r0:5d}..{me - r0:5d}|{ee - r0:5d}]{'*' if cut == 3 else ' '}")
    print(f"tile {i:4d} (jt {i % 16:2d}): " + " ".join(row[:4]))
    print(f"                    " + " ".join(row[4:]))
