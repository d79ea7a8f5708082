"""Developer diagnostic, part 2: one pair per launch, (n1, n2) swept, with the failing library variant.
usage: VITCOLMAP_HIP_LIB=tools/exp/lib_looped.so python tools/diag_looped2.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import c_oracle  # noqa: E402
from util_data import image_set  # noqa: E402
from vit_colmap_amd import _lib  # noqa: E402
from vit_colmap_amd.matching import match_pairs, prepare_descriptors  # noqa: E402

print("library:", _lib.LIB_PATH)
SENT = -77777
n_max, d = 512, 384


def run(n1, n2, n_pairs=1, kind="scene"):
    counts = np.array([n1, n2, n2, n1], np.int32)
    desc, counts = image_set(1000 + n1 + n2, 4, n_max, d, kind=kind, counts=counts, noise=0.1)
    pairs = np.array([[0, 1], [0, 2], [3, 1], [3, 2]][:n_pairs], np.int32)
    om, oc, _ = c_oracle.match_pairs(desc, counts, pairs)
    dd, dc, dp = (torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (desc, counts, pairs))
    prep = prepare_descriptors(dd, dc)
    out_counts = torch.full((len(pairs),), SENT, dtype=torch.int32, device="cuda")
    out_m = torch.full((len(pairs), n_max, 2), -1, dtype=torch.int32, device="cuda")
    m, c = match_pairs(prep, dc, 4, n_max, d, dp, out_matches=out_m, out_counts=out_counts)
    torch.cuda.synchronize()
    gc, gm = c.cpu().numpy(), m.cpu().numpy()
    ok = all(gc[p] == oc[p] and np.array_equal(gm[p, :oc[p]].view(np.uint32), om[p, :oc[p]]) for p in range(len(pairs)))
    if os.environ.get("DIAG_COUNT"):
        for w in range(8):
            tr = gm[0, 256 + w * 16: 256 + w * 16 + 16, 1]
            print(f"     wave {w}: " + " ".join(f"[jt{(v >> 24) & 63:2d} two{v & 1} d{(v >> 4) & 15} act{(v >> 12) & 1} ps{(v >> 16) & 15} cs{(v >> 20) & 15} es{(v >> 8) & 15}]" for v in tr if v != -1))
        print(f"   n1={n1} n2={n2}: barrier rounds seen by waves 0..7: {gm[0, n_max - 8:, 1].tolist()}  ok={ok} gpu={gc.tolist()} oracle={oc.tolist()}")
    return ok, gc.tolist(), oc.tolist()


if os.environ.get("DIAG_COUNT"):
    for n1, t in ((31, 8), (31, 16), (256, 16)):
        run(n1, t * 32 - 3, 1)
    sys.exit(0)
for n_pairs in (1, 2, 4):
    print(f"-- {n_pairs} pair(s) per launch; rows: n1, columns: tiles of b; '.' = equal to the oracle, 'X' = differs")
    tiles = [1, 2, 5, 6, 7, 8, 9, 10, 12, 15, 16]
    print("      " + " ".join(f"{t:3d}" for t in tiles))
    for n1 in (1, 31, 33, 64, 65, 128, 129, 256, 257, 512):
        row = []
        for t in tiles:
            ok, gc, oc = run(n1, t * 32 - 3, n_pairs)
            row.append("  ." if ok else "  X")
        print(f"{n1:5d} " + " ".join(row))
ok, gc, oc = run(1, 509, 1)
print("n1=1, 16 tiles:", ok, gc, oc)
ok, gc, oc = run(1, 509, 1, kind="vit")
print("n1=1, 16 tiles, vit rows (nothing relevant):", ok, gc, oc)
