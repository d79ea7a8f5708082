"""Developer diagnostic: run the configuration that failed with the looped two-tile body and print,
for every pair whose count differs from the oracle, what the kernel left there (sentinel = never written).
usage: VITCOLMAP_HIP_LIB=tools/exp/lib_<variant>.so python tools/diag_looped.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import c_oracle  # noqa: E402
from oracle import matcher_oracle as mo  # noqa: E402
from util_data import image_set  # noqa: E402
from vit_colmap_amd import _lib  # noqa: E402
from vit_colmap_amd.matching import match_pairs, prepare_descriptors  # noqa: E402

print("library:", _lib.LIB_PATH)
SENT = -77777
n_images, n_max, d, kind = 24, 512, 384, "scene"
rs = np.random.RandomState(n_images)
counts = rs.randint(1, n_max + 1, n_images).astype(np.int32)
counts[[3, 4, n_images - 1]] = 0
counts[[5, 6]] = [1, 32]
counts[7] = n_max
desc, counts = image_set(100 + n_images, n_images, n_max, d, kind=kind, counts=counts, noise=0.1)
pairs = mo.exhaustive_pairs(n_images)
om, oc, _ = c_oracle.match_pairs(desc, counts, pairs)
dd, dc, dp = (torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (desc, counts, pairs))
prep = prepare_descriptors(dd, dc)
for it in range(2):
    out_counts = torch.full((len(pairs),), SENT, dtype=torch.int32, device="cuda")
    out_m = torch.full((len(pairs), n_max, 2), -1, dtype=torch.int32, device="cuda")
    m, c = match_pairs(prep, dc, n_images, n_max, d, dp, out_matches=out_m, out_counts=out_counts)
    torch.cuda.synchronize()
    gc = c.cpu().numpy()
    gm = m.cpu().numpy()
    bad = np.nonzero(gc != oc)[0]
    print(f"run {it}: {len(bad)} of {len(oc)} counts differ")
    for p in bad[:40]:
        a, b = pairs[p]
        print(f"  pair {p} = ({a},{b}) n1={counts[a]} n2={counts[b]} tiles_b={(counts[b] + 31) // 32}: gpu "
              f"{'UNWRITTEN' if gc[p] == SENT else gc[p]} oracle {oc[p]}  gpu list head {gm[p, :2].tolist()} oracle {om[p, :oc[p]][:2].tolist()}")
    lists_bad = [p for p in range(len(pairs)) if gc[p] == oc[p] and not np.array_equal(gm[p, :gc[p]].view(np.uint32), om[p, :oc[p]])]
    print(f"  pairs with equal count but different lists: {lists_bad[:20]}")
