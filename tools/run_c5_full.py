#!/usr/bin/env python3
"""Developer tool: BASELINE.json configs[4] at FULL size on ONE GPU, through the product entries only —
500 synthetic 640 x 480 PNG files -> `ViTExtractor("dinov2_vitb14", 2048 keypoints, 256-D).extract` (files -> SQLite) ->
`match_exhaustive` (database in, database out: 124 750 pairs at 2048 x 256, two-view verification included).
Prints the wall time of each leg and the database row counts; nothing is compared with an oracle here (the chain's parity is
tests/test_configs_gpu.py at 24 images) — this run shows the full-size configuration fits and what it costs on one MI355X.

Images: 25 smooth random textures, each seen through 20 windows shifted by whole patches, so that images of one texture really
match (with unrelated noise images every pair would be empty and the verification leg trivial).
usage: python tools/run_c5_full.py [n_images=500]
"""
import os, sys, tempfile, time, sqlite3
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vit_colmap_amd.features.vit_extractor import ViTExtractor
from vit_colmap_amd.matching import match_exhaustive
from vit_colmap_amd.utils import image_io
from vit_colmap_amd.utils.config import MatchingConfig

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 500
views = 20
rng = np.random.default_rng(0)
tmp = tempfile.mkdtemp(prefix="c5_full_")
img_dir = os.path.join(tmp, "images")
os.makedirs(img_dir)
t0 = time.perf_counter()
k = 0
while k < n_images:
    # a texture larger than the frame: low-pass noise on three scales, so that patches are distinguishable
    big = np.zeros((480 + 14 * 8, 640 + 14 * 8, 3), np.float32)
    for cell in (7, 28, 112):
        g = rng.random((big.shape[0] // cell + 2, big.shape[1] // cell + 2, 3), dtype=np.float32)
        big += np.kron(g, np.ones((cell, cell, 1), np.float32))[: big.shape[0], : big.shape[1]]
    big = (255 * (big - big.min()) / (big.max() - big.min())).astype(np.uint8)
    for v in range(views):
        if k >= n_images:
            break
        dy, dx = 14 * (v % 5), 14 * (v // 5)
        image_io.imwrite(os.path.join(img_dir, f"img_{k:04d}.png"), big[dy:dy + 480, dx:dx + 640])
        k += 1
print(f"{n_images} PNG files written in {time.perf_counter() - t0:.1f} s", flush=True)

db_path = os.path.join(tmp, "c5.db")
ex = ViTExtractor(model_name="dinov2_vitb14", num_keypoints=2048, descriptor_dim=256, device="cuda:0", seed=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
ex.extract(img_dir, db_path, "SIMPLE_PINHOLE")
torch.cuda.synchronize()
t_ext = time.perf_counter() - t0
print(f"extract (files -> database): {t_ext:.2f} s = {n_images / t_ext:.0f} images/s", flush=True)

opts = MatchingConfig(max_ratio=1.0, max_distance=1.5).to_matching_options()   # random weights: see tests/test_configs_gpu.py
t0 = time.perf_counter()
stats = match_exhaustive(db_path, matching_options=opts)
t_match = time.perf_counter() - t0
print(f"match_exhaustive: {t_match:.2f} s for {stats['pairs']} pairs = {stats['pairs'] / t_match:.0f} pairs/s "
      f"(matching + verification {stats['gpu_s']:.2f} s, database {stats['db_s']:.2f} s); "
      f"{stats['matches']} matches, {stats['verified_pairs']} verified pairs", flush=True)

con = sqlite3.connect(db_path)
for table in ("images", "keypoints", "descriptors", "matches", "two_view_geometries"):
    print(f"  {table}: {con.execute(f'SELECT COUNT(*) FROM {table}').fetchone()[0]} rows")
r, c = con.execute("SELECT rows, cols FROM descriptors LIMIT 1").fetchone()
print(f"  descriptors per image: {r} x {c}; database file {os.path.getsize(db_path) / 2**20:.0f} MiB")
con.close()
