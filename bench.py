#!/usr/bin/env python3
"""Benchmark of the vit-colmap hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under a launcher (torch.distributed.run sets WORLD_SIZE) this
process is one rank and WORLD_SIZE must equal --gpus; without one, the process starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` itself, before anything has touched
the GPU, and exits with that job's code.  A run that cannot have N ranks fails loudly (exit 2).

One step = one pass of the hot path over one batch of synthetic input that is already resident
in HBM:
    per rank its block of the synthetic 640x480 BGR frames -> HIP preprocess -> DINOv2 ViT-S/14 (bf16, random
    weights: no checkpoint offline) -> HIP keypoint selection (512 targets) + 384-D uint8
    descriptors -> [N > 1: all-gather of the descriptor blocks] -> HIP exhaustive matcher over this
    rank's share (pair p -> rank p % N) of all pairs among the images.
Which images:
    --gpus 1 (default)      50 images: BASELINE.json configs[1] + configs[2], the configuration the metric is quoted on;
                            the line also carries `strong_scaling_200`, the same GPU on the 200-image set below, so that
                            a scaling run has its N = 1 anchor
    --gpus N > 1 (default)  STRONG scaling on the fixed 200-image set of north_star / configs[3] (--images-total 200):
                            images in contiguous blocks of ceil(200 / N), all 19 900 pairs dealt p % N
    --weak                  50 images per GPU (50 N images, all pairs among them): the round-1/2 experiment
    --config c5             configs[4]'s single-GPU share: ViT-B/14, 2048 keypoints, PCA -> 256-D, 63 images
                            (ceil(500 / 8)) and 2048 x 256 matcher blocks; the roofline is stated against int8 MFMA
`value` = images/s of that whole step (all ranks).  Nothing is copied to the host inside the timed
region; SQLite writes are host work outside the accelerated path and are not timed here.

The matcher leg of BASELINE's metric is defined on fixed-size blocks (N = 512 keypoints, D = 384, all pairs
among 50 blocks per GPU: configs[2]) because the reference's NMS keeps a data-dependent ~115
keypoints per image.  It is measured in the same run by a second timed loop of launches with HIP events on the
launch stream (every rank matches its share of the pairs among the 50*N blocks; the slowest rank's time counts);
`pair_matches_per_s` and the `roofline` object come from that loop.  `pair_matches_per_s_dense` is the same loop
on "scene" descriptors (SIFT-like, overlapping views: every similarity tile is relevant), the regime real
descriptors of overlapping images are in.

`cpu_baseline` (rank 0, N = 1 only): the CPU oracle (a port of the reference's algorithm, see
oracle/) timed on the box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

IMAGES_PER_RANK = 50
W, H = 640, 480
NUM_KEYPOINTS, DESC_DIM = 512, 384
VIT_FLOP_PER_IMAGE = 1.09e11          # ViT-S/14 at 1531 tokens
MATCH_BYTES_PER_PAIR = 2 * NUM_KEYPOINTS * DESC_DIM + 2 * NUM_KEYPOINTS * 12   # 405 504 B
MATCH_OPS_PER_PAIR = 2.0 * NUM_KEYPOINTS * NUM_KEYPOINTS * DESC_DIM
# configs[4] (C5): ViT-B/14, 2048 keypoints, PCA -> 256-D, 500 images on 8 GPUs -> 63 per GPU
C5 = dict(model="dinov2_vitb14", num_keypoints=2048, desc_dim=256, images=63, flop_per_image=3.48e11,
          bytes_per_pair=2 * 2048 * 256 + 2 * 2048 * 12, ops_per_pair=2.0 * 2048 * 2048 * 256)   # 1 097 728 B
C2 = dict(model="dinov2_vits14", num_keypoints=NUM_KEYPOINTS, desc_dim=DESC_DIM, images=IMAGES_PER_RANK,
          flop_per_image=VIT_FLOP_PER_IMAGE, bytes_per_pair=MATCH_BYTES_PER_PAIR, ops_per_pair=MATCH_OPS_PER_PAIR)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0        # dense bf16
MFMA_INT8_PEAK_TOPS = 5000.0          # dense int8 = 2x bf16 per clock (MI355X_MICROARCH.md, Matrix cores)
TRAFFIC_PROFILE = "profiles/r03_matcher_traffic.json"   # rocprofv3 --pmc passes of this command (tools/prof_r03.sh)
STRONG_IMAGES = 200                   # north_star / configs[3]: the fixed set of the scaling experiment
# Developer rehearsal of the N > 1 code path on a box with ONE GPU (VITCOLMAP_BENCH_REHEARSE=1): every rank uses cuda:0 and the
# collectives go through gloo on host copies.  The line it prints says "rehearsal": true and is not a measurement.
REHEARSE = os.environ.get("VITCOLMAP_BENCH_REHEARSE") == "1"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150, help="timed steps (default: > 1 s of timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--match-launches", type=int, default=0,
                    help="launches of the matcher micro-loops (default: enough for ~0.5 s each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--images-total", type=int, default=0,
                    help="images of the whole job (strong scaling); default 50 at --gpus 1, 200 at --gpus N > 1")
    ap.add_argument("--weak", action="store_true", help="50 images per GPU instead of a fixed set")
    ap.add_argument("--config", choices=["c2", "c5"], default="c2",
                    help="c2: configs[1]+[2] (ViT-S, 512 x 384); c5: configs[4]'s one-GPU share (ViT-B, 2048 x 256)")
    ap.add_argument("--no-strong-anchor", action="store_true", help="skip the 200-image leg of the default N = 1 line")
    ap.add_argument("--no-pipelining", action="store_true",
                    help="start every step's extraction behind the previous step's matching (no overlap between consecutive steps)")
    return ap.parse_args()


def start_ranks(args):
    """No launcher: become one.  Runs before this process has made any GPU call (torch is imported, the GPU is
    not initialised: device_count() does not do that on this image), so no process that has touched the GPU is
    ever replaced or re-executed; the ranks are fresh children."""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and not REHEARSE:
        print(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); refusing to report fewer ranks "
              "than asked for", file=sys.stderr)
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def synthetic_frames(rank, n):
    """640x480 BGR checkerboards (tile 40, reference tests/test_smoke_e2e.py:10-17) shifted per image
    plus seeded uniform noise (seed 1000 + global index) — SURVEY.md §8d."""
    import numpy as np

    base = np.zeros((H, W, 3), np.uint8)
    for y in range(0, H, 40):
        for x in range(0, W, 40):
            if ((x // 40) + (y // 40)) % 2 == 0:
                base[y:y + 40, x:x + 40] = 255
    out = np.empty((n, H, W, 3), np.uint8)
    for k in range(n):
        g = rank * n + k
        rs = np.random.RandomState(1000 + g)
        img = np.roll(base, (7 * g % W, 5 * g % H), (1, 0)).astype(np.int16)
        img += rs.randint(-40, 41, img.shape).astype(np.int16)
        out[k] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def c3_descriptor_blocks(n_images, n_rows=NUM_KEYPOINTS, d=DESC_DIM):
    """Matcher micro-bench input of SURVEY.md §8d / BASELINE.md §3 (tests/util_data.py: numpy only)."""
    import numpy as np
    from util_data import synthetic_descriptors

    return np.stack([synthetic_descriptors(k, n_rows, d) for k in range(n_images)])


def host_cores():
    """CPUs this process may actually use (cgroup quota / affinity), not the machine's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001 - cgroup v1 or no cgroup: affinity is the answer
        pass
    return n


def cpu_baseline(frames):
    """Oracle timed on the host: ViT + selection on a few images, C matcher on a sample of pairs."""
    import numpy as np
    import torch
    from oracle import c_oracle, select_oracle, vit_oracle
    from oracle import matcher_oracle as mo
    from oracle import preprocess_oracle as po
    from vit_colmap_amd.vit import build_dinov2

    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.float() for k, v in build_dinov2("dinov2_vits14").init_random(0).state_dict().items()}
    n_img = 4
    t0 = time.perf_counter()
    for k in range(n_img):
        x, _ = po.preprocess(frames[k])
        with torch.no_grad():
            tok = vit_oracle.forward_patch_tokens(sd, torch.from_numpy(x)[None], 6)[0].numpy()
        fmap = np.ascontiguousarray(tok.T.reshape(DESC_DIM, H // 14, W // 14))
        select_oracle.dense_to_sparse(fmap, (W, H), (630, 476), NUM_KEYPOINTS, DESC_DIM, "harris")
    t_img = (time.perf_counter() - t0) / n_img
    desc = c3_descriptor_blocks(16)
    counts = np.full(16, NUM_KEYPOINTS, np.int32)
    pairs = mo.exhaustive_pairs(16)                     # 120 pairs of the same 512 x 384 blocks
    t0 = time.perf_counter()
    _, _, used = c_oracle.match_pairs(desc, counts, pairs, num_threads=cores)
    t_pair = (time.perf_counter() - t0) / len(pairs)
    n_pairs_step = IMAGES_PER_RANK * (IMAGES_PER_RANK - 1) // 2
    step_s = IMAGES_PER_RANK * t_img + n_pairs_step * t_pair
    return {
        "value": IMAGES_PER_RANK / step_s, "unit": "images/s", "cores": cores, "kind": "port",
        "sample": f"{n_img} images through the float32 ViT-S + selection oracle ({t_img*1e3:.0f} ms/image, torch "
                  f"{cores} threads) and {len(pairs)} pairs of 512x384 uint8 blocks through the C matcher oracle "
                  f"({t_pair*1e3:.2f} ms/pair, OpenMP {used} threads), scaled to 50 images + 1225 pairs",
        "extract_images_per_s": 1.0 / t_img, "pair_matches_per_s": 1.0 / t_pair,
    }


def plan_job(config: str, world: int, images_total: int = 0, weak: bool = False):
    """-> (images of the whole job, "strong" | "weak").  Default: configs[1] (50 images) on one GPU, the fixed 200-image
    set of north_star / configs[3] on N > 1 GPUs (strong scaling); --weak: the per-GPU count times N; c5: 63 per GPU."""
    per_gpu = (C5 if config == "c5" else C2)["images"]
    if weak and images_total:
        raise ValueError("--weak and --images-total exclude each other")
    if weak:
        n_total, scaling = per_gpu * world, "weak"
    elif images_total:
        n_total, scaling = images_total, "strong"
    elif world > 1:
        n_total, scaling = (STRONG_IMAGES, "strong") if config == "c2" else (per_gpu * world, "weak")
    else:
        n_total, scaling = per_gpu, "strong"            # one GPU: the set is what it is
    if n_total < 2 * world:
        raise ValueError(f"{n_total} images on {world} rank(s): fewer than two images per rank")
    return n_total, scaling


def main():
    args = parse_args()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            start_ranks(args)                     # does not return
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:                    # a line with n_gpus != --gpus would be read as an N-GPU result
            if rank == 0:
                print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
            sys.exit(2)
    cfg = C5 if args.config == "c5" else C2
    try:
        n_total, scaling = plan_job(args.config, world, args.images_total, args.weak)
    except ValueError as e:
        print(f"bench.py: {e}", file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if REHEARSE:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()             # what was actually initialised
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from util_data import image_set
    from vit_colmap_amd import dist as vd
    from vit_colmap_amd.features.vit_extractor import ViTExtractor
    from vit_colmap_amd.matching import match_pairs, prepare_descriptors

    K, D = cfg["num_keypoints"], cfg["desc_dim"]
    quiet = open(os.devnull, "w")
    so, sys.stdout = sys.stdout, quiet                       # the extractor prints like the reference does
    ex = ViTExtractor(model_name=cfg["model"], num_keypoints=K, descriptor_dim=D, device=str(dev), precision="bf16", seed=0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if REHEARSE else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run_job(n_images, steps, warmup):
        """The hot path over a fixed set of n_images synthetic frames sharded over the ranks (contiguous blocks of
        ceil(n / world), the last ranks padded with empty images; pair p of the exhaustive list -> rank p % world).
        -> (elapsed seconds of `steps` steps, max over ranks; per-leg ms of rank 0; last result; pairs of the job)."""
        per = (n_images + world - 1) // world
        lo, hi = vd.shard_range(n_images, rank, world)
        frames_np = synthetic_frames(0, n_images)[lo:hi] if hi > lo else np.zeros((0, H, W, 3), np.uint8)
        frames = torch.from_numpy(frames_np).to(dev)         # resident in HBM before the timed region
        frames_ready = torch.cuda.Event()                    # (the frames never change: consecutive steps may pipeline)
        frames_ready.record()
        n_slots = per * world                                # image slots after padding (slots >= n_images are empty)
        my_pairs = torch.from_numpy(vd.pairs_for_rank(n_images, rank, world)).to(dev)
        out_m = torch.empty((my_pairs.shape[0], K, 2), dtype=torch.int32, device=dev)
        out_c = torch.empty((my_pairs.shape[0],), dtype=torch.int32, device=dev)
        pad_d = torch.zeros((per - (hi - lo), K, D), dtype=torch.uint8, device=dev)
        pad_c = torch.zeros((per - (hi - lo),), dtype=torch.int32, device=dev)
        events = []

        def step(timed):
            # No host synchronisation inside a step: the steps are enqueued back to back (the GPU never waits for Python
            # to launch the next step's first kernels) and the timed region is closed by barrier() below.
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
            if timed:
                ev[0].record()
            res = ex.extract_device(frames, input_ready=None if args.no_pipelining else frames_ready)   # preprocess + ViT + selection + descriptors
            if timed:
                ev[1].record()
            d_loc, c_loc = res["desc_u8"], res["count"]
            if pad_c.numel():
                d_loc, c_loc = torch.cat([d_loc, pad_d]), torch.cat([c_loc, pad_c])
            if REHEARSE and world > 1:
                desc, counts = (t.to(dev) for t in vd.all_gather_descriptors(d_loc.cpu(), c_loc.cpu()))
            else:
                desc, counts = vd.all_gather_descriptors(d_loc, c_loc)
            if timed:
                ev[2].record()
            prepared = prepare_descriptors(desc, counts)
            match_pairs(prepared, counts, n_slots, K, D, my_pairs, out_matches=out_m, out_counts=out_c)
            if timed:
                ev[3].record()
                events.append(ev)
            return res

        res = None
        for _ in range(warmup):
            res = step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step(True)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        legs = {"extract": 0.0, "gather": 0.0, "match": 0.0}
        for ev in events:
            legs["extract"] += ev[0].elapsed_time(ev[1])
            legs["gather"] += ev[1].elapsed_time(ev[2])
            legs["match"] += ev[2].elapsed_time(ev[3])
        return elapsed, legs, res, frames_np, hi - lo, n_images * (n_images - 1) // 2

    elapsed, leg_ms, res, frames_np, n_local, n_pairs_global = run_job(n_total, args.steps, args.warmup)
    sys.stdout = so

    # ---- matcher leg at the fixed block shape (configs[2]: 512 x 384; c5: 2048 x 256): all pairs among n_total blocks --
    def matcher_loop(blocks_np, n_blocks, rows, dim):
        blocks = torch.from_numpy(blocks_np).to(dev)
        counts = torch.full((n_blocks,), rows, dtype=torch.int32, device=dev)
        pairs = torch.from_numpy(vd.pairs_for_rank(n_blocks, rank, world)).to(dev)
        P = pairs.shape[0]
        om = torch.empty((P, rows, 2), dtype=torch.int32, device=dev)
        oc = torch.empty((P,), dtype=torch.int32, device=dev)
        prepared = prepare_descriptors(blocks, counts)
        for _ in range(3):
            match_pairs(prepared, counts, n_blocks, rows, dim, pairs, out_matches=om, out_counts=oc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                                          # HIP events on the stream the kernel is launched on
        match_pairs(prepared, counts, n_blocks, rows, dim, pairs, out_matches=om, out_counts=oc)
        e1.record()
        torch.cuda.synchronize()
        n = args.match_launches or max(20, min(20000, int(500.0 / max(e0.elapsed_time(e1), 1e-3))))   # ~0.5 s
        barrier()
        e0.record()
        for _ in range(n):
            match_pairs(prepared, counts, n_blocks, rows, dim, pairs, out_matches=om, out_counts=oc)
        e1.record()
        torch.cuda.synchronize()
        launch_ms = e0.elapsed_time(e1) / n                  # this rank's kernel: P pairs per launch
        slowest_ms = max_over_ranks(launch_ms)
        assert int((oc < 0).sum().item()) == 0, "matcher self-check failed (VC_COUNT_SELFCHECK_FAILED)"
        return P, n, launch_ms, slowest_ms, int(oc.sum().item())

    n_blocks = n_total
    P, n_launch, launch_ms, slowest_ms, _ = matcher_loop(c3_descriptor_blocks(n_blocks, K, D), n_blocks, K, D)
    pair_rate = n_pairs_global / slowest_ms * 1e3            # whole job: all ranks' pairs / slowest rank's launch
    dense_np, _ = image_set(1, n_blocks, K, D, kind="scene")
    Pd, n_launch_d, launch_ms_d, slowest_ms_d, dense_matches = matcher_loop(dense_np, n_blocks, K, D)
    pair_rate_dense = n_pairs_global / slowest_ms_d * 1e3
    c5_shape = None
    if args.config == "c2" and world == 1:
        # the configs[4] block shape (2048 x 256) on this GPU: 48 blocks, all 1128 pairs, sparse and dense input
        r5, d5, nb5 = C5["num_keypoints"], C5["desc_dim"], 48
        P5, _, ms5, _, _ = matcher_loop(c3_descriptor_blocks(nb5, r5, d5), nb5, r5, d5)
        d5_np, _ = image_set(5, nb5, r5, d5, kind="scene")
        _, _, ms5d, _, _ = matcher_loop(d5_np, nb5, r5, d5)
        c5_shape = {
            "blocks": [nb5, r5, d5], "pairs_per_launch": P5, "bytes_per_pair": C5["bytes_per_pair"],
            "pair_matches_per_s": round(P5 / ms5 * 1e3, 1), "pair_matches_per_s_dense": round(P5 / ms5d * 1e3, 1),
            "hbm_contract_frac": round(P5 * C5["bytes_per_pair"] / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "int8_tops_nominal": round(P5 * C5["ops_per_pair"] / (ms5 * 1e-3) / 1e12, 1),
            "int8_tops_nominal_dense": round(P5 * C5["ops_per_pair"] / (ms5d * 1e-3) / 1e12, 1),
            "int8_mfma_frac_dense": round(P5 * C5["ops_per_pair"] / (ms5d * 1e-3) / 1e12 / MFMA_INT8_PEAK_TOPS, 4),
            "note": "at 2048 x 256 the HBM contract (1 097 728 B per pair) would need 10.9 Pop/s of int8 at 70 %: the honest "
                    "bound is the int8 matrix pipe (dense peak 5 Pop/s); on dense data every MAC is executed, so "
                    "int8_mfma_frac_dense is a real pipe fraction; on the sparse input two thirds of the MFMAs are skipped",
        }

    # ---- the N = 1 anchor of the strong-scaling experiment: the 200-image set on this one GPU ---------------------------
    strong_anchor = None
    if world == 1 and args.config == "c2" and n_total != STRONG_IMAGES and not args.no_strong_anchor:
        sys.stdout = quiet
        st = max(4, args.steps // 10)
        el2, legs2, _, _, _, np2 = run_job(STRONG_IMAGES, st, 2)
        sys.stdout = so
        strong_anchor = {
            "images_total": STRONG_IMAGES, "pairs_per_step": np2, "steps": st, "value": round(STRONG_IMAGES * st / el2, 2),
            "unit": "images/s", "ms_per_step": round(el2 / st * 1e3, 3),
            "legs_ms_per_step_rank0": {k: round(v / st, 3) for k, v in legs2.items()},
            "note": "what `bench.py --gpus 1 --images-total 200` reports as its value: the N = 1 point of the default "
                    "--gpus N > 1 runs (strong scaling on the fixed 200-image set)",
        }

    # ---- files -> database: the plugin entry `extract(image_dir, db_path, ...)` end to end (decode, upload, GPU, SQLite) ----
    e2e = None
    if world == 1 and args.config == "c2" and rank == 0:
        import shutil
        import tempfile

        from vit_colmap_amd.utils import image_io

        tmp = tempfile.mkdtemp(prefix="vc_bench_")
        try:
            n_files = 100
            fr = synthetic_frames(0, n_files)
            os.makedirs(os.path.join(tmp, "images"))
            for k in range(n_files):
                image_io.imwrite(os.path.join(tmp, "images", f"img_{k:03d}.png"), fr[k])
            sys.stdout = quiet
            best = None
            for rep in range(3):                                 # first pass warms the pinned buffers and the page cache
                dbp = os.path.join(tmp, f"e2e_{rep}.db")
                ex.timings = {k: 0 if k == "images" else 0.0 for k in ex.timings}
                t0 = time.perf_counter()
                ex.extract(os.path.join(tmp, "images"), dbp, "SIMPLE_PINHOLE")
                dt = time.perf_counter() - t0
                if rep and (best is None or dt < best[0]):
                    best = (dt, dict(ex.timings))
            sys.stdout = so
            e2e = {
                "value": round(n_files / best[0], 1), "unit": "images/s", "images": n_files,
                "what": "ViTExtractor.extract(directory of 640x480 PNG files -> COLMAP SQLite): threaded decode, pinned upload, "
                        "GPU batches of 50 on a side stream, rows of batch k-1 written while batch k runs; host cores "
                        f"{host_cores()}; NOT part of `value` (that one starts from frames resident in HBM)",
                "seconds": round(best[0], 4),
                "host_breakdown_s": {k: round(v, 4) for k, v in best[1].items() if k != "images"},
            }
        finally:
            sys.stdout = so
            shutil.rmtree(tmp, ignore_errors=True)

    # ---- the other extractor of the plugin API, device-resident like `value`: informational ----------
    trainable = None
    if world == 1 and args.config == "c2" and rank == 0 and not args.no_strong_anchor:
        from vit_colmap_amd.features.trainable_vit_extractor import TrainableViTExtractor

        sys.stdout = quiet
        try:
            tex = TrainableViTExtractor(model_name="dinov2_vits14", num_keypoints=2048, device=f"cuda:{local_rank}")
            tb = 32
            tframes = torch.from_numpy(synthetic_frames(0, tb)).to(dev)
            for _ in range(2):
                tex.extract_device(tframes)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                tex.extract_device(tframes)
            torch.cuda.synchronize()
            tdt = (time.perf_counter() - t0) / 5
            trainable = {
                "value": round(tb / tdt, 1), "unit": "images/s", "batch": tb, "ms_per_batch": round(tdt * 1e3, 3),
                "what": "TrainableViTExtractor(dinov2_vits14, 2048 keypoints).extract_device on 640x480 frames resident in HBM: "
                        "ViT-S backbone + convolutional heads (vc_conv_taps_bf16: implicit GEMM on the 256x256 tile, 265 GFLOP per "
                        "image) + heat-map selection; random weights; NOT part of `value`",
            }
            del tex, tframes
        except Exception as e:  # noqa: BLE001 - an informational leg must not take the contract line down with it
            trainable = {"error": f"{type(e).__name__}: {e}"}
        finally:
            sys.stdout = so

    if rank == 0:
        bpp, opp = cfg["bytes_per_pair"], cfg["ops_per_pair"]
        achieved = P * bpp / (launch_ms * 1e-3) / 1e9          # rank 0's launch, rank 0's GPU
        achieved_d = Pd * bpp / (launch_ms_d * 1e-3) / 1e9
        traffic = traffic_src = executed = None
        tpath = os.path.join(ROOT, TRAFFIC_PROFILE)
        if world == 1 and args.config == "c2" and n_total == IMAGES_PER_RANK and os.path.exists(tpath):
            prof = json.load(open(tpath))
            traffic = prof.get("hbm_bytes_per_launch")
            executed = prof.get("mfma", {}).get("executed_fraction_of_nominal")
            traffic_src = f"{TRAFFIC_PROFILE}: PMC passes of this command, not a measurement of this run"
        nominal_tops = P * opp / (launch_ms * 1e-3) / 1e12
        roof = {
            "kernel": f"pair2_kernel<{(D + 31) // 32}> (persistent; fused int8-MFMA similarity + row/column top-2 + ratio/cross-check)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "launch_ms": round(launch_ms, 4), "launches_timed": n_launch, "pairs_per_launch": P,
            "bytes_per_pair": bpp,
            "regime": "CONTRACT rate on NON-MATCHING data (the SURVEY §8d input: no tile holds a relevant similarity, the exact "
                      "early-out skips two thirds of every tile's MFMAs and every epilogue): algorithmic bytes per pair x pairs / "
                      "time, not an HBM or MFMA efficiency — see roofline_dense for data where every tile matters",
            "int8_tops_nominal": round(nominal_tops, 1),
            "mfma_executed_fraction": executed,
            "int8_tops_executed": None if executed is None else round(nominal_tops * executed, 1),
            "mfma_executed_source": None if executed is None else traffic_src,
        }
        dense_tops = Pd * opp / (launch_ms_d * 1e-3) / 1e12
        roof_dense = {
            "kernel": roof["kernel"], "bound": "hbm", "achieved": round(achieved_d, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved_d / HBM_PEAK_GBS, 4), "traffic": None,
            "launch_ms": round(launch_ms_d, 4), "launches_timed": n_launch_d, "pairs_per_launch": Pd,
            "bytes_per_pair": bpp,
            "int8_tops": round(dense_tops, 1), "int8_mfma_frac": round(dense_tops / MFMA_INT8_PEAK_TOPS, 4),
            "matches_per_launch": dense_matches,
            "input": "tests/util_data.image_set(kind='scene'): every image a noisy subset of one descriptor pool, so "
                     "every 32x32 similarity tile holds relevant entries (the update path runs everywhere; every MAC executes)",
        }
        images = n_total * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        extract_ms = leg_ms["extract"] / args.steps
        # consecutive steps pipeline (the next step's ViT starts under this step's matching), so the per-leg events no longer
        # partition a step: the ViT's rate is stated against the WHOLE step time — a lower bound that needs no attribution
        vit_ms = ms_per_step if not args.no_pipelining else extract_ms
        vit_tflops = cfg["flop_per_image"] * n_local / (vit_ms * 1e-3) / 1e12
        if args.config == "c5":
            workload = (f"configs[4], one GPU's share: DINOv2 ViT-B/14 extract of {n_total} 640x480 images ({K} keypoint targets, "
                        f"768 -> {D}-D projected uint8 descriptors) then exhaustive matching of all pairs among them; "
                        f"matcher legs on {K} x {D} blocks")
        elif n_total == IMAGES_PER_RANK and world == 1:
            workload = ("configs[1]+configs[2]: DINOv2 ViT-S/14 extract of 50 640x480 images (512 keypoints, 384-D uint8 "
                        "descriptors) then exhaustive mutual-NN + ratio matching of all pairs among the extracted images")
        else:
            workload = (f"configs[3]: the fixed {n_total}-image 640x480 set, DINOv2 ViT-S/14 extract sharded in contiguous blocks "
                        f"over {world} GPU(s), one RCCL all-gather of the descriptor blocks, all {n_pairs_global} pairs dealt "
                        f"p % {world}") if scaling == "strong" else \
                       (f"weak scaling: {cfg['images']} images per GPU ({n_total} images), all {n_pairs_global} pairs dealt p % {world}")
        line = {
            "metric": "images/sec extracted + pair-matches/sec (N×D brute-force NN)",
            "value": round(images / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "bf16 (ViT) / u8+i32 (matcher)",
            "data": "synthetic",
            **({"rehearsal": True} if REHEARSE else {}),
            "config": {
                "workload": workload,
                "images_total": n_total, "images_rank0": n_local, "image_size": [W, H], "num_keypoints": K,
                "descriptor_dim": D, "pairs_per_step": n_pairs_global, "parallelism": f"images+pairs sharded x{world}",
                "keypoints_kept_per_image_mean": round(float(res["count"].float().mean().item()), 1),
                "vit_batch_shards": getattr(ex.model, "batch_shards", None) or int(os.environ.get("VITCOLMAP_VIT_SHARDS", "2")),
            },
            "timed_region_s": round(elapsed, 3),
            "extract_images_per_s": round(n_local * world / (extract_ms * 1e-3), 1),
            "steps_pipelined": not args.no_pipelining,
            "pair_matches_per_s": round(pair_rate, 1),
            "pair_matches_per_s_dense": round(pair_rate_dense, 1),
            "pair_matches_config": f"{n_blocks} blocks of {K}x{D} uint8, all {n_pairs_global} pairs dealt "
                                   f"round-robin to {world} rank(s), one launch per rank, slowest rank's time",
            "legs_ms_per_step_rank0": {k: round(v / args.steps, 3) for k, v in leg_ms.items()},
            "roofline": roof,
            "roofline_dense": roof_dense,
            "roofline_vit": {
                "bound": "mfma", "achieved": round(vit_tflops, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(vit_tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                "note": "the ViT's FLOPs of one step over the WHOLE step time (preprocess + ViT + selection + gather + matching; "
                        "consecutive steps overlap, so legs are not separable): a lower bound on the GEMM rate",
            },
            "matcher_c5_shape": c5_shape,
            "strong_scaling_200": strong_anchor,
            "extract_e2e_images_per_s": e2e,
            "trainable_extractor_images_per_s": trainable,
        }
        if world == 1 and not args.no_cpu_baseline and args.config == "c2":
            line["cpu_baseline"] = cpu_baseline(frames_np)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
