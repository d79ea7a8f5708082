#!/usr/bin/env python3
"""Benchmark of the vit-colmap hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under a launcher (torch.distributed.run sets WORLD_SIZE) this
process is one rank and WORLD_SIZE must equal --gpus; without one, the process starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` itself, before anything has touched
the GPU, and exits with that job's code.  A run that cannot have N ranks fails loudly (exit 2).

One step = one pass of the hot path over one batch of synthetic input that is already resident
in HBM (BASELINE.json configs[1] + configs[2] at N = 1):
    per rank 50 synthetic 640x480 BGR frames -> HIP preprocess -> DINOv2 ViT-S/14 (bf16, random
    weights: no checkpoint offline) -> HIP keypoint selection (512 targets) + 384-D uint8
    descriptors -> [N > 1: all-gather of the descriptor blocks] -> HIP exhaustive matcher over this
    rank's share of all pairs among the 50*N images.
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
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0        # dense bf16
TRAFFIC_PROFILE = "profiles/r02_matcher_traffic.json"   # rocprofv3 --pmc pass of this command (tools/prof_pmc.sh)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150, help="timed steps (default: > 1 s of timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--match-launches", type=int, default=0,
                    help="launches of the matcher micro-loops (default: enough for ~0.5 s each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def start_ranks(args):
    """No launcher: become one.  Runs before this process has made any GPU call (torch is imported, the GPU is
    not initialised: device_count() does not do that on this image), so no process that has touched the GPU is
    ever replaced or re-executed; the ranks are fresh children."""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus:
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


def c3_descriptor_blocks(n_images):
    """Matcher micro-bench input of SURVEY.md §8d / BASELINE.md §3 (tests/util_data.py: numpy only)."""
    import numpy as np
    from util_data import synthetic_descriptors

    return np.stack([synthetic_descriptors(k, NUM_KEYPOINTS, DESC_DIM) for k in range(n_images)])


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

    import numpy as np
    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()             # what was actually initialised
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from util_data import image_set
    from vit_colmap_amd import dist as vd
    from vit_colmap_amd.features.vit_extractor import ViTExtractor
    from vit_colmap_amd.matching import match_pairs, prepare_descriptors

    quiet = open(os.devnull, "w")
    so, sys.stdout = sys.stdout, quiet                       # the extractor prints like the reference does
    ex = ViTExtractor(model_name="dinov2_vits14", num_keypoints=NUM_KEYPOINTS, descriptor_dim=DESC_DIM,
                      device=str(dev), precision="bf16", seed=0)
    sys.stdout = so

    frames_np = synthetic_frames(rank, IMAGES_PER_RANK)
    frames = torch.from_numpy(frames_np).to(dev)             # resident in HBM before the timed region
    n_global = IMAGES_PER_RANK * world
    my_pairs = torch.from_numpy(vd.pairs_for_rank(n_global, rank, world)).to(dev)
    n_pairs_global = n_global * (n_global - 1) // 2
    out_m = torch.empty((my_pairs.shape[0], NUM_KEYPOINTS, 2), dtype=torch.int32, device=dev)
    out_c = torch.empty((my_pairs.shape[0],), dtype=torch.int32, device=dev)
    leg_ms = {"extract": 0.0, "gather": 0.0, "match": 0.0}
    step_events = []                                         # four events per timed step, read after the timed region

    def step(timed):
        # No host synchronisation inside a step: the K steps are enqueued back to back (the GPU never waits for Python to
        # launch the next step's first kernels) and the timed region is closed by barrier() below.
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if timed:
            ev[0].record()
        res = ex.extract_device(frames)                      # preprocess + ViT + selection + descriptors
        if timed:
            ev[1].record()
        desc, counts = vd.all_gather_descriptors(res["desc_u8"], res["count"])
        if timed:
            ev[2].record()
        prepared = prepare_descriptors(desc, counts)
        match_pairs(prepared, counts, n_global, NUM_KEYPOINTS, DESC_DIM, my_pairs, out_matches=out_m, out_counts=out_c)
        if timed:
            ev[3].record()
            step_events.append(ev)
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        res = step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(True)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    for ev in step_events:
        leg_ms["extract"] += ev[0].elapsed_time(ev[1])
        leg_ms["gather"] += ev[1].elapsed_time(ev[2])
        leg_ms["match"] += ev[2].elapsed_time(ev[3])

    # ---- matcher leg at the BASELINE shape (configs[2]): 50 blocks of 512 x 384 per GPU, all pairs, this rank's share ----
    def matcher_loop(blocks_np):
        blocks = torch.from_numpy(blocks_np).to(dev)
        counts = torch.full((n_global,), NUM_KEYPOINTS, dtype=torch.int32, device=dev)
        P = my_pairs.shape[0]
        prepared = prepare_descriptors(blocks, counts)
        for _ in range(3):
            match_pairs(prepared, counts, n_global, NUM_KEYPOINTS, DESC_DIM, my_pairs, out_matches=out_m, out_counts=out_c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                                          # HIP events on the stream the kernel is launched on
        match_pairs(prepared, counts, n_global, NUM_KEYPOINTS, DESC_DIM, my_pairs, out_matches=out_m, out_counts=out_c)
        e1.record()
        torch.cuda.synchronize()
        n = args.match_launches or max(20, min(20000, int(500.0 / max(e0.elapsed_time(e1), 1e-3))))   # ~0.5 s
        barrier()
        e0.record()
        for _ in range(n):
            match_pairs(prepared, counts, n_global, NUM_KEYPOINTS, DESC_DIM, my_pairs, out_matches=out_m, out_counts=out_c)
        e1.record()
        torch.cuda.synchronize()
        launch_ms = e0.elapsed_time(e1) / n                  # this rank's kernel: P pairs per launch
        slowest_ms = max_over_ranks(launch_ms)
        return P, n, launch_ms, slowest_ms, int(out_c.sum().item())

    P, n_launch, launch_ms, slowest_ms, _ = matcher_loop(c3_descriptor_blocks(n_global))
    pair_rate = n_pairs_global / slowest_ms * 1e3            # whole job: all ranks' pairs / slowest rank's launch
    dense_np, _ = image_set(1, n_global, NUM_KEYPOINTS, DESC_DIM, kind="scene")
    Pd, n_launch_d, launch_ms_d, slowest_ms_d, dense_matches = matcher_loop(dense_np)
    pair_rate_dense = n_pairs_global / slowest_ms_d * 1e3

    if rank == 0:
        achieved = P * MATCH_BYTES_PER_PAIR / (launch_ms * 1e-3) / 1e9          # rank 0's launch, rank 0's GPU
        achieved_d = Pd * MATCH_BYTES_PER_PAIR / (launch_ms_d * 1e-3) / 1e9
        traffic = traffic_src = None
        tpath = os.path.join(ROOT, TRAFFIC_PROFILE)
        if world == 1 and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            traffic_src = f"{TRAFFIC_PROFILE}: PMC passes of this command, not a measurement of this run"
        roof = {
            "kernel": "pair2_kernel<12> (persistent; fused int8-MFMA similarity + row/column top-2 + ratio/cross-check)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "launch_ms": round(launch_ms, 4), "launches_timed": n_launch, "pairs_per_launch": P,
            "bytes_per_pair": MATCH_BYTES_PER_PAIR,
            "int8_tops": round(P * MATCH_OPS_PER_PAIR / (launch_ms * 1e-3) / 1e12, 1),
        }
        roof_dense = {
            "kernel": roof["kernel"], "bound": "hbm", "achieved": round(achieved_d, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved_d / HBM_PEAK_GBS, 4), "traffic": None,
            "launch_ms": round(launch_ms_d, 4), "launches_timed": n_launch_d, "pairs_per_launch": Pd,
            "bytes_per_pair": MATCH_BYTES_PER_PAIR,
            "int8_tops": round(Pd * MATCH_OPS_PER_PAIR / (launch_ms_d * 1e-3) / 1e12, 1),
            "matches_per_launch": dense_matches,
            "input": "tests/util_data.image_set(kind='scene'): every image a noisy subset of one descriptor pool, so "
                     "every 32x32 similarity tile holds relevant entries (the update path runs everywhere)",
        }
        images = n_global * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        extract_ms = leg_ms["extract"] / args.steps
        vit_tflops = VIT_FLOP_PER_IMAGE * IMAGES_PER_RANK / (extract_ms * 1e-3) / 1e12
        line = {
            "metric": "images/sec extracted + pair-matches/sec (N×D brute-force NN)",
            "value": round(images / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (ViT) / u8+i32 (matcher)",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]+configs[2]: DINOv2 ViT-S/14 extract of 50 640x480 images per GPU "
                            "(512 keypoints, 384-D uint8 descriptors) then exhaustive mutual-NN + ratio matching of "
                            "all pairs among the extracted images",
                "images_per_gpu": IMAGES_PER_RANK, "image_size": [W, H], "num_keypoints": NUM_KEYPOINTS,
                "descriptor_dim": DESC_DIM, "pairs_per_step": n_pairs_global, "parallelism": f"images+pairs sharded x{world}",
                "keypoints_kept_per_image_mean": round(float(res["count"].float().mean().item()), 1),
            },
            "timed_region_s": round(elapsed, 3),
            "extract_images_per_s": round(IMAGES_PER_RANK * world / (extract_ms * 1e-3), 1),
            "pair_matches_per_s": round(pair_rate, 1),
            "pair_matches_per_s_dense": round(pair_rate_dense, 1),
            "pair_matches_config": f"configs[2] per GPU: {n_global} blocks of 512x384 uint8, all {n_pairs_global} pairs dealt "
                                   f"round-robin to {world} rank(s), one launch per rank, slowest rank's time",
            "legs_ms_per_step_rank0": {k: round(v / args.steps, 3) for k, v in leg_ms.items()},
            "roofline": roof,
            "roofline_dense": roof_dense,
            "roofline_vit": {
                "bound": "mfma", "achieved": round(vit_tflops, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(vit_tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                "note": "whole extract leg (preprocess + ViT + selection) against the ViT's FLOPs: a lower bound on the GEMM rate",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(frames_np)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
