#!/usr/bin/env python3
"""Benchmark of the vit-colmap hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

One step = one pass of the hot path over one batch of synthetic input that is already resident
in HBM (BASELINE.json configs[1] + configs[2] at N = 1):
    per rank 50 synthetic 640x480 BGR frames -> HIP preprocess -> DINOv2 ViT-S/14 (bf16, random
    weights: no checkpoint offline) -> HIP keypoint selection (512 targets) + 384-D uint8
    descriptors -> [N > 1: all-gather of the descriptor blocks] -> HIP exhaustive matcher over this
    rank's share of all pairs among the 50*N images.
`value` = images/s of that whole step (all ranks).  Nothing is copied to the host inside the timed
region; SQLite writes are host work outside the accelerated path and are not timed here.

The matcher leg of BASELINE's metric is defined on fixed-size blocks (N = 512 keypoints, D = 384,
all 1225 pairs: configs[2]) because the reference's NMS keeps a data-dependent
~115 keypoints per image.  It is measured in the same run by a second timed loop of the same
number of launches with HIP events on the launch stream; `pair_matches_per_s` and the `roofline`
object come from that loop (rank 0).

`cpu_baseline` (rank 0, N = 1 only): the CPU oracle (a port of the reference's algorithm, see
oracle/) timed on the box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

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


def synthetic_frames(rank, n):
    """640x480 BGR checkerboards (tile 40, reference tests/test_smoke_e2e.py:10-17) shifted per image
    plus seeded uniform noise (seed 1000 + global index) — SURVEY.md §8d."""
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
    """Matcher micro-bench input of SURVEY.md §8d / BASELINE.md §3."""
    from oracle.matcher_oracle import synthetic_descriptors   # data generator only (numpy)

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


def cpu_baseline(frames, steps_hint):
    """Oracle timed on the host: ViT + selection on a few images, C matcher on a sample of pairs."""
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

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
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    leg_ms = {"extract": 0.0, "gather": 0.0, "match": 0.0}

    def step(timed):
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
            torch.cuda.synchronize()
            leg_ms["extract"] += ev[0].elapsed_time(ev[1])
            leg_ms["gather"] += ev[1].elapsed_time(ev[2])
            leg_ms["match"] += ev[2].elapsed_time(ev[3])
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- matcher leg at the BASELINE shape (configs[2]): 50 x 512 x 384, 1225 pairs, rank 0 ----------
    roof = pair_rate = None
    if rank == 0:
        c3 = torch.from_numpy(c3_descriptor_blocks(IMAGES_PER_RANK)).to(dev)
        c3_counts = torch.full((IMAGES_PER_RANK,), NUM_KEYPOINTS, dtype=torch.int32, device=dev)
        c3_pairs = torch.from_numpy(vd.pairs_for_rank(IMAGES_PER_RANK, 0, 1)).to(dev)
        P = c3_pairs.shape[0]
        cm = torch.empty((P, NUM_KEYPOINTS, 2), dtype=torch.int32, device=dev)
        cc = torch.empty((P,), dtype=torch.int32, device=dev)
        prepared = prepare_descriptors(c3, c3_counts)
        for _ in range(max(args.warmup, 2)):
            match_pairs(prepared, c3_counts, IMAGES_PER_RANK, NUM_KEYPOINTS, DESC_DIM, c3_pairs, out_matches=cm, out_counts=cc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                                          # HIP events on the stream the kernel is launched on
        for _ in range(args.steps):
            match_pairs(prepared, c3_counts, IMAGES_PER_RANK, NUM_KEYPOINTS, DESC_DIM, c3_pairs, out_matches=cm, out_counts=cc)
        e1.record()
        torch.cuda.synchronize()
        launch_ms = e0.elapsed_time(e1) / args.steps
        pair_rate = P / launch_ms * 1e3
        achieved = P * MATCH_BYTES_PER_PAIR / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_matcher_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        roof = {
            "kernel": "pair_kernel<12,2,true> (fused int8-MFMA similarity + top-2 + ratio/cross-check)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "launch_ms": round(launch_ms, 4), "pairs_per_launch": P, "bytes_per_pair": MATCH_BYTES_PER_PAIR,
            "int8_tops": round(P * MATCH_OPS_PER_PAIR / (launch_ms * 1e-3) / 1e12, 1),
        }

    if rank == 0:
        images = n_global * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        extract_ms = leg_ms["extract"] / args.steps
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
            "extract_images_per_s": round(IMAGES_PER_RANK * world / (extract_ms * 1e-3), 1),
            "pair_matches_per_s": round(pair_rate, 1),
            "pair_matches_config": "configs[2]: 50 blocks of 512x384 uint8, all 1225 pairs, one launch",
            "legs_ms_per_step_rank0": {k: round(v / args.steps, 3) for k, v in leg_ms.items()},
            "roofline": roof,
            "roofline_vit": {
                "bound": "mfma", "achieved": round(VIT_FLOP_PER_IMAGE * IMAGES_PER_RANK / (extract_ms * 1e-3) / 1e12, 1),
                "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(VIT_FLOP_PER_IMAGE * IMAGES_PER_RANK / (extract_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                "note": "whole extract leg (preprocess + ViT + selection) against the ViT's FLOPs: a lower bound on the GEMM rate",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(frames_np, args.steps)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
