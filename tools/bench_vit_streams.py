#!/usr/bin/env python3
"""Developer tool: does running the ViT-S block loop as K independent batch shards on K HIP streams fill the
tails of the persistent kernels (MLP: 2.34 rounds of row tiles run as 3; attention 3.5 as 4)?
Times `_blocks_hip` on (50, 1531, 384) whole against 2 / 3 shards, outputs compared bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_colmap_amd.vit import hip_ops as ops  # noqa: E402
from vit_colmap_amd.vit.dinov2 import build_dinov2  # noqa: E402

B, N, C = int(os.environ.get("B", 50)), 1531, 384
dev = "cuda"
m = build_dinov2("dinov2_vits14").init_random(0).fold_layerscale().to(dev).eval()
m.prepare_hip()
x0 = torch.randn(B, N, C, device=dev, dtype=torch.bfloat16)


def run_whole(x):
    return m._blocks_hip(x.clone())


def run_sharded(x, streams, bounds):
    x = x.clone()
    cur = torch.cuda.current_stream()
    outs = [None] * len(streams)
    ev0 = torch.cuda.Event()
    ev0.record(cur)
    xs = [x[bounds[i]:bounds[i + 1]] for i in range(len(streams))]
    for s in streams:
        s.wait_event(ev0)
    for li, (blk, hw) in enumerate(zip(m.blocks, m._hip)):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                xi = xs[i]
                a = ops.attention(hw["qkv"](xi), blk.attn.num_heads, q_prescaled=True)
                hw["proj"](a, ops.EPI_RESIDUAL, residual=xi, out=xi)
                hw["mlp"](xi)
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            outs[i] = ops.layernorm_drop_first(xs[i], m.norm.weight, m.norm.bias, m.norm.eps)
        ev = torch.cuda.Event()
        ev.record(s)
        cur.wait_event(ev)
    return torch.cat(outs, 0)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ref = run_whole(x0)
t = timeit(lambda: run_whole(x0))
print(f"whole batch, one stream: {t:.3f} ms")
def even(k):
    return [B * i // k for i in range(k + 1)]


def run_cfg(n_shards, n_streams):
    """n_shards contiguous shards dealt round-robin to n_streams streams (shards of one stream run in order)."""
    bounds = even(n_shards)
    streams = [torch.cuda.Stream() for _ in range(n_streams)]

    def go(x):
        x = x.clone()
        cur = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(cur)
        for s in streams:
            s.wait_event(ev0)
        out = torch.empty((B, N - 1, C), dtype=torch.bfloat16, device=dev)
        for li, (blk, hw) in enumerate(zip(m.blocks, m._hip)):
            for i in range(n_shards):
                with torch.cuda.stream(streams[i % n_streams]):
                    xi = x[bounds[i]:bounds[i + 1]]
                    a = ops.attention(hw["qkv"](xi), blk.attn.num_heads, q_prescaled=True)
                    hw["proj"](a, ops.EPI_RESIDUAL, residual=xi, out=xi)
                    hw["mlp"](xi)
        for i in range(n_shards):
            with torch.cuda.stream(streams[i % n_streams]):
                ops.layernorm_drop_first(x[bounds[i]:bounds[i + 1]], m.norm.weight, m.norm.bias, m.norm.eps, out=out[bounds[i]:bounds[i + 1]])
        for s in streams:
            ev = torch.cuda.Event()
            ev.record(s)
            cur.wait_event(ev)
        return out

    return go


import time
for n_shards, n_streams in ((2, 2), (3, 3), (4, 2), (4, 4), (6, 2), (6, 3), (8, 2), (8, 4), (5, 5)):
    go = run_cfg(n_shards, n_streams)
    t = timeit(lambda: go(x0))
    t0 = time.time()
    for _ in range(5):
        go(x0)
    host = (time.time() - t0) / 5 * 1e3
    torch.cuda.synchronize()
    print(f"{n_shards} shards on {n_streams} streams: {t:.3f} ms   (host enqueue time {host:.2f} ms)")
