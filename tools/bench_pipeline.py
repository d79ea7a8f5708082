#!/usr/bin/env python3
"""Developer tool (experiment): the bench step with the NEXT batch's front end (preprocessing + patch embedding) issued on a
second stream behind the current batch's transformer blocks, so that it runs beside the current batch's low-occupancy tail
(selection, descriptor gather, prepare, matching).  Prints ms per step for the sequential and the pipelined order."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vit_colmap_amd.features.vit_extractor import ViTExtractor, PATCH
from vit_colmap_amd.features import hip_preprocess, hip_select
from vit_colmap_amd.matching import match_pairs, prepare_descriptors, exhaustive_pairs
from vit_colmap_amd.vit import hip_ops as ops
B, K = 50, int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
ex = ViTExtractor(model_name="dinov2_vits14", num_keypoints=512, descriptor_dim=384, device="cuda:0", precision="bf16", seed=0)
frames = torch.randint(0, 255, (B, 480, 640, 3), dtype=torch.uint8, device=dev)
model = ex.model
h, w = 480, 640
hp, wp = h // PATCH, w // PATCH
h_new, w_new = hp * PATCH, wp * PATCH
ex.extract_device(frames)          # builds the kernel operands
pairs = exhaustive_pairs(B, "cuda")
out_m = torch.empty((pairs.shape[0], 512, 2), dtype=torch.int32, device=dev)
out_c = torch.empty((pairs.shape[0],), dtype=torch.int32, device=dev)
pos = model.interpolated_pos_embed(hp, wp).to(torch.bfloat16).contiguous()
cls_row = (model.cls_token[0, 0].float() + pos[0, 0].float()).to(torch.bfloat16)

def front():
    patches = hip_preprocess.preprocess(frames, out_dtype=torch.bfloat16, layout="patches_pad")
    x = torch.empty((B, 1 + hp * wp, model.arch.dim), dtype=torch.bfloat16, device=dev)
    ops.patch_embed(patches, model._pe_w, model.patch_embed.proj.bias, pos, x)
    x[:, 0] = cls_row
    return x

def blocks(x):
    return model._blocks_hip(x).contiguous()

def tail(tokens):
    res = hip_select.dense_to_sparse(tokens, hp, wp, (w, h), (w_new, h_new), 512, ex.detection_method, None)
    prepared = prepare_descriptors(res["desc_u8"], res["count"])
    match_pairs(prepared, res["count"], B, 512, 384, pairs, out_matches=out_m, out_counts=out_c)

def sequential(n):
    for _ in range(n):
        tail(blocks(front()))

def pipelined(n):
    main = torch.cuda.current_stream()
    s2 = torch.cuda.Stream()
    x_next = front()
    ev_b, ev_f = torch.cuda.Event(), torch.cuda.Event()
    for k in range(n):
        x = x_next
        tokens = blocks(x)
        ev_b.record(main)
        if k + 1 < n:
            with torch.cuda.stream(s2):
                s2.wait_event(ev_b)
                x_next = front()
                x_next.record_stream(main)
                ev_f.record(s2)
        tail(tokens)
        if k + 1 < n:
            main.wait_event(ev_f)

with torch.inference_mode():
    for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
        fn(5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(K)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
        print(f"{name:11s} {ms:.3f} ms per step = {B / ms * 1e3:.0f} images/s")
