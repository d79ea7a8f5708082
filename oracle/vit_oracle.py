"""CPU ORACLE (test infrastructure, NOT product code) — DINOv2 patch-token forward in float32.

The reference gets the network from torch.hub (`/root/reference/vit_colmap/features/
vit_extractor.py:86-104`, unreachable offline) and reads `x_norm_patchtokens` (:135-146).
This is a functional restatement of the published DINOv2 ViT forward over a plain state dict
(hub parameter names), independent of the product module in vit_colmap_amd/vit/dinov2.py.

Pinned by `tests/test_vit.py` against the `transformers` Dinov2 architecture that ships in this
image (random weights built from a config object; nothing is fetched).  Pretrained-weight parity
is impossible offline: PARITY on identical seeded random weights only.
"""
import math

import torch
import torch.nn.functional as F

PATCH = 14


def pos_embed_for(sd, hp, wp, interpolate_offset=0.1):
    pe = sd["pos_embed"].float()
    m = int(math.sqrt(pe.shape[1] - 1))
    if hp == m and wp == m:
        return pe
    grid = pe[:, 1:].reshape(1, m, m, -1).permute(0, 3, 1, 2)
    if interpolate_offset:
        grid = F.interpolate(grid, scale_factor=((hp + interpolate_offset) / m, (wp + interpolate_offset) / m),
                             mode="bicubic", antialias=False)
    else:
        grid = F.interpolate(grid, size=(hp, wp), mode="bicubic", align_corners=False)
    return torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, hp * wp, -1)], dim=1)


def forward_patch_tokens(sd, image, heads, interpolate_offset=0.1):
    """sd: DINOv2 state dict (float32), image (B, 3, H, W) normalised -> (B, Hp*Wp, C)."""
    B, _, H, W = image.shape
    hp, wp = H // PATCH, W // PATCH
    x = F.conv2d(image, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=PATCH)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], dim=1) + pos_embed_for(sd, hp, wp, interpolate_offset)
    n_reg = 0
    if "register_tokens" in sd:
        n_reg = sd["register_tokens"].shape[1]
        x = torch.cat([x[:, :1], sd["register_tokens"].expand(B, -1, -1), x[:, 1:]], dim=1)
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    C = x.shape[-1]
    hd = C // heads
    for i in range(depth):
        p = f"blocks.{i}."
        h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        qkv = h @ sd[p + "attn.qkv.weight"].T + sd[p + "attn.qkv.bias"]
        q, k, v = qkv.reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        att = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        o = (att @ v).transpose(1, 2).reshape(B, -1, C)
        o = o @ sd[p + "attn.proj.weight"].T + sd[p + "attn.proj.bias"]
        x = x + o * sd[p + "ls1.gamma"]
        h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        if p + "mlp.fc1.weight" in sd:
            h = F.gelu(h @ sd[p + "mlp.fc1.weight"].T + sd[p + "mlp.fc1.bias"])
            h = h @ sd[p + "mlp.fc2.weight"].T + sd[p + "mlp.fc2.bias"]
        else:
            a, b = (h @ sd[p + "mlp.w12.weight"].T + sd[p + "mlp.w12.bias"]).chunk(2, dim=-1)
            h = (F.silu(a) * b) @ sd[p + "mlp.w3.weight"].T + sd[p + "mlp.w3.bias"]
        x = x + h * sd[p + "ls2.gamma"]
    x = F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    return x[:, 1 + n_reg:]


def preprocess_bgr(image_bgr):
    """vit_extractor.py:117-132 for an image whose sides are already multiples of 14:
    BGR->RGB, /255, ImageNet mean/std, float32 NCHW."""
    import numpy as np

    rgb = image_bgr[:, :, ::-1].astype(np.float32) / np.float32(255.0)
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    x = (rgb - mean) / std
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))[None])
