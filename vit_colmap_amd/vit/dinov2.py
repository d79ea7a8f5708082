"""DINOv2 ViT patch-token forward: the module definition (hub parameter names), the weight handling and the
bf16 GPU paths over the hand-written kernels of csrc/ (gemm.hip, attention.hip, vit_ops.hip).

The reference obtains this network with `torch.hub.load("facebookresearch/dinov2", name)`
(vit_colmap/features/vit_extractor.py:86-104) and reads `x_norm_patchtokens` from
`forward_features` (:135-146).  There is no network here, so the architecture is defined locally
with the hub checkpoint's parameter names (a DINOv2 state dict loads with `load_state_dict`),
and `forward_patch_tokens` returns the same quantity: post-final-LayerNorm patch tokens
(B, Hp*Wp, C), token index = y*Wp + x.

MI355X-first choices
  * the image is consumed already patchified, (B, Hp*Wp, 3*14*14) or zero padded to 640: the HIP preprocessing
    kernel (csrc/preprocess.hip) writes that layout directly, so the patch embedding is ONE GEMM instead of a
    strided 14x14 convolution;
  * ViT-S (`_blocks_hip`): patch + position embedding, LN1+qkv, proj+residual and the whole MLP
    (LN2+fc1+GELU+fc2+residual, the hidden tensor never leaves the chip) are hand-written bf16 MFMA kernels with fused
    prologues / epilogues, the attention a hand-written flash kernel — four kernels per block, no library GEMM and no
    standalone elementwise pass;
  * ViT-B / ViT-L (`_blocks_gemm`): the staged 128x128 MFMA GEMM (vc_linear_bf16) with bias / GELU / residual epilogues,
    the LayerNorm and attention kernels — hand-written as well; only the SwiGLU giant (`_blocks_fused`) still uses
    `F.linear`; plain `nn.Module` path (`Block.forward`, SDPA) on the CPU / in float32 — the oracle-side evaluation;
  * LayerScale is folded into the projection weights at load time, LayerNorm gamma / beta into the GEMM that
    follows (`prepare_hip`);
  * the forward is shape-static per batch size (no data-dependent control flow on the host).

Architecture facts [recalled from the DINOv2 repository; pinned by tests/test_vit.py against the
`transformers` Dinov2 implementation that ships in this image]: pre-norm blocks
x += ls1(attn(norm1 x)); x += ls2(mlp(norm2 x)); LayerNorm eps 1e-6; exact (erf) GELU;
bicubic position-embedding interpolation with the 0.1 scale-factor offset of the hub models.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

PATCH = 14


@dataclass(frozen=True)
class Arch:
    dim: int
    depth: int
    heads: int
    ffn: str = "mlp"            # "mlp" | "swiglu"
    registers: int = 0
    img_size: int = 518


DINOV2_ARCHS = {
    "dinov2_vits14": Arch(384, 12, 6),
    "dinov2_vitb14": Arch(768, 12, 12),
    "dinov2_vitl14": Arch(1024, 24, 16),
    "dinov2_vitg14": Arch(1536, 40, 24, ffn="swiglu"),
    "dinov2_vits14_reg": Arch(384, 12, 6, registers=4),
    "dinov2_vitb14_reg": Arch(768, 12, 12, registers=4),
    "dinov2_vitl14_reg": Arch(1024, 24, 16, registers=4),
    "dinov2_vitg14_reg": Arch(1536, 40, 24, ffn="swiglu", registers=4),
}


class PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=PATCH, stride=PATCH)  # checkpoint-compatible parameter shapes

    def forward_patches(self, patches):
        """patches (B, N, 3*14*14) in (c, dy, dx) order -> (B, N, dim): the convolution as a GEMM."""
        w = self.proj.weight.reshape(self.proj.weight.shape[0], -1)
        return F.linear(patches, w, self.proj.bias)


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)

    def forward(self, x):
        B, N, C = x.shape
        if x.is_cuda and x.dtype == torch.bfloat16 and C // self.num_heads == 64 and not self.training:
            from .hip_ops import attention  # hand-written flash attention (csrc/attention.hip)

            return self.proj(attention(self.qkv(x), self.num_heads))
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(o.transpose(1, 2).reshape(B, N, C))


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class SwiGLUFFNFused(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        hidden = (int(hidden * 2 / 3) + 7) // 8 * 8
        self.w12 = nn.Linear(dim, 2 * hidden)
        self.w3 = nn.Linear(hidden, dim)

    def forward(self, x):
        x1, x2 = self.w12(x).chunk(2, dim=-1)
        return self.w3(F.silu(x1) * x2)


class LayerScale(nn.Module):
    def __init__(self, dim, init=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class Block(nn.Module):
    def __init__(self, arch: Arch):
        super().__init__()
        d = arch.dim
        self.norm1 = nn.LayerNorm(d, eps=1e-6)
        self.attn = Attention(d, arch.heads)
        self.ls1 = LayerScale(d)
        self.norm2 = nn.LayerNorm(d, eps=1e-6)
        self.mlp = Mlp(d, 4 * d) if arch.ffn == "mlp" else SwiGLUFFNFused(d, 4 * d)
        self.ls2 = LayerScale(d)
        self.folded = False

    def forward(self, x):
        if self.folded:  # LayerScale already multiplied into attn.proj / the last MLP layer
            x = x + self.attn(self.norm1(x))
            return x + self.mlp(self.norm2(x))
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class DinoV2(nn.Module):
    def __init__(self, arch: Arch, interpolate_offset: float = 0.1):
        super().__init__()
        self.arch = arch
        self.interpolate_offset = interpolate_offset
        d = arch.dim
        m = arch.img_size // PATCH
        self.patch_embed = PatchEmbed(d)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, d))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + m * m, d))
        self.register_tokens = nn.Parameter(torch.zeros(1, arch.registers, d)) if arch.registers else None
        self.mask_token = nn.Parameter(torch.zeros(1, d))  # present in checkpoints, unused at inference
        self.blocks = nn.ModuleList([Block(arch) for _ in range(arch.depth)])
        self.norm = nn.LayerNorm(d, eps=1e-6)
        self._pos_cache = {}
        self._hip = None    # per-block XsLinear operands (prepare_hip), False when the architecture is not covered

    # -- weights --------------------------------------------------------------------------------
    def load_state_dict(self, *args, **kwargs):
        self._pos_cache = {}        # cached position embeddings / class-token rows belong to the old weights
        return super().load_state_dict(*args, **kwargs)

    @torch.no_grad()
    def init_random(self, seed: int = 0):
        """Seeded random weights of trained-checkpoint scale (there is no network for the real ones)."""
        self._pos_cache = {}
        g = torch.Generator().manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("gamma"):
                p.fill_(1.0)
            elif p.dim() == 1:
                if "norm" in name and name.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()
            else:
                fan_in = p[0].numel() if p.dim() > 1 else p.numel()
                std = 0.02 if name in ("cls_token", "pos_embed", "register_tokens", "mask_token") else 1.0 / math.sqrt(fan_in)
                p.copy_(torch.randn(p.shape, generator=g) * std)
        return self

    @torch.no_grad()
    def fold_layerscale(self):
        """x*gamma after a Linear == Linear with rows of W and b scaled by gamma (exact in real
        arithmetic; done in float32 before any cast to bf16)."""
        for b in self.blocks:
            if b.folded:
                continue
            last = b.mlp.fc2 if isinstance(b.mlp, Mlp) else b.mlp.w3
            for lin, ls in ((b.attn.proj, b.ls1), (last, b.ls2)):
                lin.weight.mul_(ls.gamma[:, None])
                lin.bias.mul_(ls.gamma)
            b.folded = True
        return self

    # -- position embedding -----------------------------------------------------------------------
    def interpolated_pos_embed(self, hp: int, wp: int):
        """(1, 1 + hp*wp, C): bicubic resize of the learned M x M grid (cached per grid size).
        With interpolate_offset the scale factor (hp + 0.1) / M is handed to `interpolate`, as the
        hub models do; 0 switches to size-based interpolation (the `transformers` behaviour)."""
        key = (hp, wp, self.pos_embed.dtype, self.pos_embed.device)
        if key in self._pos_cache:
            return self._pos_cache[key]
        pe = self.pos_embed.float()
        n = pe.shape[1] - 1
        m = int(math.sqrt(n))
        if hp == m and wp == m:
            out = self.pos_embed
        else:
            grid = pe[:, 1:].reshape(1, m, m, -1).permute(0, 3, 1, 2)
            if self.interpolate_offset:
                sf = (float(hp + self.interpolate_offset) / m, float(wp + self.interpolate_offset) / m)
                grid = F.interpolate(grid, scale_factor=sf, mode="bicubic", antialias=False)
            else:
                grid = F.interpolate(grid, size=(hp, wp), mode="bicubic", align_corners=False)
            assert grid.shape[-2:] == (hp, wp)
            out = torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, hp * wp, -1)], dim=1).to(self.pos_embed.dtype)
        self._pos_cache[key] = out
        return out

    # -- forward ------------------------------------------------------------------------------------
    def forward_patch_tokens(self, patches: torch.Tensor, hp: int, wp: int) -> torch.Tensor:
        """patches (B, hp*wp, 588) -> x_norm_patchtokens (B, hp*wp, C)."""
        B = patches.shape[0]
        if patches.shape[-1] == 640:
            # padded patches (preprocess layout "patches_pad"): patch embedding, bias and position embedding in
            # one hand-written GEMM (csrc/gemm.hip); ViT-S bf16 GPU path only
            if not (patches.is_cuda and patches.dtype == torch.bfloat16 and self.register_tokens is None and not self.training):
                raise ValueError("padded patches are only accepted by the bf16 GPU path")
            if getattr(self, "_hip", None) is None:
                self.prepare_hip()
            if not self._hip:
                raise ValueError("padded patches need prepare_hip() (ViT-S)")
            from . import hip_ops as ops

            ckey = ("hip", hp, wp, patches.device)
            cached = self._pos_cache.get(ckey)
            if cached is None:      # position embedding as the kernel wants it + the class-token row (cls + pos[0]), once per grid
                pos = self.interpolated_pos_embed(hp, wp).to(torch.bfloat16).contiguous()
                cached = (pos, (self.cls_token[0, 0].float() + pos[0, 0].float()).to(torch.bfloat16))
                self._pos_cache[ckey] = cached
            pos, cls_row = cached
            x = torch.empty((B, 1 + hp * wp, self.arch.dim), dtype=torch.bfloat16, device=patches.device)
            ops.patch_embed(patches, self._pe_w, self.patch_embed.proj.bias, pos, x)
            x[:, 0] = cls_row
            return self._blocks_hip(x)
        x = self.patch_embed.forward_patches(patches)
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], dim=1) + self.interpolated_pos_embed(hp, wp)
        if self.register_tokens is not None:
            x = torch.cat([x[:, :1], self.register_tokens.expand(B, -1, -1), x[:, 1:]], dim=1)
        if x.is_cuda and x.dtype == torch.bfloat16 and all(b.folded for b in self.blocks) and not self.training:
            x = self._blocks_fused(x.contiguous())
        else:
            for blk in self.blocks:
                x = blk(x)
            x = self.norm(x)
        return x[:, 1 + self.arch.registers:]

    @torch.no_grad()
    def prepare_hip(self):
        """Build the x-stationary GEMM operands (csrc/gemm.hip) of every block from the CURRENT parameters:
        qkv with norm1 folded in, proj, fc1 with norm2 folded in.  Call it while the parameters are still
        float32 (after fold_layerscale, on the GPU) so gamma is folded before the one rounding to bf16;
        `_blocks_fused` calls it lazily otherwise.  ViT-S gets the x-stationary / fused-MLP operands, ViT-B / ViT-L bf16
        weights for the staged GEMM (q rows pre-scaled); the SwiGLU giant is not covered (`_hip = False`)."""
        import os

        from .hip_ops import FusedMlp, XsLinear
        from .hip_ops import gelu_table as hip_ops_gelu_table

        self._hip = False
        if self.arch.ffn != "mlp" or self.arch.dim % 128 != 0 or not all(b.folded for b in self.blocks):
            return self
        if not self.pos_embed.is_cuda:
            raise RuntimeError("prepare_hip needs the model on the GPU")
        w = self.patch_embed.proj.weight.detach().float().reshape(self.arch.dim, -1)
        wp = torch.zeros(self.arch.dim, 640, dtype=torch.float32, device=w.device)
        wp[:, : w.shape[1]] = w
        self._pe_w = wp.to(torch.bfloat16).contiguous()          # conv weight as a [C][640] GEMM operand (zero padded K)
        if self.arch.dim != 384:
            # ViT-B / ViT-L: every Linear on the staged 128x128 MFMA kernel (csrc/gemm.hip, vc_linear_bf16) with bias /
            # GELU / residual epilogues, the softmax scale folded into the q rows in float32 — no library GEMM either
            bf = lambda t: t.detach().float().to(torch.bfloat16).contiguous()
            self._hip = [dict(
                kind="gemm",
                qkv=tuple(bf(t) for t in _prescale_q(b.attn.qkv.weight, b.attn.qkv.bias, self.arch.dim)),
                proj=(bf(b.attn.proj.weight), bf(b.attn.proj.bias)),
                fc1=(bf(b.mlp.fc1.weight), bf(b.mlp.fc1.bias)),
                fc2=(bf(b.mlp.fc2.weight), bf(b.mlp.fc2.bias)),
            ) for b in self.blocks]
            return self
        self._gelu_tab = hip_ops_gelu_table(w.device)
        self._use_fused_mlp = os.environ.get("VITCOLMAP_FUSED_MLP", "1") == "1"   # developer A/B switch: 0 = fc1 and fc2 as two kernels
        self._hip = [dict(
            qkv=XsLinear(*_prescale_q(b.attn.qkv.weight, b.attn.qkv.bias, self.arch.dim), b.norm1.weight, b.norm1.bias, b.norm1.eps),
            proj=XsLinear(b.attn.proj.weight, b.attn.proj.bias),
            fc1=XsLinear(b.mlp.fc1.weight, b.mlp.fc1.bias, b.norm2.weight, b.norm2.bias, b.norm2.eps),
            mlp=FusedMlp(b.mlp.fc1.weight, b.mlp.fc1.bias, b.norm2.weight, b.norm2.bias, b.mlp.fc2.weight, b.mlp.fc2.bias,
                         b.norm2.eps) if self._use_fused_mlp else None,
        ) for b in self.blocks]
        return self

    # -- batch shards on HIP streams ------------------------------------------------------------------
    # Every kernel of a block is a persistent grid of one workgroup per CU (or two): 599 row tiles of the fused MLP run as
    # 3 rounds for 2.34 rounds of work, 1800 attention units as 4 for 3.5 — a fifth of the MLP's and an eighth of the
    # attention's time is a tail in which most CUs idle.  Images are independent through the whole block stack, so the
    # batch is cut into `batch_shards` contiguous shards whose block loops are enqueued layer by layer on separate HIP
    # streams: one shard's tail overlaps the other's next kernel (measured on 50 x 1531 x 384: 7.12 -> 6.66 ms with two
    # equal shards, 6.9 with three; tools/bench_vit_streams.py).  Per-row arithmetic is unchanged; what can differ from
    # the single-stream run is which rows share a 32-row tile, hence which tiles take the float GELU path of the fused
    # MLP instead of the table (both are within the bf16 tolerance asserted in tests/test_vit_gpu.py).
    batch_shards = None   # None: VITCOLMAP_VIT_SHARDS or 2; 1 switches the streams off

    def _shard_plan(self, x):
        import os

        k = self.batch_shards if self.batch_shards is not None else int(os.environ.get("VITCOLMAP_VIT_SHARDS", "2"))
        B = x.shape[0]
        if k <= 1 or B < 8 * k:
            return None
        bounds = [B * i // k for i in range(k + 1)]
        key = (x.device, k)
        if getattr(self, "_shard_streams", None) is None or self._shard_streams[0] != key:
            self._shard_streams = (key, [torch.cuda.Stream(device=x.device) for _ in range(k - 1)])
        return bounds, self._shard_streams[1]

    def _run_sharded(self, x, layer_fn, final_fn, out):
        """layer_fn(i_layer, x_shard) updates a shard in place; final_fn(x_shard, out_shard) writes the normalised rows."""
        plan = self._shard_plan(x)
        if plan is None:
            for i in range(len(self.blocks)):
                layer_fn(i, x)
            final_fn(x, out)
            return out
        bounds, side = plan
        cur = torch.cuda.current_stream(x.device)
        streams = [cur] + side
        ready = torch.cuda.Event()
        ready.record(cur)
        for s in side:
            s.wait_event(ready)
        xs = [x[bounds[i]:bounds[i + 1]] for i in range(len(streams))]
        outs = [out[bounds[i]:bounds[i + 1]] for i in range(len(streams))]
        for li in range(len(self.blocks)):
            for s, xi in zip(streams, xs):
                with torch.cuda.stream(s):
                    layer_fn(li, xi)
        for s, xi, oi in zip(streams, xs, outs):
            with torch.cuda.stream(s):
                final_fn(xi, oi)
        for s in side:   # join: everything the side streams touched is complete before the caller's stream goes on
            done = torch.cuda.Event()
            done.record(s)
            cur.wait_event(done)
        return out

    def _blocks_hip(self, x, drop_cls: bool = True):
        """ViT-S bf16 path on the hand-written GEMMs: LayerNorm lives in the x load of qkv / fc1, GELU and
        both residual adds in GEMM epilogues; per block 4 kernels (the MLP is one) and no standalone elementwise pass.
        x (B, 1 + T, C) incl. the class token -> normalised patch tokens (B, T, C), or all rows with drop_cls=False."""
        from . import hip_ops as ops

        if self._hip and self._hip[0].get("kind") == "gemm":
            return self._blocks_gemm(x, drop_cls)
        blocks, hip = list(self.blocks), self._hip

        def layer(i, xi):
            blk, hw = blocks[i], hip[i]
            a = ops.attention(hw["qkv"](xi), blk.attn.num_heads, q_prescaled=True)   # LN1 + qkv (q pre-scaled), flash attention
            hw["proj"](a, ops.EPI_RESIDUAL, residual=xi, out=xi)            # x += proj(a)
            if hw["mlp"] is not None:
                hw["mlp"](xi)                                               # x += fc2(gelu(fc1(LN2 x))), one kernel
                return
            hdn = hw["fc1"](xi, ops.EPI_GELU, gelu_table=self._gelu_tab)    # gelu(fc1(LN2 x)), GELU by LDS table
            fc2 = blk.mlp.fc2
            ops.linear(hdn, fc2.weight, fc2.bias, ops.EPI_RESIDUAL, residual=xi, out=xi)   # x += fc2(hdn)

        return self._run_sharded(x, layer, *self._final_norm(x, drop_cls))

    def _final_norm(self, x, drop_cls):
        """-> (final_fn, out): the last LayerNorm as a per-shard step writing into a batch slice of `out`."""
        from . import hip_ops as ops

        B, N, C = x.shape
        if drop_cls:   # final norm of the patch tokens only: the class-token row is dropped here, the caller gets a dense tensor
            out = torch.empty((B, N - 1, C), dtype=torch.bfloat16, device=x.device)
            return (lambda xi, oi: ops.layernorm_drop_first(xi, self.norm.weight, self.norm.bias, self.norm.eps, out=oi)), out
        out = torch.empty_like(x)

        def all_rows(xi, oi):
            _, h = ops.add_layernorm(xi, None, self.norm.weight, self.norm.bias, self.norm.eps)
            oi.copy_(h)

        return all_rows, out

    def _blocks_gemm(self, x, drop_cls: bool = True):
        """ViT-B / ViT-L bf16 path: LayerNorm kernel + staged MFMA GEMMs with fused epilogues (bias; exact GELU; bias +
        residual written in place on the residual stream) + the flash attention kernel; six launches per block, none of
        them a library call.  x (B, 1 + T, C) -> normalised patch tokens (B, T, C), or all rows with drop_cls=False."""
        from . import hip_ops as ops

        blocks, hip = list(self.blocks), self._hip

        def layer(i, xi):
            blk, hw = blocks[i], hip[i]
            _, h = ops.add_layernorm(xi, None, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
            a = ops.attention(ops.linear(h, *hw["qkv"]), blk.attn.num_heads, q_prescaled=True)
            ops.linear(a, *hw["proj"], ops.EPI_RESIDUAL, residual=xi, out=xi)          # x += proj(a)
            _, h = ops.add_layernorm(xi, None, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            hdn = ops.linear(h, *hw["fc1"], ops.EPI_GELU)
            ops.linear(hdn, *hw["fc2"], ops.EPI_RESIDUAL, residual=xi, out=xi)        # x += fc2(gelu(fc1(LN2 x)))

        return self._run_sharded(x, layer, *self._final_norm(x, drop_cls))

    def _blocks_fused(self, x):
        """bf16 GPU path: every residual add is fused with the LayerNorm that follows it
        (csrc/vit_ops.hip), including the one that crosses into the next block / the final norm."""
        from .hip_ops import add_layernorm

        if getattr(self, "_hip", None) is None:
            self.prepare_hip()
        if self._hip:
            return self._blocks_hip(x, drop_cls=False)
        blocks = list(self.blocks)
        _, h = add_layernorm(x, None, blocks[0].norm1.weight, blocks[0].norm1.bias, 1e-6)
        for i, blk in enumerate(blocks):
            a = blk.attn(h)
            x, h = add_layernorm(a, x, blk.norm2.weight, blk.norm2.bias, 1e-6)
            m = blk.mlp(h)
            last = i + 1 == len(blocks)
            nxt = self.norm if last else blocks[i + 1].norm1
            x, h = add_layernorm(m, x, nxt.weight, nxt.bias, 1e-6, want_sum=not last)
        return h

    def forward_features(self, image: torch.Tensor):
        """(B, 3, H, W) normalised image -> dict with 'x_norm_patchtokens', the key the reference
        reads (vit_extractor.py:140-142).  H and W must be multiples of 14."""
        B, _, H, W = image.shape
        hp, wp = H // PATCH, W // PATCH
        patches = image.reshape(B, 3, hp, PATCH, wp, PATCH).permute(0, 2, 4, 1, 3, 5).reshape(B, hp * wp, 3 * PATCH * PATCH)
        return {"x_norm_patchtokens": self.forward_patch_tokens(patches, hp, wp)}


def _prescale_q(weight, bias, dim):
    """qkv projection with the softmax scale and log2(e) folded into the q rows (float32, before any rounding):
    what csrc/attention.hip's `q_prescaled` mode consumes."""
    from .hip_ops import Q_PRESCALE

    w = weight.detach().float().clone()
    b = bias.detach().float().clone()
    w[:dim] *= Q_PRESCALE
    b[:dim] *= Q_PRESCALE
    return w, b


def build_dinov2(model_name: str = "dinov2_vitb14", interpolate_offset: float = 0.1) -> DinoV2:
    if model_name not in DINOV2_ARCHS:
        raise ValueError(f"Unsupported model: {model_name}. Currently only DINOv2 models are supported.")
    return DinoV2(DINOV2_ARCHS[model_name], interpolate_offset)


def load_dinov2_weights(model: DinoV2, path: str) -> DinoV2:
    """Load a DINOv2 checkpoint (hub parameter names) from .safetensors or a weights-only .pth."""
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file

        sd = load_file(str(path))
    else:
        sd = torch.load(str(path), map_location="cpu", weights_only=True)
        for key in ("model_state_dict", "state_dict"):
            if isinstance(sd, dict) and key in sd:
                sd = sd[key]
    try:
        missing, unexpected = model.load_state_dict(sd, strict=False)
    except RuntimeError as e:  # shape mismatch: wrong architecture for this checkpoint
        raise ValueError(f"checkpoint does not fit {model.arch}: {str(e)[:300]}") from None
    missing = [k for k in missing if k != "mask_token"]
    if missing or unexpected:
        raise ValueError(f"checkpoint does not fit {model.arch}: missing {missing[:5]}, unexpected {unexpected[:5]}")
    return model
