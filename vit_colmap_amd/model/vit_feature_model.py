"""ViTFeatureModel — DINOv2 backbone + learned upsampling + keypoint / descriptor heads, inference side of the
reference's `vit_colmap/model/vit_feature_model.py:12-293` (same submodule names, so a trained reference state dict
loads: `backbone.*`, `upsampler.{0,1}.{deconv,conv,bn}`, `trunk.{0,1}`, `keypoint_head.{0,1,3}`, `descriptor_head.{0,1,3}`).

What runs where: the backbone is this package's DINOv2 (`vit/dinov2.py`: hand-written kernels for ViT-S / B / L); the
convolutional heads run, in the bf16 product path, on the hand-written implicit-GEMM convolution (`model/hip_heads.py`,
`vc_conv_taps_bf16`; round 3) — channels-last, so the token grid (B, Hp*Wp, C) of the backbone IS their input without a
transpose — and as PyTorch-ROCm (MIOpen) modules in float32 / on request; what follows the model (sigmoid, NMS, top-k,
sub-pixel keypoints, descriptor quantiser) is csrc/heatmap.hip.

Differences from the reference, deliberate: the backbone comes from `build_dinov2` (torch.hub is unreachable offline)
with seeded random weights unless a state dict is loaded; training-only members (`get_trainable_parameters`,
freezing) are kept for API compatibility but no training loop is part of this package."""
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..vit import build_dinov2


class UpsampleBlock(nn.Module):
    """ConvTranspose2d(4, stride 2, pad 1) + Conv3x3 + BatchNorm + GELU (vit_feature_model.py:12-29)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.deconv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=4, stride=2, padding=1)
        self.conv = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.bn = nn.BatchNorm2d(out_channels)
        self.activation = nn.GELU()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.activation(self.bn(self.conv(self.deconv(x))))


class ViTFeatureModel(nn.Module):
    def __init__(self, backbone_name: str = "dinov2_vitb14", descriptor_dim: int = 128, freeze_backbone: bool = True,
                 seed: int = 0):
        super().__init__()
        self.backbone_name = backbone_name
        self.descriptor_dim = descriptor_dim
        self.patch_size = 14
        self.backbone = build_dinov2(backbone_name).init_random(seed)      # vit_feature_model.py:63-66 loads the hub model
        self.backbone_dim = self.backbone.arch.dim                         # :69-77
        if freeze_backbone:                                                # :80-83
            for p in self.backbone.parameters():
                p.requires_grad = False
            self.backbone.eval()
        self.upsampler = nn.Sequential(UpsampleBlock(self.backbone_dim, 512), UpsampleBlock(512, 512))   # :89-94
        self.trunk = nn.Sequential(nn.Conv2d(512, 256, kernel_size=3, padding=1), nn.BatchNorm2d(256), nn.GELU())
        self.keypoint_head = nn.Sequential(nn.Conv2d(256, 64, kernel_size=3, padding=1), nn.BatchNorm2d(64), nn.GELU(),
                                           nn.Conv2d(64, 4, kernel_size=1))                               # :107-112
        self.descriptor_head = nn.Sequential(nn.Conv2d(256, 128, kernel_size=3, padding=1), nn.BatchNorm2d(128), nn.GELU(),
                                             nn.Conv2d(128, descriptor_dim, kernel_size=1))               # :115-120
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():                                              # seeded heads (the reference's are torch defaults)
            for name, p in self.named_parameters():
                if name.startswith("backbone."):
                    continue
                if p.dim() > 1:
                    fan_in = p[0].numel() if "deconv" not in name else p.shape[0] * p[0, 0].numel()
                    p.copy_(torch.randn(p.shape, generator=g) / fan_in ** 0.5)
                elif name.endswith("bias"):
                    p.zero_()
        print("ViTFeatureModel initialized:")
        print(f"  Backbone: {backbone_name} ({self.backbone_dim}D)")
        print(f"  Descriptor dim: {descriptor_dim}")
        print(f"  Backbone frozen: {freeze_backbone}")

    # ------------------------------------------------------------------------------------------
    def _extract_backbone_features(self, x: torch.Tensor) -> torch.Tensor:
        """(B, 3, H, W) normalised image -> (B, C, H/14, W/14) (vit_feature_model.py:127-172)."""
        B, _, H, W = x.shape
        assert H % self.patch_size == 0, f"Height {H} not divisible by patch_size {self.patch_size}"
        assert W % self.patch_size == 0, f"Width {W} not divisible by patch_size {self.patch_size}"
        hp, wp = H // self.patch_size, W // self.patch_size
        with torch.no_grad():
            tokens = self.backbone.forward_features(x)["x_norm_patchtokens"]
        return self.tokens_to_grid(tokens, hp, wp)

    @staticmethod
    def tokens_to_grid(tokens: torch.Tensor, hp: int, wp: int) -> torch.Tensor:
        """(B, hp*wp, C) -> (B, C, hp, wp) as a channels-last VIEW (no copy)."""
        B, _, C = tokens.shape
        return tokens.reshape(B, hp, wp, C).permute(0, 3, 1, 2)

    def forward(self, x: torch.Tensor, target_size: Optional[Tuple[int, int]] = None) -> Dict[str, torch.Tensor]:
        """vit_feature_model.py:174-229."""
        H, W = x.shape[2:]
        feats = self._extract_backbone_features(x)
        return self.forward_from_backbone_features(feats, target_size or (H // 4, W // 4))

    def forward_from_backbone_features(self, backbone_features: torch.Tensor,
                                       target_size: Optional[Tuple[int, int]] = None) -> Dict[str, torch.Tensor]:
        """vit_feature_model.py:231-293: upsample x4, bilinear resize to the 1/4-resolution target, trunk, heads,
        orientation = tanh * pi, descriptors L2-normalised over channels."""
        head_dtype = self.trunk[0].weight.dtype
        heads = getattr(self, "_hip_heads", None)
        if heads is not None and backbone_features.is_cuda and head_dtype == torch.bfloat16:
            B, C, hp, wp = backbone_features.shape
            tokens = backbone_features.permute(0, 2, 3, 1).reshape(B, hp * wp, C).to(torch.bfloat16)   # a view for a channels-last grid
            return heads(tokens.contiguous(), hp, wp, target_size)
        up = self.upsampler(backbone_features.to(head_dtype))
        if target_size is None:
            hp, wp = backbone_features.shape[2:]
            target_size = ((hp * self.patch_size) // 4, (wp * self.patch_size) // 4)
        if tuple(up.shape[2:]) != tuple(target_size):
            up = F.interpolate(up, size=tuple(target_size), mode="bilinear", align_corners=False)
        trunk = self.trunk(up)
        keypoints = self.keypoint_head(trunk).float().clone()
        descriptors = self.descriptor_head(trunk).float()
        keypoints[:, 3] = torch.tanh(keypoints[:, 3]) * torch.pi
        descriptors = F.normalize(descriptors, p=2, dim=1, eps=1e-8)
        return {"keypoints": keypoints, "descriptors": descriptors, "features": trunk}

    @torch.no_grad()
    def fold_batchnorm(self):
        """Inference only: every eval-mode BatchNorm follows a convolution, so its affine map goes into that convolution's
        weights and bias (w' = w g / sqrt(var + eps), b' = (b - mean) g / sqrt(var + eps) + beta; exact in real arithmetic,
        done in float32) and the BatchNorm becomes the identity — five passes over the largest activations less."""
        def fold(conv, bn):
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            conv.weight.mul_(scale[:, None, None, None])
            conv.bias.copy_((conv.bias - bn.running_mean) * scale + bn.bias)

        for blk in self.upsampler:
            fold(blk.conv, blk.bn)
            blk.bn = nn.Identity()
        for seq in (self.trunk, self.keypoint_head, self.descriptor_head):
            fold(seq[0], seq[1])
            seq[1] = nn.Identity()
        return self

    def prepare_hip_heads(self):
        """After fold_batchnorm() and the cast to bf16 on the GPU: run upsampler, trunk and heads on the hand-written
        implicit-GEMM convolution (model/hip_heads.py) instead of MIOpen.  `VITCOLMAP_HIP_HEADS=0` keeps the library path."""
        import os

        from .hip_heads import HipHeads

        self._hip_heads = HipHeads(self) if os.environ.get("VITCOLMAP_HIP_HEADS", "1") != "0" else None
        return self

    # ------------------------------------------------------------------------------------------
    def load_reference_state_dict(self, state_dict) -> None:
        """A state dict saved from the reference's ViTFeatureModel (hub backbone names under `backbone.`)."""
        back = {k[len("backbone."):]: v for k, v in state_dict.items() if k.startswith("backbone.")}
        heads = {k: v for k, v in state_dict.items() if not k.startswith("backbone.")}
        if back:
            missing, unexpected = self.backbone.load_state_dict(back, strict=False)
            missing = [k for k in missing if k != "mask_token"]
            if missing or unexpected:
                raise ValueError(f"backbone weights do not fit {self.backbone_name}: missing {missing[:5]}, unexpected {unexpected[:5]}")
        own = {k for k in self.state_dict() if not k.startswith("backbone.")}
        unknown = [k for k in heads if k not in own]
        if unknown:
            raise ValueError(f"unexpected head parameters: {unknown[:5]}")
        self.load_state_dict(heads, strict=False)      # the reference loads with strict=False as well (:108)

    def get_trainable_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    def count_parameters(self) -> Dict[str, int]:
        total = sum(p.numel() for p in self.parameters())
        trainable = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return {"total": total, "trainable": trainable, "frozen": total - trainable}
