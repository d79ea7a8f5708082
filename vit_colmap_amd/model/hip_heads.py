"""The convolutional heads of ViTFeatureModel on the hand-written implicit-GEMM convolution (csrc/gemm.hip
`gemm256_kernel<EPI, CONV>`, C entry `vc_conv_taps_bf16`) instead of MIOpen — reference
vit_colmap/model/vit_feature_model.py:12-29 (UpsampleBlock), :89-120 (upsampler, trunk, heads), :231-293 (forward).

Everything stays channels-last: the backbone's token grid (B, hp * wp, C) IS an image batch [B][hp][wp][C], and every layer
reads and writes [pixels][channels] rows (one spare zero row behind each batch for taps outside the image).
  * ConvTranspose2d(4, stride 2, pad 1): output pixel (2y + i, 2x + j) only sees input pixels (y + i - 1 + ty, x + j - 1 + tx),
    ty, tx in {0, 1}, through kernel element (3 - 2 ty - i, 3 - 2 tx - j): four 2 x 2-tap products, one per parity (i, j), whose
    results are interleaved by the kernel's epilogue (`out_parity`; weights: `deconv_class_matrices`);
  * Conv2d 3 x 3 pad 1 (+ the eval BatchNorm folded in by `fold_batchnorm`, + GELU): one 9-tap product with the GELU in
    the epilogue; the two heads' first convolutions (256 -> 64 and 256 -> 128) run as ONE product padded to 256 outputs;
  * the heads' last 1 x 1 convolutions (64 -> 4, 128 -> D) are two small library GEMMs on slices of that result.
Measured (batch 8 of 640 x 480, tools/bench_conv.py): the six big layers 2.5 ms against 6.7 ms on MIOpen's bf16 kernels.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from ..vit import hip_ops as ops


MAX_ACTIVATION_BYTES = 3.2e9      # per activation tensor of one vc_conv_taps_bf16 call (32-bit offsets, margin for the tap shifts)


def conv3x3_matrix(weight: torch.Tensor) -> torch.Tensor:
    """Conv2d weight [N][C][3][3] -> [N][(ky, kx, c)]: k order of vc_conv_taps_bf16 with (kh, kw, dy0, dx0) = (3, 3, -1, -1)."""
    n = weight.shape[0]
    return weight.detach().permute(0, 2, 3, 1).reshape(n, -1).contiguous()


def deconv_class_matrices(weight: torch.Tensor):
    """ConvTranspose2d(kernel 4, stride 2, padding 1) weight [C_in][C_out][4][4] -> four (matrix [C_out][(ty, tx, c)], i, j,
    dy0, dx0): output pixels (2y + i, 2x + j) = the 2 x 2-tap product with taps starting at (y + dy0, x + dx0)."""
    assert weight.shape[2:] == (4, 4)
    c_out = weight.shape[1]
    out = []
    for i in (0, 1):
        for j in (0, 1):
            ky = [3 - 2 * t - i for t in (0, 1)]
            kx = [3 - 2 * t - j for t in (0, 1)]
            w = weight.detach()[:, :, ky][:, :, :, kx]                    # [C_in][C_out][ty][tx]
            out.append((w.permute(1, 2, 3, 0).reshape(c_out, -1).contiguous(), i, j, i - 1, j - 1))
    return out


class HipHeads:
    """Built once from a ViTFeatureModel whose BatchNorms are folded (`fold_batchnorm`); call with the bf16 token grid."""

    def __init__(self, model):
        bf = torch.bfloat16
        self.blocks = []
        for blk in model.upsampler:
            if not isinstance(blk.bn, torch.nn.Identity):
                raise ValueError("HipHeads needs fold_batchnorm() first")
            classes = [(m.to(bf), i, j, dy0, dx0) for m, i, j, dy0, dx0 in deconv_class_matrices(blk.deconv.weight)]
            self.blocks.append(dict(classes=classes, deconv_bias=blk.deconv.bias.detach().to(bf).contiguous(),
                                    conv_w=conv3x3_matrix(blk.conv.weight).to(bf), conv_b=blk.conv.bias.detach().to(bf).contiguous(),
                                    c_out=blk.conv.weight.shape[0]))
        self.trunk_w = conv3x3_matrix(model.trunk[0].weight).to(bf)
        self.trunk_b = model.trunk[0].bias.detach().to(bf).contiguous()
        kc, dc = model.keypoint_head[0], model.descriptor_head[0]
        self.n_kp, self.n_desc = kc.weight.shape[0], dc.weight.shape[0]               # 64, 128
        n_pad = -(-(self.n_kp + self.n_desc) // 256) * 256
        hw = torch.zeros((n_pad, self.trunk_w.shape[0] * 9), dtype=bf, device=self.trunk_w.device)
        hb = torch.zeros((n_pad,), dtype=bf, device=hw.device)
        hw[: self.n_kp] = conv3x3_matrix(kc.weight).to(bf)
        hw[self.n_kp: self.n_kp + self.n_desc] = conv3x3_matrix(dc.weight).to(bf)
        hb[: self.n_kp] = kc.bias.detach().to(bf)
        hb[self.n_kp: self.n_kp + self.n_desc] = dc.bias.detach().to(bf)
        self.heads_w, self.heads_b = hw, hb
        k1, d1 = model.keypoint_head[3], model.descriptor_head[3]
        self.kp_w = k1.weight.detach().reshape(k1.weight.shape[0], -1).to(bf).contiguous()
        self.kp_b = k1.bias.detach().to(bf)
        self.desc_w = d1.weight.detach().reshape(d1.weight.shape[0], -1).to(bf).contiguous()
        self.desc_b = d1.bias.detach().to(bf)

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor, hp: int, wp: int, target_size: Optional[Tuple[int, int]] = None) -> Dict[str, torch.Tensor]:
        """tokens (B, hp * wp, C) bf16 -> the dict ViTFeatureModel.forward_from_backbone_features returns (NCHW-shaped
        tensors; descriptors and features are channels-last views)."""
        B, n_tok, c = tokens.shape
        assert n_tok == hp * wp and tokens.dtype == torch.bfloat16 and tokens.is_cuda
        # vc_conv_taps_bf16 addresses a batch with 32-bit offsets (< 4 GiB per activation tensor): large images go through in
        # batch chunks (1600 x 1200: 160 MB per image at the x4 grid, 20 images per chunk)
        widest = max([c] + [blk["c_out"] for blk in self.blocks])
        b_max = max(1, int(MAX_ACTIVATION_BYTES // (16 * hp * wp * widest * 2)))
        if B > b_max:
            parts = [self(tokens[i:i + b_max], hp, wp, target_size) for i in range(0, B, b_max)]
            return {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}
        dev = tokens.device
        H, W = hp, wp
        x = ops.conv_rows(B, H, W, c, dev)
        x[: B * H * W].copy_(tokens.reshape(B * H * W, c))
        for blk in self.blocks:
            rows, co = B * H * W, blk["c_out"]
            up = ops.conv_rows(B, 2 * H, 2 * W, co, dev)
            for m, i, j, dy0, dx0 in blk["classes"]:                                           # each call writes its parity's pixels
                ops.conv_taps(x, m, blk["deconv_bias"], B, H, W, 2, 2, dy0, dx0, ops.EPI_BIAS, out=up, out_parity=2 * i + j)
            H, W = 2 * H, 2 * W
            x = ops.conv_rows(B, H, W, co, dev)
            ops.conv_taps(up, blk["conv_w"], blk["conv_b"], B, H, W, 3, 3, -1, -1, ops.EPI_GELU, out=x)
            c = co
        if target_size is None:
            target_size = ((hp * 14) // 4, (wp * 14) // 4)
        Ht, Wt = int(target_size[0]), int(target_size[1])
        if (H, W) != (Ht, Wt):
            grid = x[: B * H * W].view(B, H, W, c).permute(0, 3, 1, 2)                        # channels-last view
            grid = F.interpolate(grid, size=(Ht, Wt), mode="bilinear", align_corners=False)
            x = ops.conv_rows(B, Ht, Wt, c, dev)
            x[: B * Ht * Wt].view(B, Ht, Wt, c).copy_(grid.permute(0, 2, 3, 1))
            H, W = Ht, Wt
        rows = B * H * W
        trunk = ops.conv_rows(B, H, W, self.trunk_w.shape[0], dev)
        ops.conv_taps(x, self.trunk_w, self.trunk_b, B, H, W, 3, 3, -1, -1, ops.EPI_GELU, out=trunk)
        hh = ops.conv_taps(trunk, self.heads_w, self.heads_b, B, H, W, 3, 3, -1, -1, ops.EPI_GELU)
        kp = F.linear(hh[:, : self.n_kp], self.kp_w, self.kp_b).float()
        desc = F.linear(hh[:, self.n_kp: self.n_kp + self.n_desc], self.desc_w, self.desc_b).float()
        keypoints = kp.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        keypoints[:, 3] = torch.tanh(keypoints[:, 3]) * torch.pi
        descriptors = F.normalize(desc.view(B, H, W, -1).permute(0, 3, 1, 2), p=2, dim=1, eps=1e-8)
        features = trunk[:rows].view(B, H, W, -1).permute(0, 3, 1, 2)
        return {"keypoints": keypoints, "descriptors": descriptors, "features": features}
