"""GPU tests of the fused ViT glue kernels against PyTorch's own ops on the same device."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,C", [(1531 * 3, 384), (777, 768), (5, 1024), (64, 1536), (1, 64)])
def test_add_layernorm_matches_torch(rows, C):
    from vit_colmap_amd.vit.hip_ops import add_layernorm

    g = torch.Generator(device="cuda").manual_seed(rows + C)
    x = (torch.randn(rows, C, device="cuda", generator=g) * 2).to(torch.bfloat16)
    r = (torch.randn(rows, C, device="cuda", generator=g) * 3 + 0.5).to(torch.bfloat16)
    w = (1 + 0.2 * torch.randn(C, device="cuda", generator=g)).to(torch.bfloat16)
    b = (0.1 * torch.randn(C, device="cuda", generator=g)).to(torch.bfloat16)
    s, y = add_layernorm(x, r, w, b, 1e-6)
    ref_s = x + r                                            # bf16 add, as the unfused model does
    assert torch.equal(s, ref_s)
    ref_y = torch.nn.functional.layer_norm(ref_s, (C,), w, b, 1e-6)
    # both round a float32 result to bf16: allow one bf16 ulp (statistics are summed in a different order)
    err = (y.float() - ref_y.float()).abs()
    tol = ref_y.float().abs() * 2 ** -7 + 1e-3
    assert bool((err <= tol).all()), float(err.max())
    assert (y != ref_y).float().mean().item() < 0.05
    _, y0 = add_layernorm(x, None, w, b, 1e-6)
    ref0 = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-6)
    assert bool(((y0.float() - ref0.float()).abs() <= ref0.float().abs() * 2 ** -7 + 1e-3).all())
    s2, y2 = add_layernorm(x, r, w, b, 1e-6, want_sum=False)
    assert s2 is None and torch.equal(y2, y)


def test_fused_block_path_equals_unfused_modules():
    """Same weights, same input: fused add+LayerNorm path vs the plain nn.Module path (both bf16)."""
    from vit_colmap_amd.vit import build_dinov2

    m = build_dinov2("dinov2_vits14").init_random(seed=2).eval().fold_layerscale().to("cuda", torch.bfloat16)
    g = torch.Generator(device="cuda").manual_seed(0)
    patches = torch.randn(2, 34 * 45, 588, device="cuda", generator=g).to(torch.bfloat16)
    with torch.inference_mode():
        fused = m.forward_patch_tokens(patches, 34, 45).float()
        for b in m.blocks:
            b.folded = False          # gammas are 1 after folding... force the module path
        for b in m.blocks:
            b.ls1.gamma.fill_(1.0); b.ls2.gamma.fill_(1.0)
        plain = m.forward_patch_tokens(patches, 34, 45).float()
    rel = ((fused - plain).norm() / plain.norm()).item()
    assert rel < 1e-2, rel           # two bf16 evaluation orders of the same network


@pytest.mark.parametrize("B,N,H", [(2, 1531, 6), (1, 64, 1), (3, 257, 12), (1, 100, 2), (2, 128, 6), (1, 1, 6)])
def test_attention_matches_float32_reference(B, N, H):
    """Hand-written flash attention (through the C ABI) vs softmax(QK^T/8)V in float32."""
    from vit_colmap_amd.vit.hip_ops import attention

    g = torch.Generator(device="cuda").manual_seed(B * 1000 + N + H)
    qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).to(torch.bfloat16)
    out = attention(qkv, H).float()
    q, k, v = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B, N, H * 64)
    err = (out - ref).abs().max().item()
    rel = ((out - ref).norm() / ref.norm()).item()
    assert rel < 1e-2 and err < 5e-2, (err, rel)          # bf16 P and bf16 output
    # and against PyTorch's own bf16 kernel: the two bf16 implementations agree to the same level
    tq, tk, tv = qkv.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    tref = torch.nn.functional.scaled_dot_product_attention(tq, tk, tv).transpose(1, 2).reshape(B, N, H * 64).float()
    assert ((out - tref).norm() / tref.norm()).item() < 1e-2


def test_attention_structured_values_catch_layout_errors():
    """Asymmetric, position-coded K and V: a swapped row/column or a wrong key permutation in the
    P^T -> MFMA operand path cannot cancel out."""
    from vit_colmap_amd.vit.hip_ops import attention

    B, N, H = 1, 200, 2
    qkv = torch.zeros(B, N, 3, H, 64, device="cuda")
    n = torch.arange(N, device="cuda", dtype=torch.float32)
    d = torch.arange(64, device="cuda", dtype=torch.float32)
    qkv[0, :, 0, 0] = (torch.sin(0.37 * n)[:, None] * torch.cos(0.11 * d)[None]) * 2
    qkv[0, :, 1, 0] = (torch.cos(0.23 * n)[:, None] * torch.sin(0.19 * d + 1)[None]) * 2
    qkv[0, :, 2, 0] = (n[:, None] / N) + 0.01 * d[None] * torch.sign(torch.sin(n))[:, None]
    qkv[0, :, :, 1] = torch.randn(N, 3, 64, device="cuda")
    qb = qkv.reshape(B, N, -1).to(torch.bfloat16)
    out = attention(qb, H).float()
    q, k, v = qb.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).transpose(1, 2).reshape(B, N, H * 64)
    assert (out - ref).abs().max().item() < 2e-2
