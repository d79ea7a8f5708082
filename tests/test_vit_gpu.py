"""GPU tests of the fused ViT glue kernels against PyTorch's own ops on the same device."""
import numpy as np
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


@pytest.mark.parametrize("rows,K,N", [(1531 * 2, 384, 1152), (300, 384, 384), (1531, 384, 1536), (1000, 1536, 384),
                                      (1, 64, 128), (129, 768, 2304), (128, 128, 256),
                                      (1531 * 3, 1536, 384), (4096, 64, 128), (4097, 640, 384),
                                      # the 256 x 256 tile kernel (n_out % 256 == 0, >= 1024 rows): ragged last row tile, one and many K steps
                                      (1531 * 2, 768, 2304), (1024, 64, 256), (1025, 3072, 768), (2047, 768, 768), (5000, 1024, 4096)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_linear_matches_float32_reference(rows, K, N, epi):
    """Hand-written bf16 GEMM + fused epilogue (through the C ABI) vs the float32 evaluation of the same bf16 data."""
    from vit_colmap_amd.vit.hip_ops import linear

    g = torch.Generator(device="cuda").manual_seed(rows + K + N + epi)
    x = torch.randn(rows, K, device="cuda", generator=g).to(torch.bfloat16)
    # asymmetric, non-uniform weights: a swapped fragment map or a transposed tile cannot pass
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5 * torch.linspace(0.5, 2.0, N, device="cuda")[:, None]).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    r = torch.randn(rows, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    out = linear(x, w, b, epi, r).float()
    ref = x.float() @ w.float().t() + b.float()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + r.float()
    # one rounding of a float32 result to bf16 (2^-9 relative) + float32 accumulation-order noise
    err = (out - ref).abs()
    tol = ref.abs() * 2 ** -8 + 2e-3
    assert bool((err <= tol).all()), (float(err.max()), int((err > tol).sum()))


def test_linear_rejects_unsupported_shapes():
    from vit_colmap_amd import _lib
    from vit_colmap_amd.vit.hip_ops import linear

    x = torch.zeros(4, 100, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(128, 100, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(_lib.HipLibraryError):
        linear(x, w, torch.zeros(128, device="cuda", dtype=torch.bfloat16))


@pytest.mark.parametrize("rows,N", [(1531 * 2, 1152), (300, 384), (1531 * 3, 1536), (1, 32), (255, 64), (257, 96),
                                    (256 * 300, 384), (1531 * 50, 1152), (1531 * 50 + 37, 384)])
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("ln", [False, True])
def test_xs_linear_matches_float32_reference(rows, N, epi, ln):
    """K=384 GEMM (+ fused LayerNorm / epilogue) through the C ABI vs float32 on the same bf16 data: the x-stationary kernel
    and, for the long launches without LayerNorm (the last three shapes), the weight-stationary one behind the same entry."""
    from vit_colmap_amd.vit.hip_ops import XsLinear

    K = 384
    g = torch.Generator(device="cuda").manual_seed(rows + N + epi + 7 * ln)
    x = (torch.randn(rows, K, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5 * torch.linspace(0.5, 2.0, N, device="cuda")[:, None]
    b = torch.randn(N, device="cuda", generator=g)
    gam = 1 + 0.2 * torch.randn(K, device="cuda", generator=g) if ln else None
    bet = 0.1 * torch.randn(K, device="cuda", generator=g) if ln else None
    r = torch.randn(rows, N, device="cuda", generator=g).to(torch.bfloat16) if epi == 2 else None
    lin = XsLinear(w, b, gam, bet, 1e-6)
    out = lin(x, epi, r).float()
    xin = x.float()
    if ln:
        xin = torch.nn.functional.layer_norm(xin, (K,), gam, bet, 1e-6)
    ref = xin @ w.t() + b
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + r.float()
    # operands are rounded to bf16 (x-hat and W diag(gamma): 2^-9 each, accumulated over K = 384 random-sign
    # terms), the result once more
    err = (out - ref).abs()
    # (operand noise ~5e-3 rms at the widest rows, 5-sigma tails over millions of outputs, + half a bf16 ulp)
    tol = ref.abs() * 2 ** -7 + 6e-2
    assert bool((err <= tol).all()), (float(err.max()), int((err > tol).sum()))
    assert float((err > ref.abs() * 2 ** -7 + 3e-2).float().mean()) < 1e-5
    assert float((out - ref).norm() / ref.norm()) < 6e-3
    if epi == 2:   # in place on the residual stream
        r2 = r.clone()
        lin(x, epi, r2, out=r2)
        assert torch.equal(r2.float(), out)


@pytest.mark.parametrize("B,H,W,C,N,kh,kw,dy0,dx0,epi", [
    (2, 34, 45, 384, 256, 3, 3, -1, -1, 0),      # 3 x 3 pad 1 on the token grid
    (1, 68, 90, 512, 512, 3, 3, -1, -1, 1),      # the upsampler's conv + GELU
    (3, 17, 23, 128, 256, 2, 2, -1, -1, 0),      # one parity class of the transposed convolution (taps up / left)
    (3, 17, 23, 128, 256, 2, 2, 0, 0, 1),        # ... (taps down / right)
    (1, 5, 7, 64, 256, 1, 1, 0, 0, 0),           # 1 x 1: a plain GEMM, fewer rows than one tile
    (1, 1, 3, 64, 256, 3, 3, -1, -1, 0),         # a one-row image: six of nine taps outside everywhere
    (2, 2, 2, 128, 512, 2, 2, 0, 0, 1),          # images smaller than the tap window
    (5, 9, 31, 192, 256, 3, 3, -1, -1, 1),       # 1395 rows: tiles straddle image boundaries, last tile ragged
    (2, 120, 160, 256, 256, 3, 3, -1, -1, 1)])   # the heads' resolution
def test_conv_taps_matches_float32_convolution(B, H, W, C, N, kh, kw, dy0, dx0, epi):
    """vc_conv_taps_bf16 (implicit GEMM over a channels-last batch, zero outside the image) vs torch's float32 conv2d on the
    same bf16 data; the spare row behind the batch is filled with junk first (the call must zero it)."""
    from vit_colmap_amd.vit.hip_ops import conv_rows, conv_taps

    g = torch.Generator(device="cuda").manual_seed(B * 1000 + H + C + kh + epi)
    xr = conv_rows(B, H, W, C, "cuda")
    xr.copy_((torch.randn(xr.shape, device="cuda", generator=g) * 1.2 + 0.2).to(torch.bfloat16))   # spare row: junk
    k = kh * kw * C
    w = (torch.randn(N, kh, kw, C, device="cuda", generator=g) / k ** 0.5 * torch.linspace(0.5, 2, N, device="cuda")[:, None, None, None]).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    out = conv_taps(xr, w.reshape(N, k).contiguous(), b, B, H, W, kh, kw, dy0, dx0, epi).float()
    assert not bool(xr[B * H * W].any())
    img = xr[: B * H * W].float().reshape(B, H, W, C).permute(0, 3, 1, 2)
    # out(y, x) = sum_t in(y + dy0 + ty, x + dx0 + tx) w[t]  ==  cross-correlation with padding (top = -dy0, left = -dx0, ...)
    pad = (-dx0, kw - 1 + dx0, -dy0, kh - 1 + dy0)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(img, pad), w.float().permute(0, 3, 1, 2), b.float())
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, N)
    err = (out - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -7 + 2e-2).all()), (float(err.max()), int((err > ref.abs() * 2 ** -7 + 2e-2).sum()))
    assert float((out - ref).norm() / ref.norm()) < 4e-3


def test_conv_taps_random_geometries():
    """Seeded random geometries (image sizes down to one pixel, windows up to 3 x 4 anywhere within +-2 of the pixel, 64..320
    channels, batches that end inside a tile) against a float32 restatement of the entry's definition."""
    from vit_colmap_amd.vit.hip_ops import conv_rows, conv_taps

    rng = np.random.RandomState(20240)
    for case in range(14):
        B, H, W = int(rng.randint(1, 5)), int(rng.randint(1, 24)), int(rng.randint(1, 24))
        C, N = 64 * int(rng.randint(1, 6)), 256 * int(rng.randint(1, 3))
        kh, kw = int(rng.randint(1, 4)), int(rng.randint(1, 5))
        dy0, dx0 = int(rng.randint(-2, 1)), int(rng.randint(-2, 1))
        epi = int(rng.randint(0, 2))
        g = torch.Generator(device="cuda").manual_seed(1000 + case)
        xr = conv_rows(B, H, W, C, "cuda")
        xr.copy_(torch.randn(xr.shape, device="cuda", generator=g).to(torch.bfloat16))
        k = kh * kw * C
        w = (torch.randn(N, k, device="cuda", generator=g) / k ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
        out = conv_taps(xr, w, b, B, H, W, kh, kw, dy0, dx0, epi).float()
        img = xr[: B * H * W].float().reshape(B, H, W, C)
        ref = b.float().reshape(1, 1, 1, N).expand(B, H, W, N).clone()
        for ty in range(kh):
            for tx in range(kw):
                dy, dx = dy0 + ty, dx0 + tx
                ys, ye, xs, xe = max(0, -dy), min(H, H - dy), max(0, -dx), min(W, W - dx)
                if ys >= ye or xs >= xe:
                    continue
                t = ty * kw + tx
                ref[:, ys:ye, xs:xe] += img[:, ys + dy:ye + dy, xs + dx:xe + dx] @ w[:, t * C:(t + 1) * C].float().t()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        ref = ref.reshape(B * H * W, N)
        err = (out - ref).abs()
        assert bool((err <= ref.abs() * 2 ** -7 + 2e-2).all()), (case, (B, H, W, C, N, kh, kw, dy0, dx0, epi), float(err.max()))


def test_conv_taps_parity_classes_are_a_transposed_convolution():
    """Four vc_conv_taps_bf16 calls with out_parity = 2 i + j and the class matrices of model/hip_heads.py fill the
    [B][2H][2W][N] tensor that ConvTranspose2d(kernel 4, stride 2, padding 1) computes (float32 on the same bf16 data)."""
    from vit_colmap_amd.model.hip_heads import deconv_class_matrices
    from vit_colmap_amd.vit.hip_ops import conv_rows, conv_taps, EPI_BIAS

    B, H, W, C, N = 3, 9, 13, 128, 256
    g = torch.Generator(device="cuda").manual_seed(11)
    xr = conv_rows(B, H, W, C, "cuda")
    xr.copy_(torch.randn(xr.shape, device="cuda", generator=g).to(torch.bfloat16))
    wt = (torch.randn(C, N, 4, 4, device="cuda", generator=g) / (4 * C) ** 0.5).to(torch.bfloat16)       # ConvTranspose2d layout
    b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.full((B * 4 * H * W + 1, N), 7.0, dtype=torch.bfloat16, device="cuda")
    for m, i, j, dy0, dx0 in deconv_class_matrices(wt):
        conv_taps(xr, m.contiguous(), b, B, H, W, 2, 2, dy0, dx0, EPI_BIAS, out=out, out_parity=2 * i + j)
    assert bool((out[-1] == 7.0).all())                                                                 # nothing written past the tensor
    img = xr[: B * H * W].float().reshape(B, H, W, C).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv_transpose2d(img, wt.float(), b.float(), stride=2, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * 4 * H * W, N)
    err = (out[:-1].float() - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -7 + 2e-2).all()), float(err.max())


def test_patch_embed_gemm_matches_reference_and_padded_preprocess():
    """vc_patch_embed_bf16 (patch embedding + bias + position embedding, class-token row skipped) vs float32 on the
    same bf16 data, and the padded patch layout of the preprocessing kernel vs the unpadded one."""
    from vit_colmap_amd.features import hip_preprocess
    from vit_colmap_amd.vit.hip_ops import patch_embed

    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.randint(0, 256, (3, 70, 98, 3), device="cuda", generator=g, dtype=torch.uint8)
    p0 = hip_preprocess.preprocess(img, layout="patches")
    p1 = hip_preprocess.preprocess(img, layout="patches_pad")
    assert p1.shape == (3, 35, 640) and torch.equal(p1[..., :588], p0) and not bool(p1[..., 588:].any())
    B, T, C = 3, 35, 384
    w = torch.zeros(C, 640, device="cuda")
    w[:, :588] = torch.randn(C, 588, device="cuda", generator=g) / 588 ** 0.5 * torch.linspace(0.5, 2, C, device="cuda")[:, None]
    w = w.to(torch.bfloat16)
    b = torch.randn(C, device="cuda", generator=g).to(torch.bfloat16)
    pos = torch.randn(1, T + 1, C, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.full((B, T + 1, C), 7.0, device="cuda", dtype=torch.bfloat16)
    patch_embed(p1, w, b, pos, out)
    assert bool((out[:, 0] == 7.0).all())                       # class-token rows untouched
    ref = p1.float() @ w.float().t() + b.float() + pos[0, 1:].float()
    err = (out[:, 1:].float() - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -8 + 1e-2).all()), float(err.max())


def test_xs_gelu_table_is_the_bf16_gelu_for_every_input():
    """With a table, the fc1 epilogue evaluates gelu on the bf16-rounded pre-activation and rounds to bf16.  Identity GEMM:
    (A) every bf16 value the table covers (2^-14 <= |x| < 8) must come out as torch's bf16 gelu (float32 erf, RNE);
    (B) every finite bf16 value at all (blocks holding an out-of-range value take the float path) within float tolerance."""
    from vit_colmap_amd.vit.hip_ops import XsLinear, gelu_table, EPI_GELU

    K = N = 384
    lin = XsLinear(torch.eye(K, device="cuda"), torch.zeros(N, device="cuda"))
    tab = gelu_table("cuda")
    bits = torch.arange(0, 65536, dtype=torch.int32, device="cuda")
    allv = bits.to(torch.int16).view(torch.bfloat16)
    mag = allv.float().abs()

    def run(vals, table):
        rows = (vals.numel() + K - 1) // K
        x = vals.repeat((rows * K + vals.numel() - 1) // vals.numel())[: rows * K].reshape(rows, K).contiguous()
        out = lin(x, EPI_GELU, gelu_table=table)
        return x.reshape(-1), out.reshape(-1)

    # (A) table range only: bit pattern equality up to torch-vs-device erff last-bit differences
    inr = allv[(mag >= 2.0 ** -14) & (mag < 8.0)]
    assert inr.numel() == 2 * 17 * 128
    x, out = run(inr, tab)
    ref = torch.nn.functional.gelu(x.float()).to(torch.bfloat16)
    d = (out.view(torch.int16).int() - ref.view(torch.int16).int()).abs()
    assert int(d.max()) <= 1, int(d.max())
    assert float((d != 0).float().mean()) < 2e-3, float((d != 0).float().mean())
    # (B) all finite values, with and without the table: float32 accuracy (the float path's erf is good to 1.5e-7 absolute)
    fin = allv[torch.isfinite(allv.float())]
    for table in (tab, None):
        x, out = run(fin, table)
        ref32 = torch.nn.functional.gelu(x.float())
        err = (out.float() - ref32).abs()
        assert bool((err <= ref32.abs() * 2 ** -8 + 2e-7 * (1 + x.float().abs())).all()), float(err.max())


@pytest.mark.parametrize("B,N,H", [(2, 1531, 6), (1, 64, 1), (3, 257, 12), (1, 100, 2), (1, 1, 6)])
@pytest.mark.parametrize("spread", [1.0, 6.0])
def test_attention_prescaled_q_lazy_max(B, N, H, spread):
    """q_prescaled mode (q carries (1/8) log2 e; running maximum subtracted inside the product, deferred max) vs float32
    softmax on the same bf16 data.  spread 6: score ranges wide enough that later blocks exceed the running maximum by
    more than the lazy threshold, so the exact-rescale path runs too."""
    import math
    from vit_colmap_amd.vit.hip_ops import attention

    g = torch.Generator(device="cuda").manual_seed(B * 100 + N + H)
    qkv = torch.randn(B, N, 3, H, 64, device="cuda", generator=g)
    qkv[:, :, 0] *= spread * 0.125 * math.log2(math.e)
    # make late keys larger so that the running maximum keeps moving
    qkv[:, :, 1] *= torch.linspace(0.5, 1.5, N, device="cuda")[None, :, None, None]
    qkv = qkv.to(torch.bfloat16).reshape(B, N, 3 * H * 64).contiguous()
    out = attention(qkv, H, q_prescaled=True).float()
    q, k, v = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    att = torch.softmax(q @ k.transpose(-1, -2) * math.log(2.0), dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B, N, H * 64)
    err = (out - ref).abs()
    assert float(err.max()) < 3e-2 and float((out - ref).norm() / ref.norm()) < 1e-2, (float(err.max()), float((out - ref).norm() / ref.norm()))


@pytest.mark.parametrize("rows", [1, 127, 128, 300, 1531 * 3, 128 * 256 + 128 * 3 + 5, 128 * 256 + 1, 1531 * 50])
def test_fused_mlp_matches_float32_reference(rows):
    """x += fc2(gelu(fc1(LN(x)))) in one kernel vs the float32 evaluation (GELU on the bf16-rounded pre-activation, as the
    bf16 pipeline and the kernel's table define it)."""
    from vit_colmap_amd.vit.hip_ops import FusedMlp

    K, Hd = 384, 1536
    g = torch.Generator(device="cuda").manual_seed(rows)
    x = (torch.randn(rows, K, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    w1 = torch.randn(Hd, K, device="cuda", generator=g) / K ** 0.5 * torch.linspace(0.5, 2.0, Hd, device="cuda")[:, None]
    b1 = 0.5 * torch.randn(Hd, device="cuda", generator=g)
    w2 = torch.randn(K, Hd, device="cuda", generator=g) / Hd ** 0.5 * torch.linspace(0.5, 1.5, K, device="cuda")[:, None]
    b2 = torch.randn(K, device="cuda", generator=g)
    gam = 1 + 0.2 * torch.randn(K, device="cuda", generator=g)
    bet = 0.1 * torch.randn(K, device="cuda", generator=g)
    mlp = FusedMlp(w1, b1, gam, bet, w2, b2, 1e-6)
    xin = x.float()
    h = torch.nn.functional.layer_norm(xin, (K,), gam, bet, 1e-6) @ w1.t() + b1
    hg = torch.nn.functional.gelu(h.to(torch.bfloat16).float()).to(torch.bfloat16).float()
    ref = xin + hg @ w2.t() + b2
    out = mlp(x.clone()).float()
    err = (out - ref).abs()
    assert float((out - ref).norm() / ref.norm()) < 8e-3, float((out - ref).norm() / ref.norm())
    assert bool((err <= ref.abs() * 2 ** -6 + 0.12).all()), (float(err.max()), int((err > ref.abs() * 2 ** -6 + 0.12).sum()))


def test_layernorm_drop_first_equals_layernorm_of_the_remaining_rows():
    """vc_layernorm_drop_first_bf16: the final norm of the ViT-S path skips the class-token row of every image and writes the
    patch tokens densely — bit-identical to vc_add_layernorm_bf16 on all rows followed by the slice."""
    from vit_colmap_amd.vit.hip_ops import add_layernorm, layernorm_drop_first

    g = torch.Generator(device="cuda").manual_seed(3)
    for (B, N, C) in ((3, 1531, 384), (1, 2, 384), (5, 17, 768)):
        x = (torch.randn(B, N, C, device="cuda", generator=g) * 2 + 0.5).to(torch.bfloat16)
        w = (1 + 0.1 * torch.randn(C, device="cuda", generator=g)).to(torch.bfloat16)
        b = (0.1 * torch.randn(C, device="cuda", generator=g)).to(torch.bfloat16)
        _, full = add_layernorm(x, None, w, b, 1e-6)
        y = layernorm_drop_first(x, w, b, 1e-6)
        assert tuple(y.shape) == (B, N - 1, C)
        assert torch.equal(y.view(torch.int16), full[:, 1:].contiguous().view(torch.int16))
