"""Torch front end of the ViT kernels (csrc/vit_ops.hip, attention.hip, gemm.hip): bf16, through the C ABI."""
import torch

from .. import _lib


def add_layernorm(x: torch.Tensor, residual, weight: torch.Tensor, bias: torch.Tensor, eps: float,
                  want_sum: bool = True):
    """y = LayerNorm(x + residual); returns (x + residual in bf16 or None, y).  residual=None: plain LN."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()
    lib = _lib.load()
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    s = torch.empty_like(x) if (residual is not None and want_sum) else None
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous()
    _lib.check(lib.vc_add_layernorm_bf16(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(weight), _lib.ptr(bias), eps, rows, C,
                                         _lib.ptr(s), _lib.ptr(y), _lib.stream_ptr()), "vc_add_layernorm_bf16")
    return s, y


def layernorm_drop_first(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, out=None) -> torch.Tensor:
    """x (B, N, C) bf16 -> LayerNorm of rows 1.. of every image as a dense (B, N - 1, C) tensor (the class-token row is
    neither normalised nor copied).  `out`: a contiguous (B, N - 1, C) bf16 destination (e.g. a batch slice)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 3
    B, N, C = x.shape
    y = torch.empty((B, N - 1, C), dtype=torch.bfloat16, device=x.device) if out is None else out
    assert y.shape == (B, N - 1, C) and y.dtype == torch.bfloat16 and y.is_contiguous()
    _lib.check(_lib.load().vc_layernorm_drop_first_bf16(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), eps, B, N, C, _lib.ptr(y),
                                                        _lib.stream_ptr()), "vc_layernorm_drop_first_bf16")
    return y


Q_PRESCALE = 0.125 * 1.4426950408889634   # 1/sqrt(64) * log2(e): what `q_prescaled=True` expects folded into q


def attention(qkv: torch.Tensor, n_heads: int, q_prescaled: bool = False) -> torch.Tensor:
    """qkv (B, N, 3*H*64) bf16 contiguous -> softmax(QK^T/8)V as (B, N, H*64) bf16 (csrc/attention.hip).
    q_prescaled: the q part already carries Q_PRESCALE (folded into the qkv projection)."""
    assert qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    B, N, C3 = qkv.shape
    hd = C3 // (3 * n_heads)
    lib = _lib.load()
    out = torch.empty((B, N, C3 // 3), dtype=torch.bfloat16, device=qkv.device)
    _lib.check(lib.vc_attention_bf16(_lib.ptr(qkv), B, N, n_heads, hd, int(q_prescaled), _lib.ptr(out), _lib.stream_ptr()),
               "vc_attention_bf16")
    return out


EPI_BIAS, EPI_GELU, EPI_RESIDUAL = 0, 1, 2


def linear_supported(weight: torch.Tensor) -> bool:
    """Shapes csrc/gemm.hip covers (every Linear of DINOv2 ViT-S/B/L/g does)."""
    n, k = weight.shape
    return n % 128 == 0 and k % 64 == 0


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, epilogue: int = EPI_BIAS,
           residual=None, out=None) -> torch.Tensor:
    """out = epi(x W^T + b) (csrc/gemm.hip): EPI_GELU applies the exact GELU, EPI_RESIDUAL adds `residual`."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()
    assert weight.dtype == torch.bfloat16 and weight.is_contiguous() and bias.dtype == torch.bfloat16
    n, k = weight.shape
    assert x.shape[-1] == k
    rows = x.numel() // k
    if out is None:
        out = torch.empty(x.shape[:-1] + (n,), dtype=torch.bfloat16, device=x.device)
    assert out.is_contiguous() and out.dtype == torch.bfloat16 and out.numel() == rows * n
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == torch.bfloat16 and residual.is_contiguous()
    lib = _lib.load()
    _lib.check(lib.vc_linear_bf16(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(out),
                                  rows, n, k, epilogue, _lib.stream_ptr()), "vc_linear_bf16")
    return out


def conv_rows(batch: int, height: int, width: int, channels: int, device) -> torch.Tensor:
    """A channels-last activation buffer for `conv_taps`: [batch * height * width + 1][channels] bf16 — the image batch plus
    the spare row the kernel keeps at zero for taps outside the image."""
    return torch.empty((batch * height * width + 1, channels), dtype=torch.bfloat16, device=device)


def conv_taps(x_rows: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, batch: int, height: int, width: int,
              kh: int, kw: int, dy0: int, dx0: int, epilogue: int = EPI_BIAS, out=None, out_parity: int = -1) -> torch.Tensor:
    """Convolution over a channels-last image batch on the 256 x 256 implicit-GEMM tile (vc_conv_taps_bf16):
    x_rows [batch * height * width + 1][c_in] (see conv_rows), weight [n_out][kh * kw * c_in] with k = (tap, channel),
    out [batch * height * width (+ anything)][n_out].  A 3 x 3 pad-1 convolution is (kh, kw, dy0, dx0) = (3, 3, -1, -1).
    out_parity = 2 i + j: the rows go to pixels (2y + i, 2x + j) of `out` [batch][2 height][2 width][n_out] (required then)."""
    rows = batch * height * width
    c_in = x_rows.shape[1]
    n = weight.shape[0]
    assert x_rows.is_cuda and x_rows.dtype == torch.bfloat16 and x_rows.is_contiguous() and x_rows.shape[0] >= rows + 1
    assert weight.dtype == torch.bfloat16 and weight.is_contiguous() and weight.shape[1] == kh * kw * c_in
    assert bias.dtype == torch.bfloat16 and bias.numel() == n
    if out is None:
        assert out_parity < 0
        out = torch.empty((rows, n), dtype=torch.bfloat16, device=x_rows.device)
    assert out.is_contiguous() and out.dtype == torch.bfloat16 and out.shape[-1] == n
    assert out.numel() >= (rows if out_parity < 0 else 4 * rows) * n
    lib = _lib.load()
    _lib.check(lib.vc_conv_taps_bf16(_lib.ptr(x_rows), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out), batch, height, width,
                                     c_in, n, kh, kw, dy0, dx0, out_parity, epilogue, _lib.stream_ptr()), "vc_conv_taps_bf16")
    return out


class XsLinear:
    """A Linear with k_in == 384 prepared for csrc/gemm.hip's x-stationary kernel (optionally with the
    preceding LayerNorm folded in).  Built once from float32 parameters; call with bf16 activations."""

    def __init__(self, weight: torch.Tensor, bias, ln_weight=None, ln_bias=None, ln_eps: float = 1e-6):
        lib = _lib.load()
        n, k = weight.shape
        nbytes = lib.vc_linear_xs_weight_bytes(n, k)
        if nbytes == 0:
            raise _lib.HipLibraryError(f"vc_linear_xs: unsupported shape {n}x{k}")
        dev = weight.device
        w = weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        g = None if ln_weight is None else ln_weight.detach().float().contiguous()
        be = None if ln_bias is None else ln_bias.detach().float().contiguous()
        self.n, self.k, self.ln, self.eps = n, k, g is not None, float(ln_eps)
        self.wp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.bias = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(lib.vc_linear_xs_prepare(_lib.ptr(w), _lib.ptr(b), _lib.ptr(g), _lib.ptr(be), n, k, _lib.ptr(self.wp),
                                            _lib.ptr(self.bias), _lib.stream_ptr()), "vc_linear_xs_prepare")

    def __call__(self, x: torch.Tensor, epilogue: int = EPI_BIAS, residual=None, out=None, gelu_table=None) -> torch.Tensor:
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[-1] == self.k
        rows = x.numel() // self.k
        if out is None:
            out = torch.empty(x.shape[:-1] + (self.n,), dtype=torch.bfloat16, device=x.device)
        if residual is not None:
            assert residual.shape == out.shape and residual.dtype == torch.bfloat16 and residual.is_contiguous()
        lib = _lib.load()
        _lib.check(lib.vc_linear_xs_bf16(_lib.ptr(x), _lib.ptr(self.wp), _lib.ptr(self.bias), _lib.ptr(residual),
                                         _lib.ptr(out), rows, self.n, self.k, epilogue, int(self.ln), self.eps,
                                         _lib.ptr(gelu_table), _lib.stream_ptr()), "vc_linear_xs_bf16")
        return out


PATCH_K_PADDED = 640


def patch_embed(patches: torch.Tensor, weight_padded: torch.Tensor, bias: torch.Tensor, pos_embed: torch.Tensor,
                out: torch.Tensor) -> torch.Tensor:
    """out[b, 1 + t] = patches[b, t] W^T + bias + pos_embed[1 + t] (csrc/gemm.hip, EPI_PATCH); row 0 of every
    image (class token) is left to the caller.  patches (B, T, 640) bf16 zero padded, weight (C, 640)."""
    B, T, K = patches.shape
    C = weight_padded.shape[0]
    assert patches.is_cuda and patches.dtype == torch.bfloat16 and patches.is_contiguous() and K == weight_padded.shape[1]
    assert weight_padded.is_contiguous() and bias.is_contiguous() and pos_embed.is_contiguous() and out.is_contiguous()
    assert pos_embed.shape[-2:] == (T + 1, C) and out.shape == (B, T + 1, C) and out.dtype == torch.bfloat16
    lib = _lib.load()
    _lib.check(lib.vc_patch_embed_bf16(_lib.ptr(patches), _lib.ptr(weight_padded), _lib.ptr(bias), _lib.ptr(pos_embed),
                                       _lib.ptr(out), B, T, C, K, _lib.stream_ptr()), "vc_patch_embed_bf16")
    return out


_GELU_TABLES = {}


def gelu_table(device) -> torch.Tensor:
    """The bf16 -> bf16 GELU lookup table of csrc/gemm.hip (one per device, built on first use)."""
    key = torch.device(device)
    if key not in _GELU_TABLES:
        lib = _lib.load()
        t = torch.empty(lib.vc_gelu_table_bytes(), dtype=torch.uint8, device=key)
        _lib.check(lib.vc_gelu_table_bf16(_lib.ptr(t), _lib.stream_ptr()), "vc_gelu_table_bf16")
        _GELU_TABLES[key] = t
    return _GELU_TABLES[key]


class FusedMlp:
    """x += fc2(gelu(fc1(LayerNorm(x)))) in one kernel (csrc/gemm.hip, mlp_kernel), dim 384.  Built once from
    float32 parameters (fc2 with LayerScale already folded in); call with the bf16 residual stream, updated in place."""

    def __init__(self, w1, b1, ln_weight, ln_bias, w2, b2, ln_eps: float = 1e-6):
        lib = _lib.load()
        n_hid, dim = w1.shape
        nbytes = lib.vc_mlp_weight_bytes(n_hid, dim)
        if nbytes == 0 or tuple(w2.shape) != (dim, n_hid):
            raise _lib.HipLibraryError(f"vc_mlp: unsupported shape {n_hid}x{dim}")
        dev = w1.device
        f = lambda t: None if t is None else t.detach().float().contiguous()
        w1f, b1f, g, be, w2f, b2f = f(w1), f(b1), f(ln_weight), f(ln_bias), f(w2), f(b2)
        self.n_hid, self.dim, self.eps = n_hid, dim, float(ln_eps)
        self.wm = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.b1 = torch.empty(n_hid, dtype=torch.float32, device=dev)
        self.b2 = torch.empty(dim, dtype=torch.float32, device=dev)
        self.table = gelu_table(dev)
        _lib.check(lib.vc_mlp_prepare(_lib.ptr(w1f), _lib.ptr(b1f), _lib.ptr(g), _lib.ptr(be), _lib.ptr(w2f), _lib.ptr(b2f),
                                      n_hid, dim, _lib.ptr(self.wm), _lib.ptr(self.b1), _lib.ptr(self.b2), _lib.stream_ptr()),
                   "vc_mlp_prepare")

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[-1] == self.dim
        lib = _lib.load()
        _lib.check(lib.vc_mlp_bf16(_lib.ptr(x), _lib.ptr(self.wm), _lib.ptr(self.b1), _lib.ptr(self.b2), _lib.ptr(self.table),
                                   x.numel() // self.dim, self.n_hid, self.dim, self.eps, _lib.stream_ptr()), "vc_mlp_bf16")
        return x
