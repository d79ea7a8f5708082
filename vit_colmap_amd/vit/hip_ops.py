"""Torch front end of csrc/vit_ops.hip: LayerNorm fused with the preceding residual add (bf16)."""
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


def attention(qkv: torch.Tensor, n_heads: int) -> torch.Tensor:
    """qkv (B, N, 3*H*64) bf16 contiguous -> softmax(QK^T/8)V as (B, N, H*64) bf16 (csrc/attention.hip)."""
    assert qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.is_contiguous()
    B, N, C3 = qkv.shape
    hd = C3 // (3 * n_heads)
    lib = _lib.load()
    out = torch.empty((B, N, C3 // 3), dtype=torch.bfloat16, device=qkv.device)
    _lib.check(lib.vc_attention_bf16(_lib.ptr(qkv), B, N, n_heads, hd, _lib.ptr(out), _lib.stream_ptr()),
               "vc_attention_bf16")
    return out
