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
