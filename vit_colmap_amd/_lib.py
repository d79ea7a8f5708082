"""ctypes binding of libvitcolmap_hip.so (C ABI declared in include/vitcolmap_hip.h).

The library is the product path.  There is no CPU fallback: if the shared object is missing or
a call fails, these functions raise.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITCOLMAP_HIP_LIB: developer override used by tools/ to A/B experimental kernel builds
LIB_PATH = os.environ.get("VITCOLMAP_HIP_LIB") or os.path.join(_HERE, "libvitcolmap_hip.so")

VC_OK = 0
ABI_VERSION = 1
VC_MAX_KEYPOINTS = 2048
VC_MAX_DESC_DIM = 1024


class HipLibraryError(RuntimeError):
    pass


_u8p, _i32p, _u32p, _f32p = POINTER(c_uint8), POINTER(c_int32), POINTER(c_uint32), POINTER(ctypes.c_float)

# name -> (restype, argtypes); the single source the loader and tests/test_abi.py share
SIGNATURES = {
    "vc_abi_version": (c_int, []),
    "vc_status_string": (c_char_p, [c_int]),
    "vc_last_hip_error": (c_int, []),
    "vc_prepared_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vc_prepare_descriptors": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vc_match_pairs_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_float,
                                  c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "vc_knn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vc_knn_top2_u8": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_size_t, c_void_p]),
    "vc_mutual_ratio": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                c_float, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "vc_theta_table": (c_int, [c_void_p, c_int, c_void_p]),
    "vc_theta_eval": (c_int, [c_void_p, c_int, c_void_p]),
    "vc_two_view_score": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "vc_two_view_inliers": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "vc_structure_tensor": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vc_score_map": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vc_select_keypoints": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vc_describe": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                            c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vc_describe_at": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                               c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vc_quantize_u8": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "vc_heatmap_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vc_heatmap_keypoints": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, c_int, c_int,
                                     c_int, c_int, c_int, c_float, c_int, c_float, c_float, c_float, c_float, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "vc_add_layernorm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p,
                                      c_void_p, c_void_p]),
    "vc_layernorm_drop_first_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vc_attention_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vc_linear_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vc_conv_taps_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    "vc_linear_xs_weight_bytes": (c_size_t, [c_int, c_int]),
    "vc_linear_xs_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vc_linear_xs_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_float, c_void_p, c_void_p]),
    "vc_gelu_table_bytes": (c_size_t, []),
    "vc_gelu_table_bf16": (c_int, [c_void_p, c_void_p]),
    "vc_patch_embed_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vc_mlp_weight_bytes": (c_size_t, [c_int, c_int]),
    "vc_mlp_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    "vc_mlp_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "vc_preprocess_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p]),
}

_lib = None


class _DevPtr(c_void_p):
    """Device pointer that remembers which GPU it lives on (see `_guarded`)."""
    device = None


class _AutoStream:
    """Placeholder for "the current stream of the device the tensor arguments live on"."""


_AUTO_STREAM = _AutoStream()


def _guarded(fn, name):
    """Wrap one C-ABI entry point so that it always runs on the GPU its tensors live on.

    The library launches on the stream it is handed and asks HIP for the *current* device (CU counts, per-device
    kernel attributes), so a call with tensors on cuda:1 while cuda:0 is current would dereference cuda:1 memory
    from a cuda:0 kernel.  Every pointer made by `ptr()` carries its tensor's device; the call is refused when
    they disagree, runs under `torch.cuda.device(that device)`, and a `stream_ptr()` placeholder becomes that
    device's current stream."""

    def call(*args):
        devices = {a.device for a in args if isinstance(a, _DevPtr) and a.device is not None}
        if len(devices) > 1:
            raise HipLibraryError(f"{name}: tensor arguments live on different devices: {sorted(map(str, devices))}")
        if not devices:
            if any(a is _AUTO_STREAM for a in args):
                import torch

                args = tuple(c_void_p(torch.cuda.current_stream().cuda_stream) if a is _AUTO_STREAM else a for a in args)
            return fn(*args)
        import torch

        dev = devices.pop()
        with torch.cuda.device(dev):
            args = tuple(c_void_p(torch.cuda.current_stream(dev).cuda_stream) if a is _AUTO_STREAM else a for a in args)
            return fn(*args)

    call.__name__ = name
    return call


class _Library:
    """The loaded shared object: one guarded callable per entry point of SIGNATURES."""

    def __init__(self, cdll):
        self._cdll = cdll
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)  # AttributeError here = ABI drift between header and library
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, _guarded(fn, name) if c_void_p in args else fn)


def load():
    """Load the shared object once and attach signatures.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `make -C vit_colmap_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = _Library(ctypes.CDLL(LIB_PATH))
    if lib.vc_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI version {lib.vc_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def check(status, what):
    if status != VC_OK:
        lib = load()
        msg = lib.vc_status_string(status).decode()
        raise HipLibraryError(f"{what}: {msg} (status {status}, hip error {lib.vc_last_hip_error()})")


def stream_ptr(stream=None):
    """The stream argument of a C-ABI call: an explicit torch stream, or (default) the current stream of the
    device the call's tensors live on, resolved inside the guarded call."""
    if stream is None:
        return _AUTO_STREAM
    return c_void_p(stream.cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None); remembers the tensor's device for the guarded call."""
    if t is None:
        return None
    p = _DevPtr(t.data_ptr())
    p.device = t.device if t.is_cuda else None
    return p
