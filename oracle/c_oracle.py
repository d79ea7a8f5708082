"""ctypes loader for oracle/libvco_oracle.so (TEST INFRASTRUCTURE — see oracle/__init__.py)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvco_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = ctypes.CDLL(path)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.vco_match_pair_u8.argtypes = [u8p, ctypes.c_int, u8p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_float, ctypes.c_float, ctypes.c_int, u32p]
        L.vco_match_pair_u8.restype = ctypes.c_int
        L.vco_match_pairs_u8.argtypes = [u8p, i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p,
                                         ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                         u32p, i32p, ctypes.c_int]
        L.vco_match_pairs_u8.restype = ctypes.c_int
        L.vco_top2_both.argtypes = [u8p, ctypes.c_int, u8p, ctypes.c_int, ctypes.c_int] + [i32p] * 6
        L.vco_top2_both.restype = None
        L.vco_theta.argtypes = [ctypes.c_int32]
        L.vco_theta.restype = ctypes.c_float
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def match_pair(d1, d2, max_ratio=0.8, max_distance=0.7, cross_check=True):
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    out = np.zeros((max(len(d1), 1), 2), np.uint32)
    d = d1.shape[1] if d1.ndim == 2 else 0
    m = lib().vco_match_pair_u8(_p(d1, ctypes.c_uint8), len(d1), _p(d2, ctypes.c_uint8), len(d2), d,
                                max_ratio, max_distance, int(cross_check), _p(out, ctypes.c_uint32))
    return out[:m].copy()


def top2_both(d1, d2):
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    n1, n2 = len(d1), len(d2)
    outs = [np.zeros(n, np.int32) for n in (n1, n1, n1, n2, n2, n2)]
    lib().vco_top2_both(_p(d1, ctypes.c_uint8), n1, _p(d2, ctypes.c_uint8), n2, d1.shape[1],
                        *[_p(o, ctypes.c_int32) for o in outs])
    return outs


def match_pairs(desc, counts, pairs, max_ratio=0.8, max_distance=0.7, cross_check=True, num_threads=0):
    """desc uint8 [n_images][n_max][D]; returns (matches [P][n_max][2] uint32, counts [P], threads)."""
    desc = np.ascontiguousarray(desc, np.uint8)
    counts = np.ascontiguousarray(counts, np.int32)
    pairs = np.ascontiguousarray(pairs, np.int32)
    n_images, n_max, d = desc.shape
    P = len(pairs)
    out = np.zeros((P, n_max, 2), np.uint32)
    cnt = np.zeros(P, np.int32)
    used = lib().vco_match_pairs_u8(_p(desc, ctypes.c_uint8), _p(counts, ctypes.c_int32), n_images, n_max, d,
                                    _p(pairs, ctypes.c_int32), P, max_ratio, max_distance,
                                    int(cross_check), _p(out, ctypes.c_uint32), _p(cnt, ctypes.c_int32),
                                    num_threads)
    return out, cnt, used


def theta(s):
    return lib().vco_theta(int(s))
