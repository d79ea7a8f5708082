"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares,
and the Python signature table covers exactly those symbols.  No kernels are launched."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vitcolmap_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("vc_prepare_descriptors", "vc_match_pairs_u8", "vc_knn_top2_u8", "vc_mutual_ratio"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from vit_colmap_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_pure_host_entry_points():
    from vit_colmap_amd import _lib

    lib = _lib.load()
    assert lib.vc_abi_version() == 1
    assert lib.vc_status_string(0) == b"ok"
    assert b"invalid" in lib.vc_status_string(-1)
    # 50 images x 512 x 384: 16 tiles x 12 fragments x 1 KiB + 512 row sums + 512 head row sums + 16 tail norm bounds
    assert lib.vc_prepared_bytes(50, 512, 384) == 50 * (16 * 12 * 1024 + 2 * 512 * 4 + 16 * 4)
    assert lib.vc_prepared_bytes(1, 300, 128) == 16 * 4 * 1024 + 2 * 512 * 4 + 16 * 4   # tiles round up to 16
    assert lib.vc_prepared_bytes(1, 512, 4096) == 0       # beyond VC_MAX_DESC_DIM
    assert lib.vc_knn_workspace_bytes(512, 512, 384) > 2 * 196608


def test_argument_validation_needs_no_gpu():
    from vit_colmap_amd import _lib

    lib = _lib.load()
    assert lib.vc_prepare_descriptors(None, None, 1, 512, 384, None, None) == -1
    assert lib.vc_match_pairs_u8(None, None, 1, 512, 384, None, 1, 0.8, 0.7, 1, None, None, None) == -1
    assert lib.vc_theta_table(None, 4, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vit_colmap_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryError):
        _lib.load()


def test_product_package_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vit_colmap_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "libvco_oracle" in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, f"product code references the oracle: {bad}"
