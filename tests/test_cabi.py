"""The C-ABI library loads and exports every symbol include/rslo_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

from rslo_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "rslo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"RSLO_API[^;(]*?\b(rslo_\w+)\s*\(", txt)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert len(syms) >= 24 and "rslo_spconv_fwd" in syms and "rslo_chamfer_nn" in syms


def test_library_builds_and_exports_all_symbols():
    so = build.build(verbose=False)
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    for s in declared_symbols():
        assert hasattr(lib, s), "missing export " + s
    assert lib.rslo_abi_version() == 1


def test_binding_covers_header():
    assert sorted(capi.SIGNATURES) == declared_symbols()
    capi.lib()  # sets restype/argtypes on every symbol


def test_host_only_helpers():
    l = capi.lib()
    assert l.rslo_hash_capacity(1000) == 2048
    assert l.rslo_hash_capacity(0) == 1024
    assert l.rslo_conv_bitmap_words(2, (ctypes.c_int32 * 3)(21, 384, 704)) == (2 * 21 * 384 * 704 + 31) // 32
    assert l.rslo_spconv_wgrad_ws_bytes(5000, 27, 64, 64) == 3 * (27 * 64 * 64 + 64) * 4


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    with pytest.raises(capi.RsloHipError):
        capi.chamfer_nn(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))
