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


def test_host_library_exports_its_header():
    """librslo_host.so (include/rslo_host.h): built by gcc, no HIP dependency, every declared symbol exported and bound."""
    import subprocess
    from rslo_amd import hostlib
    so = build.build_host(verbose=False)
    txt = open(os.path.join(ROOT, "include", "rslo_host.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    syms = sorted(set(re.findall(r"RSLO_HOST_API[^;(]*?\b(rslo_host_\w+)\s*\(", txt)))
    assert syms == sorted(hostlib.SIGNATURES) and "rslo_host_voxelize" in syms
    lib = ctypes.CDLL(so)
    for s in syms:
        assert hasattr(lib, s)
    assert hostlib.lib().rslo_host_abi_version() == 1
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "amdhip" not in needed and "torch" not in needed


def test_plan_encoder_layout_is_consistent():
    """rslo_plan_encoder_layout is host-only: level shapes follow the conv arithmetic, capacities follow the documented
    defaults, regions do not overlap (also pins the ctypes mirror of the two structs to the header's layout)."""
    import rslo_amd  # noqa: F401
    from rslo_amd import plan, workload
    net, _ = workload.build_network(device="cpu")
    pl = plan.EncoderPlanner(net, 40000)
    spec = pl._spec(7, True)
    lay = capi.plan_encoder_layout(spec, [120000] * 7 + [1000])
    assert [list(d) for d in lay.dims][:5] == [[41, 768, 1408], [21, 384, 704], [11, 192, 352], [5, 96, 176], [2, 96, 176]]
    assert list(lay.cap_rows)[:5] == [8 * 40000] * 4 + [8 * 2 * 96 * 176]      # never more rows than the level has cells
    assert list(lay.hash_cap)[:5] == [1 << 20] * 5
    offs = {"counts": lay.counts_off, "voxels": lay.voxels_off, "num": lay.num_points_off, "cf": lay.coords_frame_off,
            "vox_ws": lay.vox_ws_off, "bitmap": lay.bitmap_off, "prefix": lay.prefix_off, "scan": lay.scan_ws_off,
            "pair_ws": lay.pair_ws_off}
    for l in range(5):
        for name in ("coords_off", "keys_off", "vals_off"):
            offs["%s%d" % (name, l)] = getattr(lay, name)[l]
    for l in range(4):
        for name in ("subm_nbr_off", "subm_pin_off", "subm_pout_off", "subm_koff_off", "conv_nbr_off", "conv_nbrT_off",
                     "conv_order_off", "conv_pin_off", "conv_pout_off", "conv_koff_off"):
            offs["%s%d" % (name, l)] = getattr(lay, name)[l]
    vals = sorted(offs.values())
    assert len(set(vals)) == len(vals) and all(v % 256 == 0 for v in vals) and vals[-1] < lay.total_bytes
    assert lay.voxels_off + 320000 * 10 * 7 * 4 <= lay.num_points_off
    assert lay.scratch_words == (8 * 21 * 384 * 704 + 31) // 32
    small = capi.plan_encoder_layout(spec, [100] * 8)            # fewer points than max_voxels: capacity = sum P
    assert small.cap_rows[0] == 800


def test_every_environment_switch_is_registered():
    """rslo_amd/switches.py is the complete list of environment variables the Python host reads; the C sources read none
    (rslo_tuning_set)."""
    import glob
    import re
    from rslo_amd import switches
    found = set()
    for f in glob.glob(os.path.join(ROOT, "rslo_amd", "**", "*.py"), recursive=True):
        if f.endswith("switches.py"):
            continue
        found |= set(re.findall(r"RSLO_[A-Z0-9_]+", open(f).read()))
    found = {v for v in found if not v.startswith("RSLO_TUNE")}
    assert found <= set(switches.SWITCHES), sorted(found - set(switches.SWITCHES))
    assert set(switches.SWITCHES) <= found | {"RSLO_HEAD_NHWC"}, sorted(set(switches.SWITCHES) - found)
    for f in glob.glob(os.path.join(ROOT, "rslo_amd", "csrc", "**", "*"), recursive=True):
        if os.path.isfile(f) and f.endswith((".hip", ".h", ".c")):
            assert "getenv" not in open(f).read(), f
