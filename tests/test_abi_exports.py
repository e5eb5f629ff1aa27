"""The C-ABI library loads without a GPU and exports exactly what include/hgs.h declares."""
import os
import re
import subprocess

from hgs import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(hgs_[a-z0-9_]+)\s*\(", src))


def test_library_loads_and_reports_abi():
    lib = _lib.lib()
    assert lib.hgs_abi_version() == _lib.ABI_VERSION
    assert isinstance(lib.hgs_device_count(), int)           # -1/0 without a GPU, never a crash


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared()
    assert len(declared) >= 20
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b(hgs_[a-z0-9_]+)\b", nm))
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert declared == set(_lib.SIGNATURES), (sorted(declared - set(_lib.SIGNATURES)), sorted(set(_lib.SIGNATURES) - declared))


def test_struct_layouts_match_header():
    import ctypes as C
    # hgs_raster_args: 11 x 4-byte scalars (+ 4 bytes of padding), 14 pointers, 2 x 4-byte scalars, 1 pointer,
    # 2 x 4-byte scalars, 2 pointers, 2 x 4-byte scalars
    assert C.sizeof(_lib.RasterArgs) == 11 * 4 + 4 + 14 * 8 + 2 * 4 + 8 + 2 * 4 + 2 * 8 + 2 * 4
    assert _lib.RasterArgs.lod_render_indices.offset == 184
    assert _lib.RasterArgs.bg.offset == 48
    assert C.sizeof(_lib.RasterGrads) == 9 * 8
    assert C.sizeof(_lib.RasterViews) == 10 * 8
    assert C.sizeof(_lib.HierHost) == 16 + 7 * 8
    assert C.sizeof(_lib.AdamTensor) == 4 * 8 + 10 * 4
    assert C.sizeof(_lib.ShBwdView) == 3 * 8 + 2 * 4
    assert C.sizeof(_lib.ShColorView) == 4 * 8


def test_workspace_size_queries_need_no_gpu():
    import ctypes as C
    lib = _lib.lib()
    g, b, i, w = (C.c_size_t() for _ in range(4))
    assert lib.hgs_raster_ws_sizes(1_000_000, 1920, 1080, 2_667_604, C.byref(g), C.byref(b), C.byref(i), C.byref(w)) == 0
    assert g.value >= 1_000_000 * (64 + 4 + 8 + 12)
    assert b.value >= 2_667_604 * (16 + 12)
    assert i.value >= 1920 * 1080 * 8
    assert w.value >= 2_667_604 * 40             # 10 floats per (tile, Gaussian) record
    assert lib.hgs_raster_ws_sizes(-1, 10, 10, 0, None, None, None, None) != 0
    assert b"bad sizes" in lib.hgs_last_error()


def test_python_constants_match_the_header_macros():
    """The numbers the ctypes layer repeats (it cannot include the header) against include/hgs.h."""
    import ctypes as C
    src = open(os.path.join(ROOT, "include", "hgs.h")).read()
    macro = lambda name: int(re.search(rf"#define\s+{name}\s+(\d+)", src).group(1))
    assert macro("HGS_ABI_VERSION") == _lib.ABI_VERSION
    assert macro("HGS_INST_GRAD_STRIDE") == _lib.INST_GRAD_STRIDE
    assert macro("HGS_P2P_MAX_WORLD") == _lib.P2P_MAX_WORLD
    assert macro("HGS_P2P_HANDLE_BYTES") == _lib.P2P_HANDLE_BYTES
    assert macro("HGS_P2P_FLAG_BYTES") == _lib.P2P_FLAG_BYTES
    assert macro("HGS_RESID_COUNTER_WORDS") == _lib.RESID_COUNTER_WORDS
    assert macro("HGS_RESID_HOST_ROW_FLOATS") == _lib.RESID_HOST_ROW_FLOATS
    assert int(re.search(r"HGS_ERR_CAPACITY\s*=\s*(\d+)", src).group(1)) == _lib.ERR_CAPACITY
    assert C.sizeof(_lib.ResidRows) == 5 * 8
