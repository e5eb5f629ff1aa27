"""ctypes binding of libhgs.so (C ABI declared in include/hgs.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C
hierarchical-3d-gaussians_amd/csrc``.  There is no fallback: if the shared object is missing or no
HIP device is visible, every op raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libhgs.so")

ABI_VERSION = 7
INST_GRAD_STRIDE = 10          # floats per (tile, Gaussian) record of the backward scratch (HGS_INST_GRAD_STRIDE)
ERR_CAPACITY = 5


class RasterArgs(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("M", C.c_int32), ("sh_degree", C.c_int32),
        ("width", C.c_int32), ("height", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("do_depth", C.c_int32), ("debug", C.c_int32), ("accumulate_grads", C.c_int32),
        ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
        ("cov3D_precomp", C.c_void_p), ("interpolation_weights", C.c_void_p), ("num_node_kids", C.c_void_p),
        ("shs_rest", C.c_void_p), ("activations", C.c_int32), ("defer_sh_bwd", C.c_int32), ("lod_per_pixel", C.c_int32), ("reserved1", C.c_int32),
        ("prepare_backward", C.c_int32), ("lod_n", C.c_int32),
        ("lod_render_indices", C.c_void_p), ("lod_parent_indices", C.c_void_p),
        ("lod_rows", C.c_int32), ("lod_scatter", C.c_int32),
    ]


ACT_SCALE_EXP, ACT_ROT_NORMALIZE, ACT_OPACITY_SIGMOID, ACT_OPACITY_ABS = 1, 2, 4, 8


class RasterGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "dL_dmeans3D", "dL_dmeans2D", "dL_dshs", "dL_dcolors", "dL_dopacity",
        "dL_dscales", "dL_drotations", "dL_dcov3D", "dL_dshs_rest")]


class RasterViews(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "tile_ids_sorted", "point_list", "ranges", "tiles_touched", "offsets", "depths", "rects",
        "records", "final_T", "n_contrib")]


class HierHost(C.Structure):
    _fields_ = [("P", C.c_int32), ("N", C.c_int32), ("M", C.c_int32), ("reserved", C.c_int32),
                ("xyz", C.c_void_p), ("shs", C.c_void_p), ("alpha", C.c_void_p),
                ("log_scales", C.c_void_p), ("rots", C.c_void_p), ("nodes", C.c_void_p),
                ("boxes", C.c_void_p)]


class ResidRows(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("means3D", "shs", "opacities", "scales", "rotations")]


class ShBwdView(C.Structure):
    _fields_ = [("geom_ws", C.c_void_p), ("bwd_ws", C.c_void_p), ("campos", C.c_void_p), ("L", C.c_uint32),
                ("reserved", C.c_uint32)]


class ShColorView(C.Structure):
    _fields_ = [("campos", C.c_void_p), ("rgb", C.c_void_p), ("clamp", C.c_void_p), ("d_rgb", C.c_void_p)]


MAX_DEFERRED_VIEWS = 8
HIER_UPSTREAM, HIER_PRIVATE, HIER_UPSTREAM_HALF = 0, 1, 2


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("row_len", C.c_int32), ("step_size", C.c_float), ("beta1", C.c_float), ("one_minus_beta1", C.c_float),
                ("beta2", C.c_float), ("one_minus_beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("bias_correction2_sqrt", C.c_float), ("reserved", C.c_int32)]


ADAM_MAX_TENSORS = 8

# symbol -> (restype, argtypes); also the list the export test checks against include/hgs.h
_P = C.c_void_p
SIGNATURES = {
    "hgs_abi_version": (C.c_int, []),
    "hgs_last_error": (C.c_char_p, []),
    "hgs_device_count": (C.c_int, []),
    "hgs_raster_ws_sizes": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "hgs_raster_fwd_stage1": (C.c_int, [C.POINTER(RasterArgs), _P, _P, C.POINTER(C.c_uint32), _P, C.c_int]),
    "hgs_raster_fwd_stage2": (C.c_int, [C.POINTER(RasterArgs), _P, _P, _P, C.c_uint32, _P, _P, _P, C.c_int]),
    "hgs_raster_fwd": (C.c_int, [C.POINTER(RasterArgs), _P, _P, _P, C.c_uint32, _P, _P, _P, C.POINTER(C.c_uint32), _P,
                                 C.c_int]),
    "hgs_release_device_state": (C.c_int, [C.c_int]),
    "hgs_raster_bwd": (C.c_int, [C.POINTER(RasterArgs), _P, _P, _P, _P, C.c_uint32, _P, _P, _P, _P,
                                 C.POINTER(RasterGrads), _P, C.c_int]),
    "hgs_raster_views_get": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_uint32, _P, _P, _P,
                                       C.POINTER(RasterViews)]),
    "hgs_sort_tmp_bytes": (C.c_size_t, [C.c_uint32]),
    "hgs_sort_pairs": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint32, C.c_int, _P, C.c_int]),
    "hgs_timing_enable": (C.c_int, [C.c_int]),
    "hgs_timing_stage_count": (C.c_int, []),
    "hgs_timing_stage_name": (C.c_char_p, [C.c_int]),
    "hgs_timing_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_int]),
    "hgs_expand_tmp_bytes": (C.c_size_t, [C.c_int32]),
    "hgs_expand_to_size": (C.c_int, [_P, _P, C.c_int32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     _P, _P, _P, C.c_int32, _P, C.POINTER(C.c_int32), _P, C.c_int]),
    "hgs_expand_to_size_nested": (C.c_int, [_P, _P, C.c_int32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                            _P, _P, _P, C.c_int32, _P, C.POINTER(C.c_int32), _P, C.c_int]),
    "hgs_hier_boxes_nested": (C.c_int, [_P, _P, C.c_int32, _P, C.POINTER(C.c_int32), _P, C.c_int]),
    "hgs_interp_weights": (C.c_int, [_P, C.c_int32, C.c_float, _P, _P, C.c_int32, C.POINTER(C.c_float),
                                     C.POINTER(C.c_float), _P, _P, _P, C.c_int]),
    "hgs_lod_gather": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int]),
    "hgs_lod_gather_bwd": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                     _P, _P, C.c_int]),
    "hgs_raster_sh_bwd_batched": (C.c_int, [C.POINTER(ShBwdView), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P,
                                            _P, _P, C.c_int32, _P, C.c_int]),
    "hgs_sh_colors_batched": (C.c_int, [C.POINTER(ShColorView), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P,
                                        C.c_int]),
    "hgs_sh_colors_batched_bwd": (C.c_int, [C.POINTER(ShColorView), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P,
                                            _P, _P, C.c_int32, _P, C.c_int]),
    "hgs_adam_step": (C.c_int, [C.POINTER(AdamTensor), C.c_int32, C.c_int64, _P, C.c_int64, _P, _P, C.c_int]),
    "hgs_knn_tmp_bytes": (C.c_size_t, [C.c_int32]),
    "hgs_dist2_knn3": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int]),
    "hgs_hier_load": (C.c_int, [C.c_char_p, C.POINTER(HierHost)]),
    "hgs_hier_write": (C.c_int, [C.c_char_p, C.POINTER(HierHost)]),
    "hgs_hier_free": (None, [C.POINTER(HierHost)]),
    "hgs_p2p_alloc": (C.c_int, [C.c_size_t, C.c_int32, C.POINTER(_P), C.c_int]),
    "hgs_p2p_free": (C.c_int, [_P, C.c_int]),
    "hgs_p2p_export": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "hgs_p2p_open": (C.c_int, [C.c_char_p, C.POINTER(_P), C.c_int]),
    "hgs_p2p_close": (C.c_int, [_P, C.c_int]),
    "hgs_p2p_allreduce_sum": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(_P), C.POINTER(_P), C.c_size_t, C.c_size_t,
                                        C.c_uint32, _P, C.c_int]),
    "hgs_host_alloc": (C.c_void_p, [C.c_size_t]),
    "hgs_host_free": (None, [_P]),
    "hgs_resid_mark": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P, C.c_uint32, _P, _P, _P, _P,
                                 C.POINTER(C.c_uint32), _P, C.c_int]),
    "hgs_resid_evict": (C.c_int, [_P, _P, _P, C.c_int32, C.c_uint32, C.c_uint32, _P, _P, C.POINTER(C.c_uint32), _P,
                                  C.c_int]),
    "hgs_resid_fetch": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, _P, _P, _P, C.c_uint32, _P,
                                  C.POINTER(ResidRows), C.c_int32, _P, C.c_int]),
    "hgs_resid_remap": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P, _P, C.c_int]),
}
P2P_MAX_WORLD, P2P_HANDLE_BYTES, P2P_FLAG_BYTES = 8, 64, 256
RESID_COUNTER_WORDS = 68
RESID_HOST_ROW_FLOATS = 64     # packed host row: [0, 3M) SH, [48, 52) rotation, [52, 55) mean, [55, 58) scale, [58] opacity

_lib = None


def lib():
    """Load libhgs.so once; raise (never fall back) if it is absent or has the wrong ABI."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libhgs.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C hierarchical-3d-gaussians_amd/csrc`). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.hgs_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libhgs.so ABI {l.hgs_abi_version()} != expected {ABI_VERSION}; rebuild")
        _lib = l
    return _lib


class HgsError(RuntimeError):
    """A non-zero return code of the C ABI (``.code``: HGS_ERR_*, include/hgs.h); a RuntimeError, as the reference's
    extensions raise."""

    def __init__(self, message: str, code: int):
        super().__init__(message)
        self.code = code


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().hgs_last_error()
        raise HgsError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}", rc)


def ptr(t):
    """data_ptr of a tensor or None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def timing_enable(on, stages=None):
    """on: bool; stages: optional iterable of stage names -- time only those (fewer events on the stream)."""
    l = lib()
    mask = 1 if on else 0
    if on and stages:
        names = [l.hgs_timing_stage_name(i).decode() for i in range(l.hgs_timing_stage_count())]
        mask = 0
        for s in stages:
            mask |= 2 << names.index(s)
    check(l.hgs_timing_enable(mask), "hgs_timing_enable")


def timing_read(reset=True):
    """-> {stage: (total_ms, calls)} accumulated since the last reset."""
    l = lib()
    n = l.hgs_timing_stage_count()
    ms = (C.c_double * n)()
    calls = (C.c_uint32 * n)()
    check(l.hgs_timing_read(ms, calls, 1 if reset else 0), "hgs_timing_read")
    return {l.hgs_timing_stage_name(i).decode(): (ms[i], calls[i]) for i in range(n)}
