"""The binding shown in INTEGRATION.md, executed literally (plain ctypes over libhgs.so, no package code): it must
produce the image of the drop-in package.  Guards the documentation against drifting from the ABI."""
import ctypes as C

import pytest
import torch

import parity as pa
from hgs import _lib, synth

pytestmark = pytest.mark.gpu


def test_documented_ctypes_binding_renders_the_same_image(gpu):
    lib = C.CDLL(_lib.LIB_PATH)

    class RasterArgs(C.Structure):            # == struct hgs_raster_args in include/hgs.h (as printed in INTEGRATION.md)
        _fields_ = [("P", C.c_int32), ("M", C.c_int32), ("sh_degree", C.c_int32), ("width", C.c_int32),
                    ("height", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                    ("scale_modifier", C.c_float), ("do_depth", C.c_int32), ("debug", C.c_int32),
                    ("accumulate_grads", C.c_int32)] + \
                   [(n, C.c_void_p) for n in ("bg", "viewmatrix", "projmatrix", "campos", "means3D", "shs",
                    "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp",
                    "interpolation_weights", "num_node_kids", "shs_rest")] + \
                   [("activations", C.c_int32), ("defer_sh_bwd", C.c_int32), ("lod_per_pixel", C.c_int32), ("reserved1", C.c_int32),
                    ("prepare_backward", C.c_int32), ("lod_n", C.c_int32), ("lod_render_indices", C.c_void_p),
                    ("lod_parent_indices", C.c_void_p), ("lod_rows", C.c_int32), ("lod_scatter", C.c_int32)]

    assert C.sizeof(RasterArgs) == C.sizeof(_lib.RasterArgs)
    W, H, P = 320, 180, 5000
    cam = synth.make_camera(W, H)
    sc = synth.make_scene(P, cam, seed=2).to(gpu)
    bg = torch.zeros(3, device=gpu)
    view, proj, campos = (t.to(gpu).contiguous() for t in (cam.world_view_transform, cam.full_proj_transform,
                                                           cam.camera_center))
    a = RasterArgs(P=P, M=16, sh_degree=3, width=W, height=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                   scale_modifier=1.0, do_depth=1, bg=bg.data_ptr(), viewmatrix=view.data_ptr(),
                   projmatrix=proj.data_ptr(), campos=campos.data_ptr(), means3D=sc.means3D.data_ptr(),
                   shs=sc.shs.data_ptr(), opacities=sc.opacities.data_ptr(), scales=sc.scales.data_ptr(),
                   rotations=sc.rotations.data_ptr())
    vp = C.c_void_p
    stream = vp(torch.cuda.current_stream().cuda_stream)
    sz = [C.c_size_t() for _ in range(4)]
    lib.hgs_raster_ws_sizes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32] + [C.POINTER(C.c_size_t)] * 4
    assert lib.hgs_raster_ws_sizes(P, W, H, 0, *map(C.byref, sz)) == 0
    geom, img = (torch.empty(s.value, dtype=torch.uint8, device=gpu) for s in (sz[0], sz[2]))
    radii, L = torch.empty(P, dtype=torch.int32, device=gpu), C.c_uint32()
    lib.hgs_raster_fwd_stage1.argtypes = [C.POINTER(RasterArgs), vp, vp, C.POINTER(C.c_uint32), vp, C.c_int]
    assert lib.hgs_raster_fwd_stage1(C.byref(a), vp(geom.data_ptr()), vp(radii.data_ptr()), C.byref(L), stream, 0) == 0
    assert lib.hgs_raster_ws_sizes(P, W, H, L.value, None, C.byref(sz[1]), None, C.byref(sz[3])) == 0
    binb = torch.empty(sz[1].value, dtype=torch.uint8, device=gpu)
    color, invd = torch.empty(3, H, W, device=gpu), torch.empty(1, H, W, device=gpu)
    lib.hgs_raster_fwd_stage2.argtypes = [C.POINTER(RasterArgs), vp, vp, vp, C.c_uint32, vp, vp, vp, C.c_int]
    assert lib.hgs_raster_fwd_stage2(C.byref(a), vp(geom.data_ptr()), vp(binb.data_ptr()), vp(img.data_ptr()), L.value,
                                     vp(color.data_ptr()), vp(invd.data_ptr()), stream, 0) == 0
    torch.cuda.synchronize()
    import diff_gaussian_rasterization as dgr
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, do_depth=True, device=gpu))
    with torch.no_grad():
        c2, r2, d2 = dgr.GaussianRasterizer(rs)(means3D=sc.means3D, means2D=torch.zeros(P, 3, device=gpu), shs=sc.shs,
                                                opacities=sc.opacities, scales=sc.scales, rotations=sc.rotations)
    assert torch.equal(color, c2) and torch.equal(invd, d2) and torch.equal(radii, r2)
    assert L.value > 0 and float(color.max()) > 0.05
