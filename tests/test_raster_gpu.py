"""GPU parity tests of the rasterizer hot path: HIP (through the C ABI) vs the CPU oracle.

Mirrors how the reference calls the op (gaussian_renderer/__init__.py:44-64,105-113 for the
single-chunk path; :247-277 for the hierarchy path)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import parity as pa
from hgs import synth

pytestmark = pytest.mark.gpu

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _log(name, payload):
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "parity_log.jsonl"), "a") as f:
            f.write(json.dumps({"case": name, **payload}, default=float) + "\n")
    except OSError:
        pass


def _assert_case(name, hip, oo, og, do_depth=True, mixed_tol=1.0):
    idx = pa.check_indices(hip, oo)
    st = pa.compare(hip, oo, og, do_depth=do_depth)
    _log(name, {"indices": idx, "stats": st})
    print(name, "indices", idx)
    for k, v in st.items():
        print("   ", k, v)
    assert all(v == 0 for v in idx.values()), f"{name}: integer mismatch {idx}"
    assert st["fragile_frac"] <= pa.FRAGILE_FRAC
    pa.assert_stats(name, st, mixed_tol=mixed_tol)


def test_config1_1k_128(gpu):
    """BASELINE.json configs[0]: 1k random Gaussians, 128x128, fwd + bwd."""
    cam, scene, gc, gd = pa.default_case(1000, 128, 128)
    bg = torch.tensor([0.1, 0.2, 0.3])
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
    _assert_case("config1", hip, oo, og)


def test_ragged_image_and_sh_degrees(gpu):
    """Image size not a multiple of 16; every active SH degree."""
    for deg in (0, 1, 2, 3):
        cam, scene, gc, gd = pa.default_case(700, 200, 120, seed=3 + deg)
        scene.sh_degree = deg
        bg = torch.tensor([0.0, 0.0, 0.0])
        oo, og = pa.run_oracle(scene, cam, bg, gc, gd)
        hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
        _assert_case(f"ragged_deg{deg}", hip, oo, og)


def test_dense_early_termination(gpu):
    """Large opaque Gaussians: exercises T < 1e-4 termination and the 0.99 alpha cap."""
    cam = synth.make_camera(96, 80)
    scene = synth.make_scene(1500, cam, seed=11, s_px=(3.0, 12.0))
    scene.opacities = (0.6 + 0.39 * torch.rand(scene.P, 1, generator=torch.Generator().manual_seed(5)))
    gc, gd = synth.upstream_grads(80, 96)
    bg = torch.tensor([1.0, 1.0, 1.0])
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
    assert (oo.final_T < 1e-3).mean() > 0.2, "case should saturate a good share of the pixels"
    _assert_case("dense", hip, oo, og)


def test_precomputed_colour_and_covariance(gpu):
    """The pipe.convert_SHs_python / pipe.compute_cov3D_python branches
    (gaussian_renderer/__init__.py:75-76,84-89): colours and 3D covariances handed in."""
    from oracle import raster_oracle as ro
    cam, scene, gc, gd = pa.default_case(800, 128, 96, seed=21)
    bg = torch.tensor([0.2, 0.1, 0.0])
    cols = torch.rand(scene.P, 3, generator=torch.Generator().manual_seed(9))
    cov = torch.from_numpy(ro.cov3d_spec(scene.scales.numpy(), scene.rotations.numpy(), 1.0))
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd, colors_precomp=cols, cov3D_precomp=cov)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu, colors_precomp=cols, cov3D_precomp=cov)
    _assert_case("precomp", hip, oo, og)


def test_python_twins_agree(gpu):
    """op(shs) == op(colors_precomp = clamp(eval_sh + 0.5)) and op(scales, rot) == op(cov3D):
    the equivalence the reference's --convert_SHs_python / --compute_cov3D_python flags rely on."""
    from oracle import raster_oracle as ro
    cam, scene, gc, gd = pa.default_case(600, 112, 112, seed=31)
    bg = torch.zeros(3)
    d = scene.means3D - cam.camera_center[None]
    d = d / d.norm(dim=1, keepdim=True)
    cols = torch.clamp_min(ro.eval_sh_torch(3, scene.shs, d) + 0.5, 0.0)
    cov = torch.from_numpy(ro.cov3d_spec(scene.scales.numpy(), scene.rotations.numpy(), 1.0))
    a = pa.run_hip(scene, cam, bg, gc, gd, gpu)
    b = pa.run_hip(scene, cam, bg, gc, gd, gpu, colors_precomp=cols, cov3D_precomp=cov)
    assert torch.equal(a["radii"], b["radii"])
    assert (a["color"] - b["color"]).abs().max() < 2e-6
    assert (a["grads"]["means2D"] - b["grads"]["means2D"]).abs().max() <= 1e-5 * a["grads"]["means2D"].abs().max()


def test_hierarchy_mode_opacity(gpu):
    """render_post path: interpolation_weights / num_node_kids non-empty, do_depth False, opacities may
    exceed 1 (abs activation, scene/gaussian_model.py:393)."""
    cam, scene, gc, gd = pa.default_case(900, 128, 128, seed=41)
    g = torch.Generator().manual_seed(42)
    scene.opacities = scene.opacities * 1.3
    w = torch.rand(scene.P + 50, generator=g)
    kids = torch.randint(1, 5, (scene.P + 50,), generator=g, dtype=torch.int32)
    bg = torch.zeros(3)
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd, interpolation_weights=w, num_node_kids=kids, do_depth=False)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu, interpolation_weights=w, num_node_kids=kids, do_depth=False)
    _assert_case("hier_opacity", hip, oo, og, do_depth=False)


def test_scale_modifier_and_moved_camera(gpu):
    cam = synth.orbit_camera(160, 96, 1, 5)
    scene = synth.make_scene(900, cam, seed=51)
    gc, gd = synth.upstream_grads(96, 160)
    bg = torch.tensor([0.3, 0.3, 0.3])
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd, scale_modifier=0.7)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu, scale_modifier=0.7)
    _assert_case("scale_mod", hip, oo, og)


def test_edge_cases(gpu):
    """Empty scene, everything behind the near plane, a single Gaussian."""
    import diff_gaussian_rasterization as dgr
    cam = synth.make_camera(64, 48)
    bg = torch.tensor([0.25, 0.5, 0.75])
    # (1) P = 0
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, bg, 3, device=gpu))
    z = lambda *s: torch.zeros(*s, device=gpu)
    color, radii, invd = dgr.GaussianRasterizer(rs)(means3D=z(0, 3), means2D=z(0, 3), shs=z(0, 16, 3),
                                                    opacities=z(0, 1), scales=z(0, 3), rotations=z(0, 4))
    assert torch.allclose(color.cpu(), bg[:, None, None].expand(3, 48, 64))
    assert invd.abs().max() == 0 and radii.numel() == 0
    # (2) all culled: z <= 0.2
    scene = synth.make_scene(100, cam, seed=1)
    scene.means3D[:, 2] = 0.1
    gc, gd = synth.upstream_grads(48, 64)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
    assert (hip["radii"] == 0).all() and hip["L"] == 0
    assert torch.allclose(hip["color"], bg[:, None, None].expand(3, 48, 64))
    assert all(float(g.abs().max()) == 0.0 for g in hip["grads"].values())
    # (3) one Gaussian on the optical axis: analytic known answer (SURVEY App. D KAT 1)
    one = synth.Scene(torch.tensor([[0.0, 0.0, 4.0]]), torch.full((1, 3), 0.05), torch.tensor([[1.0, 0, 0, 0]]),
                      torch.tensor([[0.8]]), torch.zeros(1, 16, 3), 3)
    one.shs[0, 0] = torch.tensor([1.0, 0.0, -1.0])
    hip = pa.run_hip(one, cam, torch.zeros(3), gc, gd, gpu)
    fy = 48 / (2 * cam.tanfovy)
    var = (fy * 0.05 / 4.0) ** 2 + 0.3
    cx, cy = (64 - 1) / 2.0, (48 - 1) / 2.0
    ys, xs = torch.meshgrid(torch.arange(48.0), torch.arange(64.0), indexing="ij")
    alpha = torch.clamp(0.8 * torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * var)), max=0.99)
    alpha = torch.where(alpha < 1 / 255.0, torch.zeros_like(alpha), alpha)
    rad = int(np.ceil(3 * np.sqrt(var)))
    assert int(hip["radii"][0]) == rad
    c0 = 0.5 + 0.28209479177387814
    tile_mask = torch.zeros(48, 64, dtype=torch.bool)
    x0, x1 = int((cx - rad) // 16), int((cx + rad + 15) // 16)
    y0, y1 = int((cy - rad) // 16), int((cy + rad + 15) // 16)
    tile_mask[y0 * 16:y1 * 16, x0 * 16:x1 * 16] = True
    expect = torch.where(tile_mask, alpha, torch.zeros_like(alpha))
    assert (hip["color"][0] - c0 * expect).abs().max() < 2e-6
    assert (hip["color"][1] - 0.5 * expect).abs().max() < 2e-6
    assert (hip["invdepth"][0] - expect / 4.0).abs().max() < 2e-6


def test_rejects_bad_inputs(gpu):
    import diff_gaussian_rasterization as dgr
    cam = synth.make_camera(32, 32)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, device=gpu))
    r = dgr.GaussianRasterizer(rs)
    z = lambda *s: torch.zeros(*s, device=gpu)
    with pytest.raises(Exception):
        r(means3D=z(4, 3), means2D=z(4, 3), opacities=z(4, 1), scales=z(4, 3), rotations=z(4, 4))  # no colour
    with pytest.raises(Exception):
        r(means3D=z(4, 3), means2D=z(4, 3), shs=z(4, 16, 3), opacities=z(4, 1), scales=z(4, 3))   # no rotation
    with pytest.raises(RuntimeError):
        r(means3D=torch.zeros(4, 3), means2D=z(4, 3), shs=z(4, 16, 3), opacities=z(4, 1), scales=z(4, 3),
          rotations=z(4, 4))                                                                         # CPU tensor


@pytest.mark.parametrize("n,end_bit", [(1, 8), (63, 13), (4097, 38), (100000, 45), (1 << 20, 47), (300001, 64),
                                       (9_000_001, 40)])   # last case: the 16-keys-per-lane path (>= 8 Mi keys)
def test_sort_pairs_is_a_stable_sort(gpu, n, end_bit):
    """K4 alone: bit-exact against numpy's stable argsort, with many duplicate keys."""
    import ctypes as C
    from hgs import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(n)
    hi = (1 << end_bit) - 1 if end_bit < 64 else (1 << 64) - 1
    keys = rng.integers(0, min(hi, 1 << 62), size=n, dtype=np.uint64)
    keys[rng.random(n) < 0.5] &= np.uint64(0xFF)            # heavy duplication
    keys &= np.uint64(hi)
    vals = np.arange(n, dtype=np.uint32)
    order = np.argsort(keys, kind="stable")
    k_in = torch.from_numpy(keys.view(np.int64)).to(gpu)
    v_in = torch.from_numpy(vals.view(np.int32)).to(gpu)
    k_out, v_out = torch.empty_like(k_in), torch.empty_like(v_in)
    tmp = torch.empty(lib.hgs_sort_tmp_bytes(n), dtype=torch.uint8, device=gpu)
    p = _lib.ptr
    _lib.check(lib.hgs_sort_pairs(p(k_in), p(v_in), p(k_out), p(v_out), p(tmp), n, end_bit,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream), 0), "sort")
    torch.cuda.synchronize()
    assert np.array_equal(k_out.cpu().numpy().view(np.uint64), keys[order])
    assert np.array_equal(v_out.cpu().numpy().view(np.uint32), vals[order])
    assert np.array_equal(k_in.cpu().numpy().view(np.uint64), keys), "input must be left intact"


def test_full_size_properties(gpu):
    """BASELINE metric size (1080p, 1M Gaussians): size-independent properties instead of the oracle --
    sortedness of the keys, ranges consistent with the keys, transmittance in [0,1], determinism of the
    forward, finite gradients, zero gradient for culled Gaussians."""
    cam = synth.make_camera(1920, 1080)
    scene = synth.make_scene(1_000_000, cam, seed=0)
    gc, gd = synth.upstream_grads(1080, 1920)
    bg = torch.zeros(3)
    a = pa.run_hip(scene, cam, bg, gc, gd, gpu, debug=False)
    b = pa.run_hip(scene, cam, bg, gc, gd, gpu, debug=False)
    pl0 = a["views"]["point_list"].numpy()
    keys = (a["views"]["tile_ids_sorted"].numpy().astype(np.uint64) << np.uint64(32)) | \
        a["views"]["depths"].numpy().view(np.uint32)[pl0].astype(np.uint64)
    assert a["L"] == int(a["views"]["tiles_touched"].numpy().astype(np.int64).sum())
    assert np.all(keys[1:] >= keys[:-1])
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    rng_ = a["views"]["ranges"].numpy().astype(np.int64)
    cnt = np.bincount(tiles, minlength=rng_.shape[0])
    assert np.array_equal(rng_[:, 1] - rng_[:, 0], cnt)
    ft = a["views"]["final_T"]
    assert float(ft.min()) >= 0.0 and float(ft.max()) <= 1.0
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["views"]["point_list"], b["views"]["point_list"])
    for k, g in a["grads"].items():
        assert torch.isfinite(g).all(), k
        assert torch.equal(g, b["grads"][k]), f"{k}: backward must be deterministic (S=4 path has no float atomics)"
    culled = a["radii"] == 0
    assert float(a["grads"]["means3D"][culled].abs().sum()) == 0.0
    # depth order inside every tile range
    pl = a["views"]["point_list"].numpy()
    d = a["views"]["depths"].numpy()[pl]
    same_tile = tiles[1:] == tiles[:-1]
    assert np.all(d[1:][same_tile] >= d[:-1][same_tile])


def test_4k_8m_gaussians_forward_backward(gpu):
    """Scale / capacity case in the direction of BASELINE config 5 (4K frame, many Gaussians): 8 M Gaussians
    at 3840x2160 (L ~ 21 M tile instances).  Pixel parity against the oracle on a random sample of tiles,
    index invariants on the whole frame, finite deterministic gradients."""
    from oracle import raster_oracle as ro
    W, H, P = 3840, 2160, 8_000_000
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=0, sh_degree=3)
    gc, gd = synth.upstream_grads(H, W)
    bg = torch.tensor([0.05, 0.1, 0.15])
    a = pa.run_hip(scene, cam, bg, gc, gd, gpu, debug=False)
    assert a["L"] > 2 * P
    tiles_sorted = a["views"]["tile_ids_sorted"].numpy()
    assert np.all(tiles_sorted[1:] >= tiles_sorted[:-1])
    rng_ = a["views"]["ranges"].numpy().astype(np.int64)
    assert np.array_equal(rng_[:, 1] - rng_[:, 0], np.bincount(tiles_sorted, minlength=rng_.shape[0]))
    pl = a["views"]["point_list"].numpy()
    d = a["views"]["depths"].numpy()[pl]
    same = tiles_sorted[1:] == tiles_sorted[:-1]
    assert np.all(d[1:][same] >= d[:-1][same])
    for k, g in a["grads"].items():
        assert torch.isfinite(g).all(), k
    # pixel parity on 24 random tiles (the oracle renders only those)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    tiles = sorted(np.random.default_rng(7).choice(T, size=24, replace=False).tolist())
    oo = ro.rasterize(scene.means3D, None, scene.shs, None, scene.opacities, scene.scales, scene.rotations, None,
                      image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg,
                      scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                      sh_degree=3, campos=cam.camera_center, tiles=tiles)
    assert int((a["radii"].numpy() != oo.geom.radii).sum()) == 0
    assert a["L"] == oo.binning.num_rendered
    assert np.array_equal(pl, oo.binning.point_list), "sorted instance list must be bit-exact at scale too"
    gx = (W + 15) // 16
    worst = 0.0
    for t in tiles:
        y0, x0 = (t // gx) * 16, (t % gx) * 16
        ok = torch.from_numpy(~oo.fragile[y0:y0 + 16, x0:x0 + 16])
        hip_t = a["color"][:, y0:y0 + 16, x0:x0 + 16][:, ok]
        ref_t = oo.color[:, y0:y0 + 16, x0:x0 + 16][:, ok]
        worst = max(worst, float((hip_t.double() - ref_t).abs().max()))
        hd = a["invdepth"][:, y0:y0 + 16, x0:x0 + 16][:, ok]
        rd = oo.invdepth[:, y0:y0 + 16, x0:x0 + 16][:, ok]
        worst = max(worst, float((hd.double() - rd).abs().max()))
    print("4k sample max abs pixel error", worst)
    assert worst <= 1e-5


@pytest.mark.parametrize("P,W,H,lo,hi", [(6_000, 64, 48, 512, 1024), (12_000, 64, 48, 1024, 2048),
                                         (18_000, 64, 48, 2048, 4096), (22_000, 64, 48, 2048, 4096),
                                         (45_000, 64, 48, 4096, 8192), (130_000, 64, 64, 8192, 16384),
                                         (1_500_000, 64, 48, 16384, 1 << 30)])
def test_crowded_tiles_sort_paths(gpu, P, W, H, lo, hi):
    """Every class of the per-tile depth sort: one wave with 16 keys per lane (513 .. 1024 instances), two waves per
    tile and two tiles at a time (.. 2048: the 12 k case), a workgroup holding both pair-class and four-wave tiles (the
    18 k case: 1 886 .. 2 767 per tile), four waves (.. 4096: the 22 k case), 512 lanes (.. 8192, two tiles per CU: the 45 k
    case), 1024 lanes (.. 16 Ki) and the global radix fallback (> 16 Ki): the sorted instance list must still be bit-exact.  Forward only; oracle = geometry +
    binning spec."""
    import diff_gaussian_rasterization as dgr
    from oracle import raster_oracle as ro
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=13, s_px=(0.3, 1.5))
    # quantise depths so that many instances tie and the Gaussian-index tie-break is exercised
    scene.means3D[:, 2] = torch.round(scene.means3D[:, 2] * 4) / 4
    geom = ro.geometry_spec(scene.means3D.numpy(), scene.scales.numpy(), scene.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            cam.tanfovx, cam.tanfovy, 1.0)
    binning = ro.binning_spec(geom)
    per_tile = binning.ranges[:, 1] - binning.ranges[:, 0]
    assert lo < per_tile.max() <= hi, per_tile.max()
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, device=gpu))
    sc = scene.to(gpu)
    with torch.no_grad():
        color, radii, _ = dgr._RasterizeGaussians.apply(sc.means3D, torch.zeros(P, 3, device=gpu), sc.shs, None,
                                                         sc.opacities, sc.scales, sc.rotations, None, rs, None)
    assert torch.isfinite(color).all()
    assert np.array_equal(radii.cpu().numpy(), geom.radii)
    # no autograd graph under no_grad: fetch the views through a fresh forward call object
    L, color2, radii2, geomb, binb, img, invd, call = dgr._C.rasterize_gaussians(
        rs.bg, sc.means3D, None, sc.opacities, sc.scales, sc.rotations, 1.0, None, rs.viewmatrix, rs.projmatrix,
        rs.tanfovx, rs.tanfovy, H, W, sc.shs, 3, rs.campos, False, False, rs.render_indices, rs.parent_indices,
        rs.interpolation_weights, rs.num_node_kids, True)
    v = dgr._C.raster_views(call)
    assert L == binning.num_rendered
    assert np.array_equal(v["ranges"].cpu().numpy(), binning.ranges)
    assert np.array_equal(v["point_list"].cpu().numpy(), binning.point_list)
    assert torch.equal(color, color2)


@pytest.mark.parametrize("n,lo,hi", [(1500, 1024, 2048), (5200, 4096, 8192), (19000, 16384, 1 << 30)])
def test_one_crowded_tile_in_a_light_frame(gpu, n, lo, hi):
    """A light frame (mean list far below 512: the one-wave depth sort without the four-wave stage) with ONE tile holding
    ~1 500 / ~5 000 / ~19 000 instances: the workgroup that owns the tile sorts it with the radix fallback (round 5: no
    oversized-classes launch behind a light frame)."""
    import diff_gaussian_rasterization as dgr
    from oracle import raster_oracle as ro
    W, H = 640, 368
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(20_000 + n, cam, seed=17, s_px=(0.4, 1.5))
    g = torch.Generator().manual_seed(18)
    # a cluster projecting into the tile around pixel (328, 200)
    z = 4.0 + 2.0 * torch.rand(n, generator=g)
    fx = W / (2.0 * cam.tanfovx)
    scene.means3D[:n, 0] = (328.0 + 6.0 * (torch.rand(n, generator=g) - 0.5) - W / 2) / fx * z
    scene.means3D[:n, 1] = (200.0 + 6.0 * (torch.rand(n, generator=g) - 0.5) - H / 2) / fx * z
    scene.means3D[:n, 2] = torch.round(z * 8) / 8          # depth ties: the Gaussian-index tie-break
    scene.scales[:n] = (z * 0.5 / fx)[:, None] * torch.ones(n, 3)
    geom = ro.geometry_spec(scene.means3D.numpy(), scene.scales.numpy(), scene.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            cam.tanfovx, cam.tanfovy, 1.0)
    binning = ro.binning_spec(geom)
    per_tile = binning.ranges[:, 1] - binning.ranges[:, 0]
    assert lo < per_tile.max() <= hi and binning.num_rendered < 400 * per_tile.shape[0], (per_tile.max(), binning.num_rendered)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, device=gpu))
    sc = scene.to(gpu)
    L, color, radii, geomb, binb, img, invd, call = dgr._C.rasterize_gaussians(
        rs.bg, sc.means3D, None, sc.opacities, sc.scales, sc.rotations, 1.0, None, rs.viewmatrix, rs.projmatrix,
        rs.tanfovx, rs.tanfovy, H, W, sc.shs, 3, rs.campos, False, False, rs.render_indices, rs.parent_indices,
        rs.interpolation_weights, rs.num_node_kids, True)
    v = dgr._C.raster_views(call)
    assert L == binning.num_rendered
    assert np.array_equal(v["ranges"].cpu().numpy(), binning.ranges)
    assert np.array_equal(v["point_list"].cpu().numpy(), binning.point_list)


_SS_SIZES = [127, 129, 192, 193, 256, 257, 384, 385, 512, 513, 768, 769, 1024, 1025, 1280, 1281, 1536, 1537, 2048, 2049,
             2560, 2561, 3072, 3073, 4096, 4097, 5000]


@pytest.mark.parametrize("mode", ["uniform", "all_equal", "two_values", "outliers", "geometric", "runs_of_ties"])
@pytest.mark.parametrize("grid", [8, 40])
def test_sample_sort_list_sizes_and_depth_distributions(gpu, grid, mode):
    """The per-tile depth sort (binning.hip: sample sort, round 6) on lists of every length at which it changes its keys
    per lane, its group size or its route (127 .. 5 000 instances in one tile), with depth distributions that stress the
    quantile sample and the interpolated fine buckets: uniform; ALL depths equal (every key in one bucket: the overflow
    route to the network); two values; a tight cluster with a few far outliers (the skybox-behind-a-wall case); geometric;
    runs of 40 equal depths.  grid 8 x 8 tiles: a frame of long lists (the four-wave kernel); 40 x 40: a light frame.
    Ranges and sorted lists must equal the oracle's stable 64-bit sort bit for bit."""
    import diff_gaussian_rasterization as dgr
    from oracle import raster_oracle as ro
    W = H = grid * 16
    cam = synth.make_camera(W, H)
    sizes = _SS_SIZES
    P = sum(sizes)
    g = torch.Generator().manual_seed(100 + grid)
    scene = synth.make_scene(P, cam, seed=41, s_px=(0.3, 0.3))
    fx = W / (2.0 * cam.tanfovx)
    tiles = torch.randperm(grid * grid, generator=g)[:len(sizes)]
    tile_of = torch.repeat_interleave(tiles, torch.tensor(sizes))
    perm = torch.randperm(P, generator=g)                 # ids of a tile's instances are scattered over the index range
    tile_of = tile_of[perm]
    u = torch.rand(P, generator=g)
    if mode == "uniform":
        z = 2.0 + 18.0 * u
    elif mode == "all_equal":
        z = torch.full((P,), 5.0)
    elif mode == "two_values":
        z = torch.where(u < 0.5, torch.tensor(3.0), torch.tensor(7.5))
    elif mode == "outliers":
        z = 5.0 + 1e-3 * u
        z[torch.rand(P, generator=g) < 0.01] = 90.0
        z[torch.rand(P, generator=g) < 0.01] = 0.5
    elif mode == "geometric":
        z = 0.3 * torch.pow(300.0, u)
    else:
        z = 2.0 + torch.floor(u * P / 40.0) * (18.0 * 40.0 / P)
    px = (tile_of % grid).float() * 16 + 6.0 + 4.0 * torch.rand(P, generator=g)
    py = (tile_of // grid).float() * 16 + 6.0 + 4.0 * torch.rand(P, generator=g)
    scene.means3D[:, 0] = (px + 0.5 - W / 2) / fx * z
    scene.means3D[:, 1] = (py + 0.5 - H / 2) / fx * z
    scene.means3D[:, 2] = z
    scene.scales[:] = (z * 0.3 / fx)[:, None]
    geom = ro.geometry_spec(scene.means3D.numpy(), scene.scales.numpy(), scene.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            cam.tanfovx, cam.tanfovy, 1.0)
    binning = ro.binning_spec(geom)
    per_tile = binning.ranges[:, 1] - binning.ranges[:, 0]
    assert sorted(per_tile[per_tile > 0].tolist()) == sorted(sizes), "every Gaussian in exactly its own tile"
    assert (binning.num_rendered > 512 * grid * grid) == (grid == 8)             # long-list frame / light frame
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, device=gpu))
    sc = scene.to(gpu)
    for _ in range(2):                                    # (the second call takes the speculative single-call route)
        L, color, radii, geomb, binb, img, invd, call = dgr._C.rasterize_gaussians(
            rs.bg, sc.means3D, None, sc.opacities, sc.scales, sc.rotations, 1.0, None, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, H, W, sc.shs, 3, rs.campos, False, False, rs.render_indices, rs.parent_indices,
            rs.interpolation_weights, rs.num_node_kids, True)
        v = dgr._C.raster_views(call)
        assert L == binning.num_rendered
        assert np.array_equal(v["ranges"].cpu().numpy(), binning.ranges)
        assert np.array_equal(v["point_list"].cpu().numpy(), binning.point_list), mode


def test_gradient_accumulation_into_caller_buffers(gpu):
    """Data-parallel host path (RasterContext.grad_buffers): the backward writes straight into a flat bucket and
    ACCUMULATES the second view's gradients in place; the result must equal the sum of the two views' separately
    computed gradients.  Autograd receives None for the buffered inputs: ``loss.backward()`` leaves ``.grad`` alone
    (no double counting by AccumulateGrad)."""
    import diff_gaussian_rasterization as dgr
    from hgs import dp
    W, H, P = 160, 96, 1200
    base = synth.make_camera(W, H)
    scene = synth.make_scene(P, base, seed=3)
    cams = [synth.orbit_camera(W, H, j, 2, radius=0.3) for j in range(2)]
    gc, gd = synth.upstream_grads(H, W)
    bg = torch.zeros(3)
    sep = [pa.run_hip(scene, c, bg, gc, gd, gpu)["grads"] for c in cams]
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    sc = scene.to(gpu)
    params = {n: getattr(sc, n).clone().requires_grad_(True) for n in names}
    bucket = dp.GradBucket({n: tuple(v.shape) for n, v in params.items()}, gpu)
    bucket.flat.fill_(123.0)                      # stale contents must be overwritten by the first view
    rc = dgr.RasterContext(grad_buffers=bucket.views)
    for j, c in enumerate(cams):
        rc.grad_accumulate = j > 0
        rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(c, bg, 3, device=gpu))
        m2 = torch.zeros(P, 3, device=gpu, requires_grad=True)
        color, radii, invd = dgr.GaussianRasterizer(rs, context=rc)(
            means3D=params["means3D"], means2D=m2, shs=params["shs"], opacities=params["opacities"],
            scales=params["scales"], rotations=params["rotations"])
        ((color * gc.to(gpu)).sum() + (invd * gd.to(gpu)).sum()).backward()
        assert all(params[n].grad is None for n in names), "buffered gradients must not reach .grad"
        assert torch.allclose(m2.grad.cpu(), sep[j]["means2D"], rtol=1e-6, atol=1e-6)   # per-view, never accumulated
    for n in names:
        want = sep[0][n] + sep[1][n]
        assert torch.allclose(bucket.views[n].cpu(), want, rtol=1e-5, atol=1e-6 * float(want.abs().max())), n
    # a rasterizer WITHOUT the context, in the same process, is the plain autograd op again
    plain = pa.run_hip(scene, cams[0], bg, gc, gd, gpu)["grads"]
    assert all(torch.equal(plain[n], sep[0][n]) for n in names)


def test_awkward_inputs(gpu):
    """Screen-filling Gaussians (rectangle = whole tile grid), zero / tiny opacities, needle-like anisotropy,
    Gaussians straddling the near plane and the frustum border, off-centre principal point."""
    import math
    W, H = 208, 144
    fovy = math.radians(50.0)
    fy = H / (2 * math.tan(fovy / 2))
    fovx = 2 * math.atan(W / (2 * fy))
    wv = torch.eye(4)
    proj = synth.projection_matrix(0.01, 100.0, fovx, fovy, primx=0.42, primy=0.58).transpose(0, 1)
    cam = synth.Camera(W, H, fovx, fovy, wv, (wv @ proj).contiguous(), torch.zeros(3))
    g = torch.Generator().manual_seed(77)
    scene = synth.make_scene(400, cam, seed=5, s_px=(0.5, 6.0))
    P = scene.P
    scene.scales[:6] = torch.tensor([3.0, 2.0, 2.5])               # fill the whole screen
    scene.opacities[6:12] = 0.0                                    # never visible in the blend
    scene.opacities[12:18] = 1.0 / 300.0                           # below 1/255 everywhere
    scene.scales[18:40] = torch.tensor([0.4, 0.002, 0.002])        # needles
    scene.means3D[40:50, 2] = torch.linspace(0.15, 0.25, 10)       # around the 0.2 near plane
    scene.means3D[50:70, 0] = scene.means3D[50:70, 2] * cam.tanfovx * 1.4   # outside the frustum, EWA clamp region
    gc, gd = synth.upstream_grads(H, W)
    bg = torch.tensor([0.3, 0.2, 0.1])
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
    assert oo.geom.tiles_touched.max() == ((W + 15) // 16) * ((H + 15) // 16)
    # element-wise bound relaxed to 2 x (1e-5 |ref| + 1e-6 max|ref|) for THIS case only: the gradient of a screen-filling
    # Gaussian is a float32 sum over ~30 000 pixels (256 per tile inside K7, as the float32 reference lineage sums them)
    # with heavy cancellation -- measured 1.54 x the bound on d_scales, 1.26 x on d_rotations, norm-wise 4e-6 / 7e-6
    _assert_case("awkward", hip, oo, og, mixed_tol=2.0)
    assert float(hip["grads"]["opacities"][12:18].abs().max()) == 0.0


def test_closed_form_known_answers_on_the_device(gpu):
    """The HIP path against hand-derivable numbers, not against the oracle: five very wide
    layers of opacity 0.95 on one axis -- T goes 1 -> 0.05 -> 0.0025 -> 1.25e-4, the fourth layer would give
    6.25e-6 < 1e-4 and ends the pixel WITHOUT being blended: three layers with weights 0.95, 0.0475, 0.002375, final T
    1.25e-4, last contributor 3, and only the three blended layers receive gradients."""
    cam = synth.make_camera(32, 32)
    zs = [2.0, 3.0, 4.0, 5.0, 6.0]
    cols = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), (1.0, 1.0, 0.0), (0.0, 1.0, 1.0)]
    sc = synth.Scene(torch.tensor([[0.0, 0.0, z] for z in zs]), torch.tensor([[20.0 * z / 2.0] * 3 for z in zs]),
                     torch.tensor([[1.0, 0, 0, 0]] * 5), torch.full((5, 1), 0.95), torch.zeros(5, 16, 3), 0)
    for i, c in enumerate(cols):
        sc.shs[i, 0] = (torch.tensor(c) - 0.5) / 0.28209479177387814
    gc = torch.zeros(3, 32, 32); gc[:, 16, 16] = 1.0           # upstream gradient on the one pixel that is checked
    gd = torch.zeros(1, 32, 32)
    hip = pa.run_hip(sc, cam, torch.tensor([0.5, 0.5, 0.5]), gc, gd, gpu)
    y = x = 16
    w = [0.95, 0.95 * 0.05, 0.95 * 0.0025]
    expect = torch.tensor(w) + 1.25e-4 * 0.5
    assert float((hip["color"][:, y, x] - expect).abs().max()) < 2e-5
    assert abs(float(hip["invdepth"][0, y, x]) - (w[0] / 2.0 + w[1] / 3.0 + w[2] / 4.0)) < 2e-5
    assert int(hip["views"]["n_contrib"][y, x]) == 3
    assert abs(float(hip["views"]["final_T"][y, x]) - 1.25e-4) < 1e-7
    g = hip["grads"]["shs"][:, 0, :]                           # dL/d(dc) = C0 * weight of the layer, its own channel
    assert float(g[3:].abs().max()) == 0.0                     # layers 4, 5 were never blended into that pixel
    for i in range(3):
        assert abs(float(g[i, i]) - 0.28209479177387814 * w[i]) < 2e-6


def test_extremely_elongated_gaussians_stay_well_behaved(gpu):
    """Needles thousands of pixels long and half a pixel wide: their conic is positive definite only by a relative
    1e-6 .. 1e-8 of its entries, less than float32 resolves.  The kernels clamp the exponent at 0 instead of testing its
    sign, so K1 must keep the ROUNDED conic positive definite -- otherwise pixels far off the needle's axis would be
    blended at full opacity.  Checked against the float64 oracle with a tolerance that allows for the conditioning."""
    import math
    W, H = 160, 112
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(60, cam, seed=8)
    g = torch.Generator().manual_seed(4)
    fx = W / (2 * cam.tanfovx)
    z = scene.means3D[:, 2]
    long_px = torch.exp(math.log(100.0) + (math.log(30000.0) - math.log(100.0)) * torch.rand(60, generator=g))
    scene.scales[:, 0] = long_px * z / fx
    scene.scales[:, 1:] = (0.3 * z / fx)[:, None]
    ang = math.pi * torch.rand(60, generator=g)                        # rotation about the view axis
    scene.rotations = torch.stack([torch.cos(ang / 2), torch.zeros(60), torch.zeros(60), torch.sin(ang / 2)], 1)
    scene.opacities[:] = 0.5
    gc, gd = synth.upstream_grads(H, W)
    bg = torch.zeros(3)
    oo, _ = pa.run_oracle(scene, cam, bg, gc, gd)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
    assert torch.isfinite(hip["color"]).all() and all(torch.isfinite(t).all() for t in hip["grads"].values())
    diff = (hip["color"].double() - oo.color.detach()).abs().amax(0)
    assert float((diff > 2e-3).float().mean()) < 2e-3, float((diff > 2e-3).float().mean())
    # off-axis pixels must stay dark: nothing is blended where the oracle sees (almost) nothing
    dark = oo.color.detach().amax(0) < 1e-3
    assert float(hip["color"].amax(0)[dark].max()) < 1e-2


def test_instance_runs_longer_than_the_staging_pass(gpu):
    """K8a streams a wave's instance records through LDS in passes of 256 records; a Gaussian whose rectangle covers
    more tiles than that has its run split over several passes (and shares passes with its neighbours' short runs).
    672x400 = 42 x 25 = 1 050 tiles; four screen-filling Gaussians (1 050 records each) among ordinary ones, placed so
    that the long runs start at different lanes of their waves (indices 0, 63, 64, 130)."""
    W, H = 672, 400
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(260, cam, seed=21, s_px=(0.5, 5.0))
    for i in (0, 63, 64, 130):
        scene.means3D[i] = torch.tensor([0.1 * (i % 3 - 1), 0.05, 6.0])
        scene.scales[i] = torch.tensor([4.0, 3.0, 3.5])
        scene.opacities[i] = 0.08
    gc, gd = synth.upstream_grads(H, W)
    bg = torch.tensor([0.05, 0.1, 0.15])
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
    assert int(oo.geom.tiles_touched[[0, 63, 64, 130]].min()) == 42 * 25
    _assert_case("long_runs", hip, oo, og, mixed_tol=2.0)      # (screen-filling sums: see test_awkward_inputs)


def test_backward_scratch_needs_no_initialisation(gpu):
    """The backward's instance scratch is handed over uninitialised: K7 writes EVERY instance record -- the sums, or
    zeros for instances that no pixel blended (behind every pixel's last contributor, outside the alpha >= 1/255 box,
    tiles that blended nothing) -- so stale contents (NaN here) cannot leak into a gradient.  A dense opaque scene
    makes most instances of a tile lie behind the last contributor.  An announced backward (K1 stores d(rgb)/d(dir))
    and an unannounced one (recomputed from the coefficients) agree up to rounding."""
    import diff_gaussian_rasterization as dgr
    C_ = dgr._C
    W, H, P = 200, 120, 6000
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=5, s_px=(2.0, 12.0))
    scene.opacities[:] = torch.clamp(scene.opacities * 1.5, max=0.995)
    sc = scene.to(gpu)
    gc, gd = (t.to(gpu) for t in synth.upstream_grads(H, W))
    e_i = torch.empty(0, dtype=torch.int32, device=gpu)
    e_f = torch.empty(0, device=gpu)
    res = []
    for prepare in (True, False, True):
        out = C_.rasterize_gaussians(torch.zeros(3, device=gpu), sc.means3D, None, sc.opacities, sc.scales, sc.rotations,
                                     1.0, None, cam.world_view_transform.to(gpu), cam.full_proj_transform.to(gpu),
                                     cam.tanfovx, cam.tanfovy, H, W, sc.shs, 3, cam.camera_center.to(gpu), False, False,
                                     e_i, e_i, e_f, e_i, True, prepare_backward=prepare)
        color, invd, call = out[1], out[6], out[7]
        torch.cuda.synchronize()
        poison = torch.full((256 << 20,), float("nan"), device=gpu)   # the caching allocator hands out NaN-filled blocks
        del poison
        g = C_.rasterize_gaussians_backward(call, color, invd, gc, gd)
        res.append([t.clone() for t in g if t is not None])
        assert all(torch.isfinite(t).all() for t in res[-1])
    views = C_.raster_views(call)
    nc_max = int(views["n_contrib"].max())
    longest = int((views["ranges"][:, 1] - views["ranges"][:, 0]).max())
    assert nc_max < longest, "the case must have instances behind the last contributor"
    for a, b in zip(res[0], res[1]):
        assert torch.allclose(a, b, rtol=0, atol=2e-6 * float(a.abs().max()))
    for a, b in zip(res[0], res[2]):
        assert torch.equal(a, b)


def test_huge_tile_grid_takes_the_radix_path(gpu):
    """Tile grids above 32 768 tiles (here 320 x 180 = 57 600 at 5120 x 2880) do not fit the LDS histogram of the
    counting tile-binning and fall back to the stable radix sort by tile id + ranges kernel.  Forward only;
    oracle = geometry + binning spec (bit-exact ranges, point list, sorted tile ids)."""
    import diff_gaussian_rasterization as dgr
    from oracle import raster_oracle as ro
    W, H, P = 5120, 2880, 60_000
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=21, s_px=(1.0, 12.0))
    geom = ro.geometry_spec(scene.means3D.numpy(), scene.scales.numpy(), scene.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            cam.tanfovx, cam.tanfovy, 1.0)
    binning = ro.binning_spec(geom)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, device=gpu))
    sc = scene.to(gpu)
    for _ in range(2):        # two-stage call first, speculative single call second
        L, color, radii, geomb, binb, img, invd, call = dgr._C.rasterize_gaussians(
            rs.bg, sc.means3D, None, sc.opacities, sc.scales, sc.rotations, 1.0, None, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, H, W, sc.shs, 3, rs.campos, False, False, rs.render_indices, rs.parent_indices,
            rs.interpolation_weights, rs.num_node_kids, True)
        v = dgr._C.raster_views(call)
        assert L == binning.num_rendered
        assert np.array_equal(radii.cpu().numpy(), geom.radii)
        assert np.array_equal(v["ranges"].cpu().numpy(), binning.ranges)
        assert np.array_equal(v["point_list"].cpu().numpy(), binning.point_list)
        assert np.array_equal(v["tile_ids_sorted"].cpu().numpy().astype(np.uint64),
                              binning.keys_sorted >> np.uint64(32))
        assert torch.isfinite(color).all() and float(color.max()) > 0.05


def test_deferred_batched_sh_backward(gpu):
    """Gradient accumulation over several views with the SH part of the backward deferred and done in ONE pass
    (hgs_raster_sh_bwd_batched) must equal the sum of the views' separately computed gradients -- for every input,
    since the view-direction term of dL_dmeans3D moves into the batched pass as well."""
    import diff_gaussian_rasterization as dgr
    from hgs import dp
    W, H, P, K = 176, 112, 1500, 3
    base = synth.make_camera(W, H)
    scene = synth.make_scene(P, base, seed=7)
    cams = [synth.orbit_camera(W, H, j, K, radius=0.4) for j in range(K)]
    gc, gd = synth.upstream_grads(H, W)
    bg = torch.zeros(3)
    sep = [pa.run_hip(scene, c, bg, gc, gd, gpu)["grads"] for c in cams]
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    sc = scene.to(gpu)
    params = {n: getattr(sc, n).clone().requires_grad_(True) for n in names}
    bucket = dp.GradBucket({n: tuple(v.shape) for n, v in params.items()}, gpu)
    bucket.flat.fill_(float("nan"))               # stale contents must not survive
    rc = dgr.RasterContext(grad_buffers=bucket.views, defer_sh_backward=True)
    for j, c in enumerate(cams):
        rc.grad_accumulate = j > 0
        rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(c, bg, 3, device=gpu))
        m2 = torch.zeros(P, 3, device=gpu, requires_grad=True)
        color, radii, invd = dgr.GaussianRasterizer(rs, context=rc)(
            means3D=params["means3D"], means2D=m2, shs=params["shs"], opacities=params["opacities"],
            scales=params["scales"], rotations=params["rotations"])
        torch.autograd.backward([color, invd], [gc.to(gpu), gd.to(gpu)])
    assert len(rc.pending_sh) == K
    rc.finish_deferred_sh_backward()
    assert len(rc.pending_sh) == 0
    for n in names:
        want = sum(s[n] for s in sep)
        got = bucket.views[n].cpu()
        assert torch.isfinite(got).all(), n
        assert torch.allclose(got, want, rtol=1e-5, atol=2e-6 * float(want.abs().max())), \
            (n, float((got - want).abs().max()), float(want.abs().max()))


def test_batched_sh_colors_route(gpu):
    """Colours of several views in one pass (hgs_sh_colors_batched), rasterization with colors_precomp, one pass for
    dL/dSH (hgs_sh_colors_batched_bwd) -- the batched form of the reference's convert_SHs_python route -- must give
    the colours of the in-op SH evaluation and the summed gradients of the views' standard backward."""
    import diff_gaussian_rasterization as dgr
    from oracle import raster_oracle as ro
    W, H, P, K = 176, 112, 1500, 3
    base = synth.make_camera(W, H)
    scene = synth.make_scene(P, base, seed=17)
    cams = [synth.orbit_camera(W, H, j, K, radius=0.4) for j in range(K)]
    gc, gd = synth.upstream_grads(H, W)
    bg = torch.zeros(3)
    sep = [pa.run_hip(scene, c, bg, gc, gd, gpu) for c in cams]
    sc = scene.to(gpu)
    campos = [c.camera_center.to(gpu) for c in cams]
    rgbs, clamps = dgr.sh_colors_batched(sc.means3D, sc.shs, 3, campos)
    for j, c in enumerate(cams):        # forward: against the torch restatement of utils/sh_utils.eval_sh
        d = scene.means3D - c.camera_center
        d = d / d.norm(dim=1, keepdim=True)
        ref = torch.clamp_min(ro.eval_sh_torch(3, scene.shs.double(), d.double()) + 0.5, 0.0)
        assert float((rgbs[j].cpu().double() - ref).abs().max()) <= 2e-6
        assert bool(((clamps[j].cpu() != 0) <= (rgbs[j].cpu() == 0).any(dim=1)).all())     # clamped => a zero channel
    names = ("means3D", "opacities", "scales", "rotations")
    params = {n: getattr(sc, n).clone().requires_grad_(True) for n in names}
    tot = {n: torch.zeros_like(params[n]) for n in names}
    d_rgbs = []
    for j, c in enumerate(cams):
        rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(c, bg, 3, device=gpu))
        m2 = torch.zeros(P, 3, device=gpu, requires_grad=True)
        rgb = rgbs[j].requires_grad_(True)
        color, radii, invd = dgr.GaussianRasterizer(rs)(
            means3D=params["means3D"], means2D=m2, colors_precomp=rgb, opacities=params["opacities"],
            scales=params["scales"], rotations=params["rotations"])
        assert float((color.detach().cpu() - sep[j]["color"]).abs().max()) <= 1e-5
        g = torch.autograd.grad([color, invd], [params[n] for n in names] + [rgb], [gc.to(gpu), gd.to(gpu)])
        for n, t in zip(names, g):
            tot[n] += t
        d_rgbs.append(g[-1])
    d_shs = torch.full_like(sc.shs, float("nan"))
    dgr.sh_colors_batched_backward(sc.means3D, sc.shs, 3, campos, clamps, d_rgbs, d_shs, tot["means3D"])
    tot["shs"] = d_shs
    for n in ("means3D", "shs", "opacities", "scales", "rotations"):
        want = sum(s["grads"][n] for s in sep)
        got = tot[n].cpu()
        assert torch.isfinite(got).all(), n
        assert torch.allclose(got, want, rtol=1e-5, atol=2e-6 * float(want.abs().max())), \
            (n, float((got - want).abs().max()), float(want.abs().max()))


def test_backward_on_second_stream_gives_identical_gradients(gpu):
    """RasterContext.backward_stream: the backwards of several views run on a second HIP stream next to the
    following forwards.  Same kernels, same inputs: the accumulated gradients must be bit-identical to the
    single-stream schedule, repeatedly (a missing dependency or a recycled workspace would show up as a mismatch)."""
    import diff_gaussian_rasterization as dgr
    from hgs import dp
    W, H, P, K = 480, 270, 60_000, 4
    base = synth.make_camera(W, H)
    sc = synth.make_scene(P, base, seed=4).to(gpu)
    cams = [synth.orbit_camera(W, H, j, K).to(gpu) for j in range(K)]
    gc, gd = (t.to(gpu) for t in synth.upstream_grads(H, W))
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    params = {n: getattr(sc, n).clone().requires_grad_(True) for n in names}

    def run(stream):
        shapes = {n: tuple(v.shape) for n, v in params.items()}
        bucket = dp.GradBucket(shapes, gpu)
        bucket.flat.fill_(float("nan"))
        m2_bufs = [torch.full((P, 3), float("nan"), device=gpu) for _ in cams]
        rc = dgr.RasterContext(backward_stream=stream)
        for j, c in enumerate(cams):
            rc.grad_buffers = dict(bucket.views, means2D=m2_bufs[j])
            rc.grad_accumulate = j > 0
            rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(c, torch.zeros(3), 3, device=gpu))
            m2 = torch.zeros(P, 3, device=gpu, requires_grad=True)
            color, radii, invd = dgr.GaussianRasterizer(rs, context=rc)(
                means3D=params["means3D"], means2D=m2, shs=params["shs"], opacities=params["opacities"],
                scales=params["scales"], rotations=params["rotations"])
            torch.autograd.backward([color, invd], [gc, gd])
            assert m2.grad is None
        rc.wait_backward_stream()
        torch.cuda.synchronize()
        return bucket.flat.clone(), [m.clone() for m in m2_bufs]

    ref, ref_m2 = run(None)
    assert torch.isfinite(ref).all()
    sb = torch.cuda.Stream(device=gpu)
    for _ in range(3):
        got, got_m2 = run(sb)
        assert torch.equal(got, ref)
        assert all(torch.equal(a, b) for a, b in zip(got_m2, ref_m2))


def test_backward_stream_with_autograd_consumers_is_safe(gpu):
    """Gradients RETURNED to autograd from a backward that ran on the side stream are consumed by nodes on the
    forward's stream (here: the backward of the reference's activations, scene/gaussian_model.py:108-128, and
    AccumulateGrad over two views).  The op makes the forward's stream wait for them; the result must equal the
    single-stream one bit for bit."""
    import diff_gaussian_rasterization as dgr
    W, H, P = 480, 270, 60_000
    base = synth.make_camera(W, H)
    sc = synth.make_scene(P, base, seed=14).to(gpu)
    cams = [synth.orbit_camera(W, H, j, 2).to(gpu) for j in range(2)]
    gc, gd = (t.to(gpu) for t in synth.upstream_grads(H, W))

    def run(stream):
        raw = dict(xyz=sc.means3D.clone(), sh=sc.shs.clone(), op=torch.logit(sc.opacities.clamp(0.01, 0.99)),
                   sc=sc.scales.log(), rot=sc.rotations * 1.7)
        for t in raw.values():
            t.requires_grad_(True)
        rc = dgr.RasterContext(backward_stream=stream)
        for c in cams:
            rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(c, torch.zeros(3), 3, device=gpu))
            m2 = torch.zeros(P, 3, device=gpu, requires_grad=True)
            color, radii, invd = dgr.GaussianRasterizer(rs, context=rc)(
                means3D=raw["xyz"], means2D=m2, shs=raw["sh"], opacities=torch.sigmoid(raw["op"]),
                scales=torch.exp(raw["sc"]), rotations=torch.nn.functional.normalize(raw["rot"]))
            ((color * gc).sum() + (invd * gd).sum()).backward()
        torch.cuda.synchronize()
        return {k: v.grad.clone() for k, v in raw.items()}

    ref = run(None)
    sb = torch.cuda.Stream(device=gpu)
    for _ in range(3):
        got = run(sb)
        for k in ref:
            assert torch.isfinite(got[k]).all() and torch.equal(got[k], ref[k]), k


def test_two_contexts_and_two_resolutions_share_nothing(gpu):
    """A viewer-style render at another resolution in the middle of a training step (train_single.py:76-78), through
    a second rasterizer without a context: the training rasterizer's buffers, pending views and speculative workspace
    size are untouched."""
    import diff_gaussian_rasterization as dgr
    from hgs import dp
    P = 5000
    cam_a, cam_b = synth.make_camera(256, 144), synth.make_camera(96, 64)
    scene = synth.make_scene(P, cam_a, seed=23)
    sc = scene.to(gpu)
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    params = {n: getattr(sc, n).clone().requires_grad_(True) for n in names}
    bucket = dp.GradBucket({n: tuple(v.shape) for n, v in params.items()}, gpu)
    rc = dgr.RasterContext(grad_buffers=bucket.views)
    gc, gd = (t.to(gpu) for t in synth.upstream_grads(144, 256))
    rs_a = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam_a, torch.zeros(3), 3, device=gpu))
    rs_b = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam_b, torch.zeros(3), 3, device=gpu))
    kw = dict(means3D=params["means3D"], shs=params["shs"], opacities=params["opacities"], scales=params["scales"],
              rotations=params["rotations"])

    def train_view():
        m2 = torch.zeros(P, 3, device=gpu, requires_grad=True)
        color, _, invd = dgr.GaussianRasterizer(rs_a, context=rc)(means2D=m2, **kw)
        return color, invd

    color, invd = train_view()
    torch.autograd.backward([color, invd], [gc, gd])
    ref = bucket.flat.clone()
    color, invd = train_view()
    with torch.no_grad():                                    # the interleaved small render
        small, _, _ = dgr.GaussianRasterizer(rs_b)(means2D=torch.zeros(P, 3, device=gpu), **kw)
    m2 = torch.zeros(P, 3, device=gpu, requires_grad=True)   # ... and a small differentiable one, plain autograd
    c2, _, i2 = dgr.GaussianRasterizer(rs_b)(means2D=m2, **kw)
    c2.sum().backward()
    plain = {n: params[n].grad.clone() for n in names}
    torch.autograd.backward([color, invd], [gc, gd])
    assert torch.equal(bucket.flat, ref)
    assert all(torch.equal(params[n].grad, plain[n]) for n in names)
    keys = [k for k in dgr._C._last_L if k[3] == P]
    assert {k[1:3] for k in keys} >= {(256, 144), (96, 64)}


def test_noncontiguous_inputs_and_inplace_update_detection(gpu):
    """Like the upstream extension the op accepts sliced / expanded inputs (it makes them contiguous itself), e.g. an
    ``override_color`` broadcast (gaussian_renderer/__init__.py:90-91).  And because the inputs are saved through
    autograd, an in-place parameter update between forward and backward raises instead of mixing two states."""
    import diff_gaussian_rasterization as dgr
    cam, scene, gc, gd = pa.default_case(900, 128, 96, seed=51)
    bg = torch.zeros(3)
    sc = scene.to(gpu)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, bg, 3, device=gpu))
    col = torch.tensor([0.3, 0.5, 0.7], device=gpu).expand(scene.P, 3)          # stride (0, 1)
    wide = torch.cat([sc.scales, sc.scales], 1)[:, :3]                            # row stride 6
    a = dgr.GaussianRasterizer(rs)(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), colors_precomp=col,
                                   opacities=sc.opacities, scales=wide, rotations=sc.rotations)
    b = dgr.GaussianRasterizer(rs)(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D),
                                   colors_precomp=col.contiguous(), opacities=sc.opacities, scales=sc.scales,
                                   rotations=sc.rotations)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    op = sc.opacities.clone().requires_grad_(True)
    color, _, _ = dgr.GaussianRasterizer(rs)(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), shs=sc.shs,
                                             opacities=op, scales=sc.scales, rotations=sc.rotations)
    with torch.no_grad():
        op.mul_(0.5)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        color.sum().backward()


def test_large_footprints_take_the_long_run_route(gpu):
    """Gaussians that cover tens to hundreds of tiles (a trained scene, a coarse hierarchy cut): 60 instance records per
    Gaussian on average, runs of several hundred.  The backward sums such runs with a wave per run in front of K8a
    (k8_presum_long_kernel, preprocess.hip; the frame's mean run decides) -- pixels, indices and every gradient against
    the oracle."""
    W, H, P = 320, 192, 1200
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=29, s_px=(6.0, 40.0))
    scene.opacities = scene.opacities * 0.35                  # keep the pixels from saturating behind a few layers
    gc, gd = synth.upstream_grads(H, W, seed=5)
    bg = torch.tensor([0.1, 0.0, 0.2])
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
    idx = pa.check_indices(hip, oo)
    assert all(v == 0 for v in idx.values()), idx
    tt = oo.geom.tiles_touched
    assert hip["L"] > 6 * P and int(tt.max()) > 48 * 2, (hip["L"], int(tt.max()))
    assert int(((tt > 0) & (tt <= 48)).sum()) > 0             # short runs next to long ones
    pa.assert_stats("large footprints", pa.compare(hip, oo, og))


def test_batches_with_more_pairs_than_accumulator_slots(gpu):
    """K7 keeps the ten sums of every (instance, quadrant) pair of a 64-instance batch in a slot of its own (128 slots,
    render.hip): Gaussians that reach all four quadrants of a tile make up to 256 pairs per batch, the batch then keeps
    its back-most instances and the next one starts at the first instance left out.  Tiles whose lists are several
    hundred wide Gaussians (every batch cut), mixed with small ones (cuts at varying lanes, batches that just fit), low
    opacities so that pixels blend hundreds of entries -- pixels, indices and every gradient against the oracle."""
    W, H = 160, 96
    cam = synth.make_camera(W, H)
    big = synth.make_scene(500, cam, seed=41, s_px=(30.0, 90.0))
    small = synth.make_scene(2500, cam, seed=42, s_px=(0.5, 5.0))
    scene = synth.Scene(torch.cat([big.means3D, small.means3D]), torch.cat([big.scales, small.scales]),
                        torch.cat([big.rotations, small.rotations]), torch.cat([big.opacities * 0.05, small.opacities * 0.3]),
                        torch.cat([big.shs, small.shs]), 3)
    perm = torch.randperm(scene.P, generator=torch.Generator().manual_seed(3))
    scene = synth.Scene(scene.means3D[perm], scene.scales[perm], scene.rotations[perm], scene.opacities[perm], scene.shs[perm], 3)
    gc, gd = synth.upstream_grads(H, W, seed=9)
    bg = torch.tensor([0.2, 0.1, 0.0])
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
    per_tile = (oo.binning.ranges[:, 1].astype(np.int64) - oo.binning.ranges[:, 0].astype(np.int64))
    assert int(per_tile.min()) > 300 and int(oo.n_contrib.max()) > 256, (int(per_tile.min()), int(oo.n_contrib.max()))
    idx = pa.check_indices(hip, oo)
    assert all(v == 0 for v in idx.values()), idx
    st = pa.compare(hip, oo, og)
    _log("more pairs than slots", {"indices": idx, "stats": st})
    # (every entry of a list has its own small chance of a knife-edge decision, and the band grows with the footprint:
    # 4e-6 per entry in the at-scale suite, 6e-6 here where every fifth entry is 30 - 90 px wide; measured 2.5e-3)
    assert st["fragile_frac"] <= max(pa.FRAGILE_FRAC, 6e-6 * float(per_tile.mean())), st["fragile_frac"]
    # element-wise bound x 2, as for the other case of screen-filling Gaussians (float32 sums over thousands of pixels)
    pa.assert_stats("more pairs than slots", st, mixed_tol=2.0)


def test_forwards_on_two_streams_keep_their_own_superblock_totals(gpu):
    """K1 adds its workgroup sums to zeroed superblock totals the library keeps PER (device, stream) (preprocess.hip): two
    streams rendering different scenes in turns -- the speculative single-call path on both -- must each see only their own
    sums: sorted lists and ranges of every frame equal to the first frame of its scene (checked against the oracle)."""
    import diff_gaussian_rasterization as dgr
    from oracle import raster_oracle as ro
    W, H = 400, 240
    cam = synth.make_camera(W, H)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, device=gpu))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    scenes, refs = [], []
    for k, P in enumerate((9000, 14000)):
        sc = synth.make_scene(P, cam, seed=60 + k)
        geom = ro.geometry_spec(sc.means3D.numpy(), sc.scales.numpy(), sc.rotations.numpy(), None,
                                cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                                cam.tanfovx, cam.tanfovy, 1.0)
        b = ro.binning_spec(geom)
        scenes.append(sc.to(gpu))
        refs.append((int(b.num_rendered), torch.from_numpy(b.ranges.astype(np.int64)).to(gpu),
                     torch.from_numpy(b.point_list.astype(np.int64)).to(gpu)))
    torch.cuda.synchronize()
    bad = 0
    for it in range(40):
        k = it % 2 if it % 5 else (it // 5) % 2              # mostly alternating, sometimes twice on one stream
        sc = scenes[k]
        with torch.cuda.stream(streams[k]):
            L, color, radii, _, _, _, invd, call = dgr._C.rasterize_gaussians(
                rs.bg, sc.means3D, None, sc.opacities, sc.scales, sc.rotations, 1.0, None, rs.viewmatrix, rs.projmatrix,
                rs.tanfovx, rs.tanfovy, H, W, sc.shs, 3, rs.campos, False, False, rs.render_indices, rs.parent_indices,
                rs.interpolation_weights, rs.num_node_kids, False)
            v = dgr._C.raster_views(call)
            assert L == refs[k][0], (it, k, L)
            bad += int((v["ranges"].to(torch.int64) != refs[k][1]).sum()) + int((v["point_list"].to(torch.int64) != refs[k][2]).sum())
    torch.cuda.synchronize()
    assert bad == 0


_CLUSTERED_BIG = r"""
import os, sys
root, out = sys.argv[1], sys.argv[2]
for p in (root, os.path.join(root, "hierarchical-3d-gaussians_amd"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import parity as pa
from hgs import synth
import diff_gaussian_rasterization as dgr
W, H = 640, 384
cam = synth.make_camera(W, H)
sc = synth.make_scene(12000, cam, seed=77)
sc.scales[:300] *= 40.0                  # the first 300 rows cover hundreds of tiles each: workgroups 0 and 1 of K1 / K3
sc = sc.to("cuda:0")
rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, device="cuda:0"))
for it in range(3):                      # the first call sizes the workspaces; the later ones take the single-call path
    L, color, radii, _, _, _, invd, call = dgr._C.rasterize_gaussians(
        rs.bg, sc.means3D, None, sc.opacities, sc.scales, sc.rotations, 1.0, None, rs.viewmatrix, rs.projmatrix,
        rs.tanfovx, rs.tanfovy, H, W, sc.shs, 3, rs.campos, False, False, rs.render_indices, rs.parent_indices,
        rs.interpolation_weights, rs.num_node_kids, False)
    v = dgr._C.raster_views(call)
torch.cuda.synchronize()
np.savez(out, L=np.int64(L), ranges=v["ranges"].cpu().numpy(), point_list=v["point_list"].cpu().numpy(),
         speculative=np.int64(dgr._C.stats["speculative_calls"]))
"""


def test_clustered_big_gaussians_are_shared_out(gpu, tmp_path):
    """A hierarchy cut lists its big nodes side by side: one workgroup of K3 then has a hundred times the mean emission.
    K3 shares the excess of such blocks out over all workgroups (binning.hip, the rare path); HGS_K3_SHARE=1 lists one
    heavy block only (the second emits all of its own), HGS_K3_SHARE=0 turns the sharing off.  Ranges and sorted lists of
    the single-call path against the oracle's binning, each setting in a process of its own (read once)."""
    import subprocess
    from oracle import raster_oracle as ro
    W, H = 640, 384
    cam = synth.make_camera(W, H)
    sc = synth.make_scene(12000, cam, seed=77)
    sc.scales[:300] *= 40.0
    geom = ro.geometry_spec(sc.means3D.numpy(), sc.scales.numpy(), sc.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            cam.tanfovx, cam.tanfovy, 1.0)
    b = ro.binning_spec(geom)
    tt = geom.tiles_touched.astype(np.int64)
    blocks = np.add.reduceat(tt, np.arange(0, len(tt), 256))
    # the library's threshold (binning.hip, k3_heavy_threshold): twice the mean emission the frame's CAPACITY allows --
    # the speculative capacity is 1.25 L + 64 Ki rounded up by at most 1/16
    cap = (int(tt.sum()) * 1.25 + 65536) * (1 + 1 / 16)
    thr = max(4096, 2 * -(-int(cap) // len(blocks)))
    assert int((blocks > thr).sum()) >= 2 and blocks[0] > 20 * np.median(blocks), (blocks[:3], thr)   # the case is the case
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, env in (("default", {}), ("one", {"HGS_K3_SHARE": "1"}), ("off", {"HGS_K3_SHARE": "0"})):
        out = str(tmp_path / f"{name}.npz")
        r = subprocess.run([sys.executable, "-c", _CLUSTERED_BIG, root, out], env={**os.environ, **env},
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        d = np.load(out)
        assert int(d["speculative"]) >= 2, name
        assert int(d["L"]) == int(b.num_rendered), (name, int(d["L"]), int(b.num_rendered))
        assert np.array_equal(d["ranges"].astype(np.int64).reshape(-1), b.ranges.astype(np.int64).reshape(-1)), name
        assert np.array_equal(d["point_list"].astype(np.int64)[:int(d["L"])], b.point_list.astype(np.int64)), name

