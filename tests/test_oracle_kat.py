"""Hand-derivable known answers for the oracle + internal consistency:
vectorised blend == literal per-pixel loop; float32 geometry spec == float64 torch geometry."""
import math

import numpy as np
import pytest
import torch

from hgs import synth
from oracle import raster_oracle as ro


def _render(scene, cam, bg=None, **kw):
    bg = torch.zeros(3) if bg is None else bg
    return ro.rasterize(scene.means3D, None, scene.shs, None, scene.opacities, scene.scales, scene.rotations, None,
                        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx,
                        tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                        projmatrix=cam.full_proj_transform, sh_degree=scene.sh_degree, campos=cam.camera_center, **kw)


def _one(z=4.0, s=0.05, o=0.8, dc=(1.0, 0.0, -1.0), xy=(0.0, 0.0)):
    sc = synth.Scene(torch.tensor([[xy[0], xy[1], z]]), torch.full((1, 3), s), torch.tensor([[1.0, 0, 0, 0]]),
                     torch.tensor([[o]]), torch.zeros(1, 16, 3), 3)
    sc.shs[0, 0] = torch.tensor(dc)
    return sc


def test_single_isotropic_gaussian_on_axis():
    cam = synth.make_camera(64, 48)
    out = _render(_one(), cam)
    fy = 48 / (2 * cam.tanfovy)
    var = (fy * 0.05 / 4.0) ** 2 + 0.3
    cx, cy = 31.5, 23.5
    ys, xs = torch.meshgrid(torch.arange(48.0), torch.arange(64.0), indexing="ij")
    alpha = torch.clamp(0.8 * torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * var)), max=0.99)
    alpha = torch.where(alpha < 1 / 255.0, torch.zeros_like(alpha), alpha).double()
    rad = math.ceil(3 * math.sqrt(var))
    assert int(out.radii[0]) == rad
    mask = torch.zeros(48, 64, dtype=torch.bool)
    mask[16:32, 16:48] = True                                     # the Gaussian's 3-sigma tile rectangle
    expect = torch.where(mask, alpha, torch.zeros_like(alpha))
    c0 = 0.5 + 0.28209479177387814
    assert (out.color[0] - c0 * expect).abs().max() < 1e-7
    assert (out.color[1] - 0.5 * expect).abs().max() < 1e-7
    assert (out.color[2] - (0.5 - 0.28209479177387814) * expect).abs().max() < 1e-7
    assert (out.invdepth[0] - expect / 4.0).abs().max() < 1e-7
    assert abs(out.final_T[24, 32] - (1 - float(expect[24, 32]))) < 1e-7


def test_two_gaussians_depth_order_independent_of_input_order():
    cam = synth.make_camera(48, 48)
    a, b = _one(z=3.0, o=0.6, dc=(1, 1, 1)), _one(z=6.0, s=0.2, o=0.9, dc=(-1, 0, 1))
    cat = lambda f, g: torch.cat([f, g])
    def scene(x, y):
        return synth.Scene(cat(x.means3D, y.means3D), cat(x.scales, y.scales), cat(x.rotations, y.rotations),
                           cat(x.opacities, y.opacities), cat(x.shs, y.shs), 3)
    o1, o2 = _render(scene(a, b), cam), _render(scene(b, a), cam)
    assert torch.equal(o1.color, o2.color)
    # centre pixel: C = c1 a1 + c2 a2 (1 - a1)
    oa, ob = _render(a, cam), _render(b, cam)
    y, x = 24, 24
    a1 = 1 - oa.final_T[y, x]
    assert abs(float(o1.color[0, y, x]) - float(oa.color[0, y, x] + ob.color[0, y, x] * (1 - a1))) < 1e-9


def test_near_plane_cull_and_background_passthrough():
    cam = synth.make_camera(32, 32)
    bg = torch.tensor([0.2, 0.4, 0.6])
    out = _render(_one(z=0.2), cam, bg)                           # z <= 0.2 is culled
    assert int(out.radii[0]) == 0 and out.binning.num_rendered == 0
    assert torch.allclose(out.color, bg.double()[:, None, None].expand(3, 32, 32))
    assert float(out.invdepth.abs().max()) == 0.0
    out = _render(_one(z=0.2001), cam, bg)
    assert int(out.radii[0]) > 0


def test_vectorised_blend_equals_literal_per_pixel_loop():
    cam = synth.make_camera(40, 24)
    scene = synth.make_scene(60, cam, seed=5, s_px=(1.0, 6.0))
    bg = torch.tensor([0.1, 0.0, 0.3])
    out = _render(scene, cam, bg)
    geom, binning = out.geom, out.binning
    # continuous per-Gaussian values in float64, computed independently of rasterize()
    p = scene.means3D.double()
    ph = torch.cat([p, torch.ones(scene.P, 1, dtype=torch.float64)], 1) @ cam.full_proj_transform.double()
    ndc = ph[:, :2] / (ph[:, 3:4] + 1e-7)
    gx = ((ndc[:, 0] + 1) * 40 - 1) * 0.5
    gy = ((ndc[:, 1] + 1) * 24 - 1) * 0.5
    d = p - cam.camera_center.double()
    d = d / d.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(ro.eval_sh_torch(3, scene.shs.double(), d) + 0.5, 0)
    tz = (torch.cat([p, torch.ones(scene.P, 1, dtype=torch.float64)], 1) @ cam.world_view_transform.double())[:, 2]
    ref, dep = ro.naive_per_pixel_blend(gx.numpy(), gy.numpy(), geom.conic.astype(np.float64),
                                        scene.opacities.double().reshape(-1).numpy(), rgb.numpy(), (1 / tz).numpy(),
                                        binning.point_list, binning.ranges, 40, 24, bg.double().numpy())
    ok = ~out.fragile
    assert np.abs(out.color.numpy() - ref)[:, ok].max() < 5e-6      # conic: float32 spec vs float64 torch
    assert np.abs(out.invdepth.numpy()[0] - dep)[ok].max() < 5e-6


def test_binning_spec_invariants():
    cam = synth.make_camera(200, 120)
    scene = synth.make_scene(3000, cam, seed=2)
    out = _render(scene, cam, tiles=[])
    g, b = out.geom, out.binning
    assert b.num_rendered == int(g.tiles_touched.sum())
    assert np.all(b.keys_sorted[1:] >= b.keys_sorted[:-1])
    tiles = (b.keys_sorted >> np.uint64(32)).astype(np.int64)
    cnt = np.bincount(tiles, minlength=b.ranges.shape[0])
    assert np.array_equal(b.ranges[:, 1] - b.ranges[:, 0], cnt)
    # stability: equal keys keep ascending Gaussian order
    same = b.keys_sorted[1:] == b.keys_sorted[:-1]
    assert np.all(b.point_list[1:][same] > b.point_list[:-1][same])
    # every instance's tile lies inside its Gaussian's rectangle
    gx = g.grid[0]
    tx, ty = tiles % gx, tiles // gx
    pl = b.point_list
    assert np.all((tx >= g.rect_min[pl, 0]) & (tx < g.rect_max[pl, 0]) & (ty >= g.rect_min[pl, 1]) & (ty < g.rect_max[pl, 1]))


def test_oracle_gradients_match_finite_differences():
    cam = synth.make_camera(32, 32)
    scene = synth.make_scene(12, cam, seed=9, s_px=(2.0, 6.0))
    gc, gd = synth.upstream_grads(32, 32)
    def loss_of(m3, op):
        o = ro.rasterize(m3, None, scene.shs, None, op, scene.scales, scene.rotations, None, image_height=32,
                         image_width=32, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3),
                         scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                         sh_degree=3, campos=cam.camera_center)
        return (o.color * gc.double()).sum() + (o.invdepth * gd.double()).sum()
    m3 = scene.means3D.double().clone().requires_grad_(True)
    op = scene.opacities.double().clone().requires_grad_(True)
    loss_of(m3, op).backward()
    eps = 1e-6
    for (i, j) in [(0, 0), (3, 1), (7, 2)]:
        d = torch.zeros_like(m3); d[i, j] = eps
        fd = (loss_of((m3 + d).detach(), op.detach()) - loss_of((m3 - d).detach(), op.detach())) / (2 * eps)
        assert abs(float(fd) - float(m3.grad[i, j])) <= 1e-4 * max(1.0, abs(float(fd))), (i, j, float(fd), float(m3.grad[i, j]))
    d = torch.zeros_like(op); d[5, 0] = eps
    fd = (loss_of(m3.detach(), (op + d).detach()) - loss_of(m3.detach(), (op - d).detach())) / (2 * eps)
    assert abs(float(fd) - float(op.grad[5, 0])) <= 1e-4 * max(1.0, abs(float(fd)))


def test_lod_opacity_identity_and_stacking():
    o = torch.tensor([0.1, 0.5, 0.9, 1.4], dtype=torch.float64)
    w = torch.tensor([0.0, 0.3, 1.0, 0.5], dtype=torch.float64)
    kids1 = torch.tensor([1, 1, 1, 1], dtype=torch.int32)
    assert torch.equal(ro.lod_opacity(o, w, kids1), o)
    kids = torch.tensor([3, 3, 3, 3], dtype=torch.int32)
    r = ro.lod_opacity(o, torch.zeros(4, dtype=torch.float64), kids)
    assert torch.allclose(1 - (1 - r[:3]) ** 3, o[:3])            # k stacked copies composite like the parent
    assert torch.allclose(ro.lod_opacity(o, torch.ones(4, dtype=torch.float64), kids), o)


def lod_parent_vs_children(render, k, o, w=0.0):
    """One parent Gaussian of opacity ``o`` against its ``k`` coincident children drawn with interpolation weight ``w``
    and ``num_node_kids = k`` (at w = 0 a child "looks like its parent", gaussian_renderer/__init__.py:204-218).
    ``render(scene, cam, weights, kids) -> color [3,H,W]``.  Returns (max |children - parent| over the pixels, the
    parent's image).  Shared by the oracle KAT below and the device KAT in tests/test_lod_gpu.py."""
    cam = synth.make_camera(64, 64)
    z, s = 4.0, 0.45                                  # sigma' = sqrt((fy s / z)^2 + 0.3) = 6.3 px
    sh = torch.zeros(1, 16, 3, dtype=torch.float32)
    sh[:, 0] = (1.0 - 0.5) / ro.SH_C0                 # DC colour 1: the image IS the accumulated alpha
    one = synth.Scene(torch.tensor([[0.0, 0.0, z]]), torch.full((1, 3), s), torch.tensor([[1.0, 0.0, 0.0, 0.0]]),
                      torch.tensor([[o]], dtype=torch.float32), sh, 3)
    rep = lambda t: t.repeat((k,) + (1,) * (t.dim() - 1)).contiguous()
    kids_scene = synth.Scene(rep(one.means3D), rep(one.scales), rep(one.rotations), rep(one.opacities), rep(one.shs), 3)
    parent = render(one, cam, None, None)
    children = render(kids_scene, cam, torch.full((k,), w, dtype=torch.float32), torch.full((k,), k, dtype=torch.int32))
    return float((children - parent).abs().max()), parent


def _oracle_lod_render(mode):
    def render(scene, cam, weights, kids):
        with torch.no_grad():
            return ro.rasterize(scene.means3D, None, scene.shs, None, scene.opacities, scene.scales, scene.rotations,
                                None, image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx,
                                tanfovy=cam.tanfovy, bg=torch.zeros(3), scale_modifier=1.0,
                                viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3,
                                campos=cam.camera_center, interpolation_weights=weights, num_node_kids=kids,
                                lod_mode=mode).color
    return render


def lod_remap_expected(k, o):
    """sup over the falloff G in (0, 1] of |1 - (1 - o' G)^k - min(0.99, o G)| with o' = lod_opacity(o, w=0, k): what the
    per-Gaussian remap costs against its own rationale (zero only at G = 1), alpha caps included, the 1/255 skip not."""
    G = np.linspace(1e-4, 1.0, 20001)
    op = 1.0 - (1.0 - min(o, 0.99)) ** (1.0 / k)
    return float(np.abs(1.0 - (1.0 - np.minimum(0.99, op * G)) ** k - np.minimum(0.99, o * G)).max())


@pytest.mark.parametrize("k", [2, 4, 8])
@pytest.mark.parametrize("o", [0.3, 0.9, 1.3])
def test_lod_remap_parent_vs_children(k, o):
    """does the hierarchy-mode remap do what its rationale says -- k coincident children at w = 0
    composite like their parent?  Per-Gaussian remap of o (``lod_opacity``, what the kernels implement): only at the
    centre; the worst pixel is off by what ``lod_remap_expected`` predicts (up to 0.24 in alpha for o = 1.3, k = 8).
    Per-pixel remap of alpha (``lod_alpha``, oracle only): exact wherever the children pass the alpha >= 1/255 test
    (each child carries ~alpha / k, so the parent's alpha below ~k / 255 is lost to the skip rule in BOTH modes: the
    floor of this comparison).  Numbers: profiles/r04_lod_remap_kat.txt."""
    e_op, parent = lod_parent_vs_children(_oracle_lod_render("opacity"), k, o)
    e_al, _ = lod_parent_vs_children(_oracle_lod_render("alpha"), k, o)
    centre = float(parent[0, 32, 32])
    assert abs(centre - min(0.99, o * np.exp(-0.5 * (2 * 0.5 ** 2) / ((0.5 * 64 / np.tan(np.pi / 6) * 0.45 / 4.0) ** 2 + 0.3)))) < 1e-6
    exp = lod_remap_expected(k, o)
    floor = k / 255.0
    assert e_al <= floor * 1.01
    assert abs(e_op - max(exp, min(floor, e_al))) <= 0.02 * exp + floor, (e_op, exp)   # pixel sampling of G + the skip
    if exp > 4 * floor:
        assert e_op > 2 * e_al
    print(f"lod remap KAT k={k} o={o}: per-Gaussian remap max|children - parent| = {e_op:.5f} (predicted {exp:.5f}); "
          f"per-pixel remap = {e_al:.2e}")


def test_saturation_stop_rule_and_last_contributor():
    """App. A.8: a pixel is finished when T (1 - alpha) would drop below 1e-4 -- the Gaussian that triggers the test is
    NOT blended.  Five very wide layers of opacity 0.95 on one axis: T = 1 -> 0.05 -> 0.0025 -> 1.25e-4, the fourth layer
    would give 6.25e-6 < 1e-4 and ends the pixel: three layers blended, final T = 1.25e-4, last contributor = 3.
    (Numbers chosen away from the threshold: with alpha = 0.99 the second layer sits at 1e-4 +- rounding.)"""
    cam = synth.make_camera(32, 32)
    zs = [2.0, 3.0, 4.0, 5.0, 6.0]
    cols = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), (1.0, 1.0, 0.0), (0.0, 1.0, 1.0)]
    sc = synth.Scene(torch.tensor([[0.0, 0.0, z] for z in zs]), torch.tensor([[20.0 * z / 2.0] * 3 for z in zs]),
                     torch.tensor([[1.0, 0, 0, 0]] * 5), torch.full((5, 1), 0.95), torch.zeros(5, 16, 3), 0)
    for i, c in enumerate(cols):                              # SH degree 0: rgb = 0.5 + C0 * dc
        sc.shs[i, 0] = (torch.tensor(c) - 0.5) / 0.28209479177387814
    out = _render(sc, cam, torch.tensor([0.5, 0.5, 0.5]))
    y = x = 16
    w = [0.95, 0.95 * 0.05, 0.95 * 0.0025]                    # layer weights (the footprint factor is 1 - 1e-5 here)
    assert int(out.n_contrib[y, x]) == 3
    assert abs(float(out.final_T[y, x]) - 1.25e-4) < 1e-7
    expect = np.array([w[0], w[1], w[2]]) + 1.25e-4 * 0.5
    assert np.abs(out.color[:, y, x].numpy() - expect).max() < 2e-5
    assert abs(float(out.invdepth[0, y, x]) - (w[0] / 2.0 + w[1] / 3.0 + w[2] / 4.0)) < 2e-5


def test_frustum_cull_keeps_gaussians_up_to_the_ewa_clamp():
    """App. A.3-4: only z <= 0.2 culls; a centre far outside the image is kept, and its view-space x / z is clamped to
    1.3 tan(fov / 2) inside the EWA Jacobian (the projected centre itself is not clamped): the Gaussian keeps a positive
    radius but touches no tile once its 3-sigma square leaves the image."""
    cam = synth.make_camera(64, 64)
    inside = _render(_one(z=4.0, s=0.05, xy=(0.0, 0.0)), cam)
    off = _render(_one(z=4.0, s=0.05, xy=(4.0 * 3.0 * cam.tanfovx, 0.0)), cam)      # x / z = 3 tan(fov / 2)
    assert int(inside.radii[0]) > 0 and inside.binning.num_rendered > 0
    assert int(off.radii[0]) == 0 and off.binning.num_rendered == 0                  # nothing to draw: radius reported 0
    # just outside the right edge, but its 3-sigma square still reaches the last tile column
    edge = _render(_one(z=4.0, s=0.3, xy=(4.0 * 1.05 * cam.tanfovx, 0.0)), cam)
    assert int(edge.radii[0]) > 0 and edge.binning.num_rendered > 0
    gx = edge.geom.grid[0]
    assert int(edge.geom.rect_max[0, 0]) == gx and int(edge.geom.rect_min[0, 0]) < gx


def test_ewa_projection_matches_the_textbook_formulas():
    """EWA splatting (Zwicker et al.) written out independently of the oracle's code: Sigma = R S S R^T
    with R from the unit quaternion, view-space mean t = W p + T, J = [[fx/tz, 0, -fx tx/tz^2], [0, fy/tz, -fy ty/tz^2]]
    with tx/tz, ty/tz clamped to +-1.3 tan(fov/2), Sigma' = J W Sigma W^T J^T + 0.3 I, conic = Sigma'^-1,
    radius = ceil(3 sqrt(lambda_max)), pixel centre = ((ndc + 1) size - 1) / 2."""
    rng = np.random.default_rng(5)
    n = 64
    ang = 0.3
    Rc = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])   # camera-to-world
    Tc = np.array([0.1, -0.2, 0.3])
    cam = synth.make_camera(200, 120, R=Rc, T=Tc)
    p = np.stack([rng.uniform(-2.5, 2.5, n), rng.uniform(-1.2, 1.2, n), rng.uniform(2.0, 6.0, n)], 1)
    s = np.exp(rng.uniform(math.log(0.01), math.log(0.3), (n, 3)))
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    geom = ro.geometry_spec(p.astype(np.float32), s.astype(np.float32), q.astype(np.float32), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), 200, 120,
                            float(np.float32(cam.tanfovx)), float(np.float32(cam.tanfovy)), 1.0)
    Wm = Rc.T                                                      # world-to-camera rotation
    fx, fy = 200 / (2 * cam.tanfovx), 120 / (2 * cam.tanfovy)
    checked = 0
    for i in range(n):
        r, x, y, z = q[i]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                      [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                      [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        Sigma = R @ np.diag(s[i] ** 2) @ R.T
        t = Wm @ p[i] + Tc
        if t[2] <= 0.2:
            assert not geom.visible[i]
            continue
        lx, ly = 1.3 * cam.tanfovx, 1.3 * cam.tanfovy
        tx = min(lx, max(-lx, t[0] / t[2])) * t[2]
        ty = min(ly, max(-ly, t[1] / t[2])) * t[2]
        J = np.array([[fx / t[2], 0, -fx * tx / t[2] ** 2], [0, fy / t[2], -fy * ty / t[2] ** 2]])
        cov = J @ Wm @ Sigma @ Wm.T @ J.T + 0.3 * np.eye(2)
        conic = np.linalg.inv(cov)
        lam = 0.5 * (cov[0, 0] + cov[1, 1]) + math.sqrt(max(0.1, (0.5 * (cov[0, 0] + cov[1, 1])) ** 2 - np.linalg.det(cov)))
        rad = math.ceil(3 * math.sqrt(lam))
        px = ((t[0] / t[2] / cam.tanfovx + 1) * 200 - 1) / 2
        py = ((t[1] / t[2] / cam.tanfovy + 1) * 120 - 1) / 2
        assert abs(geom.px[i] - px) < 2e-3 and abs(geom.py[i] - py) < 2e-3
        got = np.array([[geom.conic[i, 0], geom.conic[i, 1]], [geom.conic[i, 1], geom.conic[i, 2]]], dtype=np.float64)
        assert np.abs(got - conic).max() <= 2e-4 * np.abs(conic).max(), (i, got, conic)
        assert abs(float(geom.depth[i]) - t[2]) < 1e-5
        if geom.tiles_touched[i] > 0:
            assert int(geom.radii[i]) in (rad - 1, rad, rad + 1)   # float32 ceil at an integer boundary may differ by one
            assert int(geom.radii[i]) == rad or abs(3 * math.sqrt(lam) - round(3 * math.sqrt(lam))) < 1e-3
            checked += 1
    assert checked >= 20
