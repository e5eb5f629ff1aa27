"""Raw-parameter path (include/hgs.h HGS_ACT_*, shs_rest): the op applies exp / normalize / sigmoid|abs
and reads features_dc / features_rest separately.  Checked against the oracle driven through the activation spec
(oracle.raster_oracle.activate_raw): indices bit-exact, pixels and gradients w.r.t. the RAW tensors <= 1e-5, and
against the op's own standard path fed with torch-activated inputs."""
import pytest
import torch

import parity as pa
from hgs import synth
from oracle import raster_oracle as ro

pytestmark = pytest.mark.gpu


def _raw_from_scene(scene, seed, logit=True):
    g = torch.Generator().manual_seed(seed)
    op = scene.opacities.clamp(1e-4, 1 - 1e-4)
    return dict(
        xyz=scene.means3D.clone(),
        f_dc=scene.shs[:, :1].contiguous().clone(),
        f_rest=scene.shs[:, 1:].contiguous().clone(),
        opacity=(torch.log(op / (1 - op)) if logit else op * torch.where(torch.rand(op.shape, generator=g) < 0.5, -1.0, 1.0)),
        scaling=torch.log(scene.scales),
        rotation=scene.rotations * (0.5 + torch.rand(scene.P, 1, generator=g) * 2.0),   # un-normalised
    )


def _run_oracle_raw(raw, cam, bg, gc, gd, sh_degree, act):
    leaves = {k: v.clone().double().requires_grad_(True) for k, v in raw.items()}
    s, r, o = ro.activate_raw(leaves["scaling"], leaves["rotation"], leaves["opacity"], act)
    shs = torch.cat([leaves["f_dc"], leaves["f_rest"]], 1)
    m2 = torch.zeros(raw["xyz"].shape[0], 3, dtype=torch.float64, requires_grad=True)
    out = ro.rasterize(leaves["xyz"], m2, shs, None, o, s, r, None, image_height=cam.image_height,
                       image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg,
                       scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                       sh_degree=sh_degree, campos=cam.camera_center)
    ((out.color * gc.double()).sum() + (out.invdepth * gd.double()).sum()).backward()
    grads = {k: v.grad for k, v in leaves.items()}
    grads["means2D"] = m2.grad
    return out, grads


def _run_hip_raw(raw, cam, bg, gc, gd, sh_degree, act, device, debug=True):
    import diff_gaussian_rasterization as dgr
    leaves = {k: v.clone().to(device).requires_grad_(True) for k, v in raw.items()}
    m2 = torch.zeros(raw["xyz"].shape[0], 3, device=device, requires_grad=True)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, bg, sh_degree, do_depth=True, debug=debug,
                                                                device=device))
    color, radii, invd = dgr.GaussianRasterizer(rs).forward_raw(
        leaves["xyz"], m2, leaves["f_dc"], leaves["f_rest"], leaves["opacity"], leaves["scaling"],
        leaves["rotation"], opacity_activation=act)
    call = color.grad_fn.call
    views = {k: v.cpu().clone() for k, v in dgr._C.raster_views(call).items()}
    ((color * gc.to(device)).sum() + (invd * gd.to(device)).sum()).backward()
    torch.cuda.synchronize()
    grads = {k: v.grad.detach().cpu() for k, v in leaves.items()}
    grads["means2D"] = m2.grad.detach().cpu()
    return dict(color=color.detach().cpu(), radii=radii.cpu(), invdepth=invd.detach().cpu(), views=views, L=call.L,
                grads=grads)


def _compare(hip, oo, og):
    idx = pa.check_indices(hip, oo)
    assert all(v == 0 for v in idx.values()), idx
    st = pa.compare(hip, oo, og)
    assert st["fragile_frac"] <= pa.FRAGILE_FRAC
    for k, v in st.items():
        if isinstance(v, dict):
            assert v["maxrel"] <= pa.REL_TOL and v["l2"] <= pa.REL_TOL, (k, v)


@pytest.mark.parametrize("P,size,deg,act", [(1000, 128, 3, "sigmoid"), (1000, 128, 3, "abs"), (777, 96, 1, "sigmoid"),
                                            (5000, 256, 2, "sigmoid")])
def test_raw_path_matches_oracle(gpu, P, size, deg, act):
    cam = synth.make_camera(size, size)
    scene = synth.make_scene(P, cam, seed=3, sh_degree=deg)
    raw = _raw_from_scene(scene, seed=4, logit=(act == "sigmoid"))
    bg = torch.tensor([0.1, 0.2, 0.3])
    gc, gd = synth.upstream_grads(size, size, seed=1)
    oo, og = _run_oracle_raw(raw, cam, bg, gc, gd, deg, act)
    hip = _run_hip_raw(raw, cam, bg, gc, gd, deg, act, gpu)
    _compare(hip, oo, og)


def test_raw_path_equals_standard_path_on_activated_inputs(gpu):
    """Same op, two entrances: raw tensors + fused activations vs torch activations + cat (what
    scene/gaussian_model.py:108-128 does).  Equal up to float32 rounding of the activations."""
    import diff_gaussian_rasterization as dgr
    W, H = 640, 360
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(50_000, cam, seed=8)
    raw = _raw_from_scene(scene, seed=9)
    bg = torch.zeros(3)
    gc, gd = synth.upstream_grads(H, W, seed=1)
    hip = _run_hip_raw(raw, cam, bg, gc, gd, 3, "sigmoid", gpu, debug=False)
    leaves = {k: v.clone().to(gpu).requires_grad_(True) for k, v in raw.items()}
    m2 = torch.zeros(scene.P, 3, device=gpu, requires_grad=True)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, bg, 3, do_depth=True, device=gpu))
    color, radii, invd = dgr.GaussianRasterizer(rs)(
        means3D=leaves["xyz"], means2D=m2, shs=torch.cat([leaves["f_dc"], leaves["f_rest"]], 1),
        opacities=torch.sigmoid(leaves["opacity"]), scales=torch.exp(leaves["scaling"]),
        rotations=torch.nn.functional.normalize(leaves["rotation"]))
    ((color * gc.to(gpu)).sum() + (invd * gd.to(gpu)).sum()).backward()
    assert (radii.cpu() != hip["radii"]).float().mean() < 1e-4      # activations differ in the last float32 bit
    assert pa.err_stats(color.detach().cpu(), hip["color"])["l2"] < 1e-5
    for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
        st = pa.err_stats(leaves[k].grad.cpu(), hip["grads"][k])
        assert st["l2"] < 1e-4, (k, st)


def test_raw_path_argument_checks(gpu):
    import diff_gaussian_rasterization as dgr
    cam = synth.make_camera(64, 64)
    scene = synth.make_scene(10, cam, seed=0).to(gpu)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, device=gpu))
    m2 = torch.zeros(10, 3, device=gpu)
    with pytest.raises(RuntimeError):      # features_dc must be [P,1,3]
        dgr.GaussianRasterizer(rs).forward_raw(scene.means3D, m2, scene.shs[:, :2].contiguous(),
                                               scene.shs[:, 2:].contiguous(), scene.opacities, scene.scales,
                                               scene.rotations)
    with pytest.raises(RuntimeError):
        dgr.GaussianRasterizer(rs).forward_raw(scene.means3D, m2, scene.shs[:, :1].contiguous(),
                                               scene.shs[:, 1:].contiguous(), scene.opacities, scene.scales,
                                               scene.rotations, opacity_activation="tanh")
    # a contiguous slice whose first byte is not 16-byte aligned (12-byte rows) is refused, not mis-read
    dc11 = torch.zeros(11, 1, 3, device=gpu)
    with pytest.raises(RuntimeError, match="16-byte"):
        dgr.GaussianRasterizer(rs).forward_raw(scene.means3D, m2, dc11[1:], scene.shs[:, 1:].contiguous(),
                                               scene.opacities, scene.scales, scene.rotations)
