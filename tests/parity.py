"""Shared helpers of the GPU parity tests: run the HIP op and the CPU oracle on the same
seeded inputs and report the differences.

Tolerances (BASELINE.json north_star: 'bit-exact on tile/sort indices, within 1e-5 rel on
pixels and gradients'):
  * integers (radii, tile rects, depth bits, sorted keys, point list, tile ranges): exact;
  * pixels / gradients: max|hip - oracle| <= REL_TOL * max|oracle| per tensor, and relative
    L2 error <= REL_TOL, oracle in float64.  Pixels where the oracle reports a discrete blend
    decision within 1e-5 (relative) of its threshold ("fragile": alpha vs 1/255, T vs 1e-4)
    are excluded from the pixel comparison; their count is bounded by FRAGILE_FRAC.
    The same pixels are excluded from the LOSS whose gradients are compared (``run_oracle`` zeroes the upstream
    gradient there and hands the mask to the ``run_hip`` call that follows): a float32 and a float64 evaluation of
    ``T (1 - alpha) < 1e-4`` legitimately disagree on a knife edge -- e.g. two stacked Gaussians capped at alpha = 0.99
    give T = 9.99998e-5 in float32 (stop, as the float32 reference lineage does) and 1.0000000000000002e-4 in float64
    (continue) -- and one such pixel changes a covering Gaussian's gradient by percents (found by fuzz seed 1002).
"""
import numpy as np
import torch

from hgs import synth
from oracle import raster_oracle as ro

REL_TOL = 1e-5
FRAGILE_FRAC = 1e-3
# element-wise bound: |hip - oracle| <= MIXED_REL * |oracle| + MIXED_ABS * max|oracle| for EVERY
# entry of every compared tensor -- a relative bound on the entries that matter plus an absolute floor (in units of the
# tensor's largest entry) for entries that are small sums of large cancelling terms
MIXED_REL = 1e-5
MIXED_ABS = 1e-6
_PAIRED = {}     # fragile-pixel mask of the last run_oracle, consumed by the next run_hip of the same image size


def settings_kwargs(cam, bg, sh_degree, do_depth=True, debug=False, scale_modifier=1.0, device="cpu",
                    interpolation_weights=None, num_node_kids=None):
    e_i = torch.empty(0, dtype=torch.int32, device=device)
    e_f = torch.empty(0, dtype=torch.float32, device=device)
    return dict(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx,
                tanfovy=cam.tanfovy, bg=bg.to(device), scale_modifier=scale_modifier,
                viewmatrix=cam.world_view_transform.to(device), projmatrix=cam.full_proj_transform.to(device),
                sh_degree=sh_degree, campos=cam.camera_center.to(device), prefiltered=False, debug=debug,
                do_depth=do_depth, render_indices=e_i, parent_indices=e_i,
                interpolation_weights=e_f if interpolation_weights is None else interpolation_weights.to(device),
                num_node_kids=e_i if num_node_kids is None else num_node_kids.to(device))


def run_oracle(scene, cam, bg, gc, gd, *, colors_precomp=None, cov3D_precomp=None, scale_modifier=1.0,
               interpolation_weights=None, num_node_kids=None, do_depth=True, dtype=torch.float64, mask_fragile=True,
               lod_mode="opacity", positive_power="skip"):
    req = lambda t: None if t is None else t.clone().requires_grad_(True)
    m3, sc, rot, op = req(scene.means3D), req(scene.scales), req(scene.rotations), req(scene.opacities)
    sh = req(scene.shs) if colors_precomp is None else None
    col = req(colors_precomp)
    cov = req(cov3D_precomp)
    if cov is not None:
        sc = rot = None
    m2 = torch.zeros(scene.P, 3, requires_grad=True)
    out = ro.rasterize(m3, m2, sh, col, op, sc, rot, cov, image_height=cam.image_height,
                       image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg,
                       scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform,
                       projmatrix=cam.full_proj_transform, sh_degree=scene.sh_degree, campos=cam.camera_center,
                       interpolation_weights=interpolation_weights, num_node_kids=num_node_kids, dtype=dtype,
                       lod_mode=lod_mode, positive_power=positive_power)
    # see the module docstring: undecidable pixels leave the loss
    ok = torch.from_numpy(~out.fragile) if mask_fragile else torch.ones(out.fragile.shape, dtype=torch.bool)
    out.grad_mask = ok
    _PAIRED.clear()
    if not bool(ok.all()):
        _PAIRED[tuple(ok.shape)] = ok
    loss = (out.color * (gc * ok).to(dtype)).sum()
    if do_depth:
        loss = loss + (out.invdepth * (gd * ok).to(dtype)).sum()
    loss.backward()
    grads = dict(means3D=m3.grad, means2D=m2.grad, opacities=op.grad)
    if sh is not None: grads["shs"] = sh.grad
    if col is not None: grads["colors_precomp"] = col.grad
    if sc is not None: grads["scales"] = sc.grad; grads["rotations"] = rot.grad
    if cov is not None: grads["cov3D_precomp"] = cov.grad
    return out, grads


def run_hip(scene, cam, bg, gc, gd, device, *, colors_precomp=None, cov3D_precomp=None, scale_modifier=1.0,
            interpolation_weights=None, num_node_kids=None, do_depth=True, debug=True, grad_mask="paired"):
    """``grad_mask``: [H,W] bool mask applied to the upstream gradients; "paired" (default) = the fragile-pixel mask of
    the ``run_oracle`` call that preceded this one (same image size; used once), None = no mask."""
    import diff_gaussian_rasterization as dgr
    if isinstance(grad_mask, str):
        grad_mask = _PAIRED.pop(tuple(gc.shape[-2:]), None)
    if grad_mask is not None:
        gc, gd = gc * grad_mask, gd * grad_mask
    req = lambda t: None if t is None else t.clone().to(device).requires_grad_(True)
    m3, sc, rot, op = req(scene.means3D), req(scene.scales), req(scene.rotations), req(scene.opacities)
    sh = req(scene.shs) if colors_precomp is None else None
    col = req(colors_precomp)
    cov = req(cov3D_precomp)
    if cov is not None:
        sc = rot = None
    m2 = torch.zeros(scene.P, 3, device=device, requires_grad=True)
    rs = dgr.GaussianRasterizationSettings(**settings_kwargs(
        cam, bg, scene.sh_degree, do_depth=do_depth, debug=debug, scale_modifier=scale_modifier, device=device,
        interpolation_weights=interpolation_weights, num_node_kids=num_node_kids))
    rast = dgr.GaussianRasterizer(raster_settings=rs)
    color, radii, invd = rast(means3D=m3, means2D=m2, shs=sh, colors_precomp=col, opacities=op, scales=sc,
                              rotations=rot, cov3D_precomp=cov)
    call = color.grad_fn.call if color.grad_fn is not None else None
    views = dgr._C.raster_views(call) if call is not None else None
    views_cpu = {k: v.cpu().clone() for k, v in views.items()} if views else None
    L = call.L if call is not None else 0
    loss = (color * gc.to(device)).sum()
    if do_depth:
        loss = loss + (invd * gd.to(device)).sum()
    loss.backward()
    torch.cuda.synchronize()
    grads = dict(means3D=m3.grad, means2D=m2.grad, opacities=op.grad)
    if sh is not None: grads["shs"] = sh.grad
    if col is not None: grads["colors_precomp"] = col.grad
    if sc is not None: grads["scales"] = sc.grad; grads["rotations"] = rot.grad
    if cov is not None: grads["cov3D_precomp"] = cov.grad
    return dict(color=color.detach().cpu(), radii=radii.cpu(), invdepth=invd.detach().cpu(), views=views_cpu, L=L,
                grads={k: v.detach().cpu() for k, v in grads.items()})


def err_stats(hip, ref):
    """Norm-wise (``maxrel`` = max|d| / max|ref|, ``l2``) AND element-wise figures of one tensor pair:
    ``p999_rel``  99.9th percentile of |d| / |ref| over the entries with |ref| >= 1e-3 max|ref|;
    ``mixed``     max over ALL entries of |d| / (MIXED_REL |ref| + MIXED_ABS max|ref|)  (<= 1 passes ``assert_stats``)."""
    a, b = hip.double().reshape(-1), ref.double().reshape(-1)
    scale = b.abs().max().item() if b.numel() else 0.0
    if scale == 0:
        return dict(maxrel=a.abs().max().item() if a.numel() else 0.0, l2=a.norm().item(), scale=0.0, p999_rel=0.0,
                    mixed=0.0 if not a.numel() or a.abs().max().item() == 0 else float("inf"))
    d = (a - b).abs()
    big = b.abs() >= 1e-3 * scale
    rel = (d[big] / b.abs()[big])
    p999 = torch.quantile(rel, 0.999).item() if rel.numel() <= 16_000_000 else \
        rel.kthvalue(max(1, int(0.999 * rel.numel()))).values.item()
    return dict(maxrel=(d.max() / scale).item(), l2=((a - b).norm() / b.norm()).item(), scale=scale, p999_rel=p999,
                mixed=(d / (MIXED_REL * b.abs() + MIXED_ABS * scale)).max().item())


def assert_stats(name, stats, rel_tol=REL_TOL, mixed_tol=1.0):
    """Every tensor of a ``compare`` result within tolerance: norm-wise (maxrel, relative L2 <= rel_tol) and
    element-wise (``mixed`` <= mixed_tol, see MIXED_REL / MIXED_ABS)."""
    for k, v in stats.items():
        if not isinstance(v, dict):
            continue
        assert v["maxrel"] <= rel_tol, f"{name}: {k} max error {v['maxrel']:.3e} (rel. to max) > {rel_tol}"
        assert v["l2"] <= rel_tol, f"{name}: {k} rel-L2 error {v['l2']:.3e} > {rel_tol}"
        assert v["mixed"] <= mixed_tol, (f"{name}: {k} element-wise error {v['mixed']:.3f} x the bound "
                                         f"{MIXED_REL}|ref| + {MIXED_ABS} max|ref|")


def rows_touching_fragile(oracle_out):
    """How many Gaussians (gradient rows) have a fragile pixel inside their footprint rectangle -- an upper bound on
    the rows whose gradients lost a term to the fragile-pixel mask."""
    fr = oracle_out.fragile
    if not fr.any():
        return 0
    g = oracle_out.geom
    H, W = fr.shape
    ii = np.zeros((H + 1, W + 1), dtype=np.int64)
    ii[1:, 1:] = fr.astype(np.int64).cumsum(0).cumsum(1)
    vis = g.visible
    x0 = np.clip(np.floor(g.px[vis] - g.radii[vis]).astype(np.int64), 0, W)
    x1 = np.clip(np.ceil(g.px[vis] + g.radii[vis]).astype(np.int64) + 1, 0, W)
    y0 = np.clip(np.floor(g.py[vis] - g.radii[vis]).astype(np.int64), 0, H)
    y1 = np.clip(np.ceil(g.py[vis] + g.radii[vis]).astype(np.int64) + 1, 0, H)
    cnt = ii[y1, x1] - ii[y0, x1] - ii[y1, x0] + ii[y0, x0]
    return int((cnt > 0).sum())


def check_indices(hip, oracle_out):
    """Bit-exact comparison of every integer the forward produces.  Returns a dict of mismatch counts."""
    geom, binning = oracle_out.geom, oracle_out.binning
    v = hip["views"]
    res = {}
    res["radii"] = int((hip["radii"].numpy() != geom.radii).sum())
    res["tiles_touched"] = int((v["tiles_touched"].numpy().astype(np.uint32) != geom.tiles_touched).sum())
    res["num_rendered"] = abs(int(hip["L"]) - int(binning.num_rendered))
    vis = geom.visible
    res["depth_bits"] = int((v["depths"].numpy().view(np.uint32)[vis] != geom.depth.view(np.uint32)[vis]).sum())
    rects = v["rects"].numpy().astype(np.uint32)
    rmin = (geom.rect_min[:, 0].astype(np.uint32) | (geom.rect_min[:, 1].astype(np.uint32) << 16))
    rmax = (geom.rect_max[:, 0].astype(np.uint32) | (geom.rect_max[:, 1].astype(np.uint32) << 16))
    res["rects"] = int(((rects[:, 0] != rmin) | (rects[:, 1] != rmax))[vis].sum())
    if res["num_rendered"] == 0:
        excl = np.cumsum(geom.tiles_touched.astype(np.int64)) - geom.tiles_touched
        res["offsets"] = int((v["offsets"].numpy().astype(np.int64)[vis] != excl[vis]).sum())
        # the op sorts by tile globally and by depth per tile; its (tile | depth) key sequence is rebuilt here
        keys = (v["tile_ids_sorted"].numpy().astype(np.uint64) << np.uint64(32)) | \
            v["depths"].numpy().view(np.uint32)[v["point_list"].numpy()].astype(np.uint64)
        res["keys_sorted"] = int((keys != binning.keys_sorted).sum())
        res["point_list"] = int((v["point_list"].numpy() != binning.point_list).sum())
        res["ranges"] = int((v["ranges"].numpy() != binning.ranges).sum())
    return res


def compare(hip, oracle_out, oracle_grads, do_depth=True):
    ok = torch.from_numpy(~oracle_out.fragile)
    stats = {"fragile_frac": float(oracle_out.fragile.mean()), "rows_touching_fragile": rows_touching_fragile(oracle_out)}
    c_h, c_o = hip["color"][:, ok], oracle_out.color.detach()[:, ok]
    stats["color"] = err_stats(c_h, c_o)
    if do_depth:
        stats["invdepth"] = err_stats(hip["invdepth"][:, ok], oracle_out.invdepth.detach()[:, ok])
    for k, g in oracle_grads.items():
        stats["d_" + k] = err_stats(hip["grads"][k], g)
    return stats


def default_case(P, W, H, seed=0, sh_degree=3, fovy=60.0):
    cam = synth.make_camera(W, H, fovy)
    scene = synth.make_scene(P, cam, seed=seed, sh_degree=sh_degree)
    gc, gd = synth.upstream_grads(H, W)
    return cam, scene, gc, gd


def chain_to_raw(raw, og, opacity_activation="sigmoid"):
    """The oracle's gradients w.r.t. the ACTIVATED attributes chained, in float64, through the activations of
    scene/gaussian_model.py:108-128 (exp, normalize, sigmoid or abs, cat of features_dc / features_rest) to the model's
    raw parameters.  raw: dict of float tensors (xyz, scaling, rotation, opacity, features_dc, features_rest); og: the
    gradient dict of ``run_oracle``.  Returns {name: gradient w.r.t. raw[name]} (float64)."""
    leaf = {k: v.detach().double().clone().requires_grad_(True) for k, v in raw.items()}
    act = dict(means3D=leaf["xyz"], scales=torch.exp(leaf["scaling"]),
               rotations=torch.nn.functional.normalize(leaf["rotation"]),
               opacities=torch.sigmoid(leaf["opacity"]) if opacity_activation == "sigmoid" else torch.abs(leaf["opacity"]),
               shs=torch.cat((leaf["features_dc"], leaf["features_rest"]), dim=1))
    keys = [k for k in act if k in og]
    torch.autograd.backward([act[k] for k in keys], [og[k].double().reshape(act[k].shape) for k in keys])
    return {k: v.grad for k, v in leaf.items()}
