"""A miniature of the reference's optimisation loop (train_single.py:57-190), renderer-agnostic, used by the
PSNR-parity tests: raw parameters with the reference's activations (scene/gaussian_model.py:108-128: exp scales,
normalised quaternions, sigmoid opacity, cat(dc, rest) SH), one camera per step (train_single.py:57-59), L1 colour
loss plus an inverse-depth L1 term (train_single.py:110-117, without the DSSIM term), Adam with per-group learning
rates in the proportions of arguments/__init__.py:86-95, PSNR as utils/image_utils.py:17-19.
"""
import torch

from hgs import synth


def psnr(img, target):
    mse = ((img.double() - target.double()) ** 2).mean()
    return (20.0 * torch.log10(1.0 / torch.sqrt(mse))).item()


def raw_params_from_scene(scene, device, jitter_seed=None):
    g = torch.Generator().manual_seed(0 if jitter_seed is None else jitter_seed)
    j = (lambda t, s: t) if jitter_seed is None else (lambda t, s: t + s * torch.randn(t.shape, generator=g))
    op = scene.opacities.clamp(1e-4, 1 - 1e-4)
    raw = dict(
        xyz=j(scene.means3D, 0.02),
        f_dc=j(scene.shs[:, :1], 0.3),
        f_rest=j(scene.shs[:, 1:], 0.02),
        opacity=j(torch.log(op / (1 - op)), 0.5),
        scaling=j(torch.log(scene.scales), 0.2),
        rotation=j(scene.rotations, 0.1),
    )
    return {k: v.clone().to(device).contiguous().requires_grad_(True) for k, v in raw.items()}


def activate(raw):
    return dict(means3D=raw["xyz"], shs=torch.cat([raw["f_dc"], raw["f_rest"]], dim=1).contiguous(),
                opacities=torch.sigmoid(raw["opacity"]), scales=torch.exp(raw["scaling"]),
                rotations=torch.nn.functional.normalize(raw["rotation"], dim=1))


LRS = dict(xyz=1.6e-3, f_dc=2.5e-2, f_rest=2.5e-2 / 20.0, opacity=5e-2, scaling=5e-3, rotation=1e-3)


def ssim(img1, img2, window_size=11, sigma=1.5):
    """Mean SSIM with an 11x11 Gaussian window (sigma 1.5), C1 = 0.01^2, C2 = 0.03^2, zero padding -- the standard
    definition, the one the reference's loss uses (utils/loss_utils.py:33-63)."""
    x = torch.arange(window_size, dtype=img1.dtype, device=img1.device) - window_size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    c = img1.shape[-3]
    win = (g[:, None] * g[None, :]).expand(c, 1, window_size, window_size).contiguous()
    conv = lambda t: torch.nn.functional.conv2d(t[None], win, padding=window_size // 2, groups=c)[0]
    mu1, mu2 = conv(img1), conv(img2)
    s1, s2, s12 = conv(img1 * img1) - mu1 * mu1, conv(img2 * img2) - mu2 * mu2, conv(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()


def optimise(render_fn, raw, cams, targets, steps, depth_weight=0.1, lambda_dssim=0.0):
    """render_fn(cam, activated dict) -> (color [3,H,W], invdepth [1,H,W]) on the parameters' device.
    targets: list of (color, invdepth) per camera.  ``lambda_dssim`` > 0: the reference's colour loss
    (1 - l) * L1 + l * (1 - SSIM) (train_single.py:101-108, arguments/__init__.py:98: l = 0.2).
    Returns the per-step loss list."""
    opt = torch.optim.Adam([dict(params=[raw[k]], lr=LRS[k], name=k) for k in LRS], eps=1e-15)
    losses = []
    for it in range(steps):
        k = it % len(cams)
        color, invd = render_fn(cams[k], activate(raw))
        tc, td = targets[k]
        tc = tc.to(color)
        l1 = (color - tc).abs().mean()
        if lambda_dssim > 0:
            l1 = (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim(color, tc))
        loss = l1 + depth_weight * (invd - td.to(invd)).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses


def evaluate(render_fn, raw, cams, targets):
    with torch.no_grad():
        vals = []
        for cam, (tc, _) in zip(cams, targets):
            color, _ = render_fn(cam, {k: v.detach() for k, v in activate(raw).items()})
            vals.append(psnr(color.clamp(0, 1).cpu(), tc.clamp(0, 1).cpu()))
    return sum(vals) / len(vals)


def oracle_render_fn(bg, sh_degree, dtype=torch.float64):
    from oracle import raster_oracle as ro

    def fn(cam, a):
        m2 = torch.zeros(a["means3D"].shape[0], 3)
        out = ro.rasterize(a["means3D"], m2, a["shs"], None, a["opacities"], a["scales"], a["rotations"], None,
                           image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx,
                           tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                           projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center,
                           dtype=dtype)
        return out.color.float(), out.invdepth.float()
    return fn


def hip_render_fn(bg, sh_degree, device):
    import diff_gaussian_rasterization as dgr
    from parity import settings_kwargs

    def fn(cam, a):
        rs = dgr.GaussianRasterizationSettings(**settings_kwargs(cam, bg, sh_degree, do_depth=True, device=device))
        m2 = torch.zeros(a["means3D"].shape[0], 3, device=device, requires_grad=a["means3D"].requires_grad)
        color, _, invd = dgr.GaussianRasterizer(raster_settings=rs)(
            means3D=a["means3D"], means2D=m2, shs=a["shs"], colors_precomp=None, opacities=a["opacities"],
            scales=a["scales"], rotations=a["rotations"], cov3D_precomp=None)
        return color, invd
    return fn


def make_problem(P=1000, size=128, n_views=4, seed=0, height=None):
    height = size if height is None else height
    cams = [synth.orbit_camera(size, height, k, n_views) for k in range(n_views)]
    scene = synth.make_scene(P, synth.make_camera(size, height), seed=seed)
    return cams, scene
