"""CPU oracle for the differentiable hierarchical-Gaussian rasterizer.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.

PARITY UNPINNED: the reference's implementation of this path lives in the
un-vendored submodule ``graphdeco-inria/hierarchy-rasterizer`` (declared at
/root/reference/.gitmodules:4-6, directory empty, pinned SHA unknown) and the
reference ships no tests or golden vectors for it.  This file restates the
public tile-based 3D-Gaussian-splatting rasterization algorithm that the
submodule's Python call sites imply, and is pinned only by

* the reference's in-tree Python twins of the kernel math
  (utils/sh_utils.py:57-112 ``eval_sh``, utils/general_utils.py:82-114
  ``build_rotation``/``build_scaling_rotation``, utils/graphics_utils.py:38-77
  camera matrices) -- see tests/golden/ and tests/test_oracle_pins.py,
* hand-derived known answers (tests/test_oracle_kat.py).

Structure
---------
``geometry_spec``  numpy *float32*, one rounding per operation, operation order
                   fixed below.  It decides everything discrete: culling,
                   screen radius, tile rectangle, depth bits, (tile|depth)
                   keys, the stable sort and the tile ranges.  The HIP
                   preprocess kernel follows the same operation order with
                   floating-point contraction disabled, so these integers are
                   compared bit-exactly.
``rasterize``      torch (float64 by default), differentiable.  Recomputes the
                   continuous quantities from the raw inputs and blends every
                   tile densely ([pixels x sorted Gaussians]); autograd yields
                   the oracle gradients.  Compared within a tolerance.

Call sites being restated: gaussian_renderer/__init__.py:44-62,105-113
(settings + op call), :247-277 (hierarchy mode), train_single.py:97,123
(forward + backward).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

TILE = 16
NEAR_Z = np.float32(0.2)
ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.99
T_EPS = 1e-4
FRAGILE_FP32_K = 8.0 * 2.0 ** -24     # float32 roundings of the exponent's chain, per unit of its terms' magnitudes
F = np.float32

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


# --------------------------------------------------------------------------
# float32 geometry specification (bit-exact contract)
# --------------------------------------------------------------------------
@dataclass
class Geometry:
    visible: np.ndarray        # [P] bool
    radii: np.ndarray          # [P] int32 (0 when culled)
    depth: np.ndarray          # [P] float32 view-space z
    px: np.ndarray             # [P] float32 pixel-centre x
    py: np.ndarray             # [P] float32
    rect_min: np.ndarray       # [P,2] int32 (tile units, x,y)
    rect_max: np.ndarray       # [P,2] int32
    tiles_touched: np.ndarray  # [P] uint32
    conic: np.ndarray          # [P,3] float32 (A,B,C)
    cov3d: np.ndarray          # [P,6] float32
    grid: tuple                # (gx, gy)


def _quat_to_rot_f32(q):
    r, x, y, z = (q[:, 0], q[:, 1], q[:, 2], q[:, 3])
    two = F(2.0)
    one = F(1.0)
    R = np.empty((q.shape[0], 3, 3), dtype=np.float32)
    # utils/general_utils.py:94-102 (caller passes normalised quaternions)
    R[:, 0, 0] = one - two * (y * y + z * z)
    R[:, 0, 1] = two * (x * y - r * z)
    R[:, 0, 2] = two * (x * z + r * y)
    R[:, 1, 0] = two * (x * y + r * z)
    R[:, 1, 1] = one - two * (x * x + z * z)
    R[:, 1, 2] = two * (y * z - r * x)
    R[:, 2, 0] = two * (x * z - r * y)
    R[:, 2, 1] = two * (y * z + r * x)
    R[:, 2, 2] = one - two * (x * x + y * y)
    return R


def cov3d_spec(scales, rotations, scale_modifier):
    """Sigma = (R diag(s)) (R diag(s))^T, 6-vector (xx,xy,xz,yy,yz,zz).

    utils/general_utils.py:68-80,104-114; scene/gaussian_model.py:30-34.
    """
    s = (F(scale_modifier) * scales.astype(np.float32)).astype(np.float32)
    R = _quat_to_rot_f32(rotations.astype(np.float32))
    L = (R * s[:, None, :]).astype(np.float32)          # L_ik = R_ik * s_k
    out = np.empty((scales.shape[0], 6), dtype=np.float32)
    k = 0
    for i in range(3):
        for j in range(i, 3):
            out[:, k] = (L[:, i, 0] * L[:, j, 0] + L[:, i, 1] * L[:, j, 1]) + L[:, i, 2] * L[:, j, 2]
            k += 1
    return out


def _xform3(m, x, y, z, row):
    # m = flattened stored matrix (row-vector convention, scene/cameras.py:95-97)
    return ((m[row] * x + m[4 + row] * y) + m[8 + row] * z) + m[12 + row]


def geometry_spec(means3D, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                  W, H, tanfovx, tanfovy, scale_modifier=1.0) -> Geometry:
    """All-float32 restatement of the per-Gaussian geometry (SURVEY App. A 1-5,7)."""
    with np.errstate(all="ignore"):
        return _geometry_spec(means3D, scales, rotations, cov3D_precomp, viewmatrix,
                              projmatrix, W, H, tanfovx, tanfovy, scale_modifier)


def _geometry_spec(means3D, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                   W, H, tanfovx, tanfovy, scale_modifier):
    p = np.ascontiguousarray(means3D, dtype=np.float32)
    P = p.shape[0]
    vm = np.ascontiguousarray(viewmatrix, dtype=np.float32).reshape(16)
    pm = np.ascontiguousarray(projmatrix, dtype=np.float32).reshape(16)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    gx = (W + TILE - 1) // TILE
    gy = (H + TILE - 1) // TILE

    tx = _xform3(vm, x, y, z, 0)
    ty = _xform3(vm, x, y, z, 1)
    tz = _xform3(vm, x, y, z, 2)
    visible = tz > NEAR_Z

    hx = _xform3(pm, x, y, z, 0)
    hy = _xform3(pm, x, y, z, 1)
    hw = _xform3(pm, x, y, z, 3)
    pw = F(1.0) / (hw + F(1e-7))
    ndcx = hx * pw
    ndcy = hy * pw

    if cov3D_precomp is not None:
        c3 = np.ascontiguousarray(cov3D_precomp, dtype=np.float32)
    else:
        c3 = cov3d_spec(scales, rotations, scale_modifier)

    tfx = F(tanfovx)
    tfy = F(tanfovy)
    focal_x = F(W) / (F(2.0) * tfx)
    focal_y = F(H) / (F(2.0) * tfy)
    limx = F(1.3) * tfx
    limy = F(1.3) * tfy
    txtz = tx / tz
    tytz = ty / tz
    txc = np.minimum(limx, np.maximum(-limx, txtz)) * tz
    tyc = np.minimum(limy, np.maximum(-limy, tytz)) * tz
    tz2 = tz * tz
    J00 = focal_x / tz
    J02 = -(focal_x * txc) / tz2
    J11 = focal_y / tz
    J12 = -(focal_y * tyc) / tz2
    # view rotation, standard orientation: Wm[i][j] = vm[j*4+i]
    Wm = [[vm[j * 4 + i] for j in range(3)] for i in range(3)]
    T0 = [J00 * Wm[0][j] + J02 * Wm[2][j] for j in range(3)]
    T1 = [J11 * Wm[1][j] + J12 * Wm[2][j] for j in range(3)]
    S = [[c3[:, 0], c3[:, 1], c3[:, 2]],
         [c3[:, 1], c3[:, 3], c3[:, 4]],
         [c3[:, 2], c3[:, 4], c3[:, 5]]]
    U0 = [(T0[0] * S[0][j] + T0[1] * S[1][j]) + T0[2] * S[2][j] for j in range(3)]
    U1 = [(T1[0] * S[0][j] + T1[1] * S[1][j]) + T1[2] * S[2][j] for j in range(3)]
    a = ((U0[0] * T0[0] + U0[1] * T0[1]) + U0[2] * T0[2]) + F(0.3)
    b = (U0[0] * T1[0] + U0[1] * T1[1]) + U0[2] * T1[2]
    c = ((U1[0] * T1[0] + U1[1] * T1[1]) + U1[2] * T1[2]) + F(0.3)
    det = a * c - b * b
    visible &= (det != F(0.0))
    det_inv = F(1.0) / det
    conic = np.stack([c * det_inv, (-b) * det_inv, a * det_inv], axis=1).astype(np.float32)

    mid = F(0.5) * (a + c)
    disc = np.maximum(F(0.1), mid * mid - det)
    sq = np.sqrt(disc)
    lam = np.maximum(mid + sq, mid - sq)
    rad_f = np.ceil(F(3.0) * np.sqrt(lam))
    px = ((ndcx + F(1.0)) * F(W) - F(1.0)) * F(0.5)
    py = ((ndcy + F(1.0)) * F(H) - F(1.0)) * F(0.5)

    def _tile(v, hi):
        t = np.trunc(v * F(1.0 / TILE))
        t = np.where(np.isnan(t), F(0.0), t)
        return np.minimum(F(hi), np.maximum(F(0.0), t)).astype(np.int32)

    rminx = _tile(px - rad_f, gx)
    rmaxx = _tile(px + rad_f + F(TILE - 1), gx)
    rminy = _tile(py - rad_f, gy)
    rmaxy = _tile(py + rad_f + F(TILE - 1), gy)
    touched = ((rmaxx - rminx) * (rmaxy - rminy)).astype(np.int64)
    ok = np.isfinite(rad_f) & np.isfinite(px) & np.isfinite(py)
    visible &= ok & (touched > 0)
    touched = np.where(visible, touched, 0).astype(np.uint32)
    radii = np.where(visible, np.where(ok, rad_f, 0), 0).astype(np.int32)
    return Geometry(visible=visible, radii=radii, depth=tz.astype(np.float32),
                    px=px.astype(np.float32), py=py.astype(np.float32),
                    rect_min=np.stack([rminx, rminy], 1), rect_max=np.stack([rmaxx, rmaxy], 1),
                    tiles_touched=touched, conic=conic, cov3d=c3, grid=(gx, gy))


@dataclass
class Binning:
    keys_sorted: np.ndarray    # [L] uint64
    point_list: np.ndarray     # [L] int32 gaussian id per sorted instance
    ranges: np.ndarray         # [T,2] int32
    num_rendered: int


def binning_spec(geom: Geometry) -> Binning:
    """duplicateWithKeys + stable sort by (tile|depth) + identifyTileRanges."""
    gx, gy = geom.grid
    cnt = geom.tiles_touched.astype(np.int64)
    L = int(cnt.sum())
    excl = np.cumsum(cnt) - cnt
    ids = np.repeat(np.arange(cnt.shape[0], dtype=np.int64), cnt)          # emission order: ascending id
    k = np.arange(L, dtype=np.int64) - np.repeat(excl, cnt)                 # then row-major inside the rect
    w = (geom.rect_max[:, 0] - geom.rect_min[:, 0]).astype(np.int64)[ids]
    ty = geom.rect_min[ids, 1].astype(np.int64) + k // np.maximum(w, 1)
    tx = geom.rect_min[ids, 0].astype(np.int64) + k % np.maximum(w, 1)
    depth_bits = geom.depth.view(np.uint32).astype(np.uint64)
    keys = ((ty * gx + tx).astype(np.uint64) << np.uint64(32)) | depth_bits[ids]
    ids = ids.astype(np.int32)
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    pl = ids[order]
    T = gx * gy
    ranges = np.zeros((T, 2), dtype=np.int32)
    if L:
        tiles = (ks >> np.uint64(32)).astype(np.int64)
        starts = np.searchsorted(tiles, np.arange(T), side="left")
        ends = np.searchsorted(tiles, np.arange(T), side="right")
        nz = ends > starts
        ranges[nz, 0] = starts[nz]
        ranges[nz, 1] = ends[nz]
    return Binning(keys_sorted=ks, point_list=pl, ranges=ranges, num_rendered=L)


# --------------------------------------------------------------------------
# differentiable blend (torch)
# --------------------------------------------------------------------------
def eval_sh_torch(deg, sh, dirs):
    """sh: [P,M,3]; dirs: [P,3] unit.  Same polynomial as utils/sh_utils.py:57-112."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def lod_opacity(opacity, interp_w, kids):
    """Hierarchy-mode opacity remap (DESIGN.md 'LOD opacity').

    A node in transition stands in for one of ``k`` siblings that at w=0 all
    coincide with their parent:
        o' = w*o + (1-w) * (1 - (1 - min(o, 0.99))^(1/k))      for k >= 2
    and o' = o for k < 2.  Identity when the LOD tensors are empty.

    What the remap does and does not guarantee: ``k`` stacked copies of opacity ``1-(1-o)^(1/k)`` leave the
    transmittance of ONE copy of opacity ``o`` only where the Gaussian's falloff G is 1 (its centre).  The blended
    quantity is alpha = o*G, and off-centre ``1-(1-o'G)^k != o*G`` (o = 0.9, k = 2, G = 0.5: 0.567 against 0.45) --
    the k children are MORE opaque than their parent there.  The exact invariant belongs to a per-PIXEL remap of
    alpha (``lod_alpha`` below, ``rasterize(..., lod_mode="alpha")``, oracle only).  Which of the two the upstream
    kernel applies cannot be seen from /root/reference (gaussian_renderer/__init__.py:258-265 only passes the two
    tensors on); tests/test_oracle_kat.py::test_lod_remap_parent_vs_children measures both against the rationale, and
    the pin kit's ``upstream_raster_post.npz`` discriminates between them on arrival.
    """
    k = kids.to(opacity.dtype).clamp_min(1.0)
    oc = opacity.clamp(max=ALPHA_MAX)
    stacked = 1.0 - torch.pow(1.0 - oc, 1.0 / k)
    out = interp_w * opacity + (1.0 - interp_w) * stacked
    return torch.where(kids >= 2, out, opacity)


def lod_alpha(alpha_raw, interp_w, kids):
    """ORACLE-ONLY alternative reading of the two hierarchy tensors (``lod_mode="alpha"``): the remap of
    ``lod_opacity`` applied per pixel to alpha = o*G instead of per Gaussian to o.  With it k coincident children at
    w = 0 composite EXACTLY like their parent at every pixel ((1 - a')^k = 1 - a).  No kernel implements it."""
    k = kids.to(alpha_raw.dtype).clamp_min(1.0)
    ac = alpha_raw.clamp(max=ALPHA_MAX)
    stacked = 1.0 - torch.pow(1.0 - ac, 1.0 / k)
    out = interp_w * alpha_raw + (1.0 - interp_w) * stacked
    return torch.where(kids >= 2, out, alpha_raw)


def _cov3d_torch(scales, rotations, scale_modifier):
    s = scale_modifier * scales
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    Lm = R * s[:, None, :]
    return Lm @ Lm.transpose(1, 2)


@dataclass
class OracleOut:
    color: torch.Tensor        # [3,H,W]
    radii: torch.Tensor        # [P] int32
    invdepth: torch.Tensor     # [1,H,W]
    fragile: np.ndarray        # [H,W] bool: a discrete blend decision sat within tolerance of its threshold
    geom: Geometry
    binning: Binning
    n_contrib: np.ndarray      # [H,W] int32 index (1-based, in the tile list) of the last blended Gaussian
    final_T: np.ndarray        # [H,W] float


def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
              *, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
              projmatrix, sh_degree, campos, interpolation_weights=None, num_node_kids=None,
              dtype=torch.float64, tiles=None, fragile_tol=1e-5, geom_dtype=None, lod_mode="opacity",
              positive_power="skip") -> OracleOut:
    """Dense per-tile oracle.  All tensor arguments are CPU torch tensors; the
    differentiable ones may require grad.  ``tiles``: optional iterable of tile
    ids to restrict the blend to (bench cpu_baseline sampling); other pixels
    are left at zero.  ``geom_dtype`` (default: ``dtype``): precision of the per-Gaussian
    continuous stage (projection, conic, colour); the blend runs in ``dtype`` on tile-relative
    coordinates -- ``dtype=float32, geom_dtype=float64`` mirrors the HIP kernels' precision split.
    ``lod_mode``: "opacity" = ``lod_opacity`` per Gaussian (what the kernels do); "alpha" = ``lod_alpha`` per pixel
    (oracle-only alternative, see ``lod_opacity``)."""
    if lod_mode not in ("opacity", "alpha"):
        raise ValueError(f"lod_mode {lod_mode!r}")
    H, W = int(image_height), int(image_width)
    # the op receives tanfov / scale_modifier as C floats (GaussianRasterizationSettings -> float32)
    tanfovx, tanfovy = float(np.float32(tanfovx)), float(np.float32(tanfovy))
    scale_modifier = float(np.float32(scale_modifier))
    npf = lambda t: None if t is None else t.detach().to(torch.float32).cpu().numpy()
    geom = geometry_spec(npf(means3D), npf(scales), npf(rotations), npf(cov3D_precomp),
                         npf(viewmatrix), npf(projmatrix), W, H, tanfovx, tanfovy, scale_modifier)
    binning = binning_spec(geom)
    gx, gy = geom.grid
    P = means3D.shape[0]

    # Continuous quantities are computed for the visible subset only (culled rows
    # would put inf/nan into the autograd graph).
    vis_np = np.nonzero(geom.visible)[0]
    vidx = torch.from_numpy(vis_np.astype(np.int64))
    remap = np.full(P, -1, dtype=np.int64)
    remap[vis_np] = np.arange(vis_np.shape[0])
    V = vidx.shape[0]
    gdt = dtype if geom_dtype is None else geom_dtype
    blend_dtype = dtype
    dtype = gdt
    cv = lambda t: None if t is None else t.to(dtype)
    sel = lambda t: None if t is None else cv(t)[vidx]
    p = sel(means3D)
    vm = cv(viewmatrix.detach())
    pm = cv(projmatrix.detach())
    cam = cv(campos.detach())
    bgc = cv(bg.detach())
    ones = torch.ones(V, 1, dtype=dtype)
    ph = torch.cat([p, ones], 1)
    pv = ph @ vm                       # row-vector convention
    phom = ph @ pm
    pw = 1.0 / (phom[:, 3] + 1e-7)
    ndc = phom[:, :2] * pw[:, None]
    tz = pv[:, 2]

    if cov3D_precomp is not None:
        c = sel(cov3D_precomp)
        Sig = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4],
                           c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    else:
        Sig = _cov3d_torch(sel(scales), sel(rotations), float(scale_modifier))
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txc = torch.clamp(pv[:, 0] / tz, -limx, limx) * tz
    tyc = torch.clamp(pv[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * txc) / (tz * tz),
                     zero, fy / tz, -(fy * tyc) / (tz * tz)], 1).reshape(-1, 2, 3)
    Wm = vm[:3, :3].t()                # standard-orientation view rotation
    Tm = J @ Wm
    cov2 = Tm @ Sig @ Tm.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    cc = cov2[:, 1, 1] + 0.3
    det = a * cc - b * b
    A, B, C = cc / det, -b / det, a / det

    if means2D is not None:
        m2 = sel(means2D)
    else:
        m2 = torch.zeros(V, 3, dtype=dtype)
    gxp = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5 + m2[:, 0] * (0.5 * W)
    gyp = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5 + m2[:, 1] * (0.5 * H)

    if colors_precomp is not None:
        rgb = sel(colors_precomp)
    else:
        d = p - cam[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh_torch(int(sh_degree), sel(shs), d) + 0.5, 0.0)
    opac = sel(opacities).reshape(-1)
    lod_w = lod_k = None
    if interpolation_weights is not None and interpolation_weights.numel() > 0:
        lod_w = cv(interpolation_weights.detach()).reshape(-1)[:P][vidx]
        lod_k = num_node_kids.detach().reshape(-1)[:P][vidx]
        if lod_mode == "opacity":
            opac = lod_opacity(opac, lod_w, lod_k)
            lod_w = lod_k = None
    invz = 1.0 / tz
    # hand the per-Gaussian values to the blend precision (pixel centres stay in geom precision
    # until they have been made tile-relative)
    dtype = blend_dtype
    A, B, C, rgb, opac, invz, bgc = (t.to(dtype) for t in (A, B, C, rgb, opac, invz, bgc))

    color = torch.zeros(3, H, W, dtype=dtype)
    invd = torch.zeros(1, H, W, dtype=dtype)
    fragile = np.zeros((H, W), dtype=bool)
    n_contrib = np.zeros((H, W), dtype=np.int32)
    final_T = np.ones((H, W), dtype=np.float64)
    tile_iter = range(gx * gy) if tiles is None else tiles
    col_tiles, dep_tiles, coords = [], [], []
    for t in tile_iter:
        s, e = binning.ranges[t]
        ty0, tx0 = (t // gx) * TILE, (t % gx) * TILE
        ys = torch.arange(ty0, min(ty0 + TILE, H))
        xs = torch.arange(tx0, min(tx0 + TILE, W))
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        pxs = xx.reshape(-1).to(dtype)
        pys = yy.reshape(-1).to(dtype)
        npx = pxs.shape[0]
        if e <= s:
            ct = bgc[:, None].expand(3, npx)
            dt = torch.zeros(npx, dtype=dtype)
        else:
            ids = torch.from_numpy(remap[binning.point_list[s:e]])
            dx = (gxp[ids] - tx0).to(dtype)[None, :] - (pxs - tx0)[:, None]
            dy = (gyp[ids] - ty0).to(dtype)[None, :] - (pys - ty0)[:, None]
            power = -0.5 * (A[ids][None] * dx * dx + C[ids][None] * dy * dy) - B[ids][None] * dx * dy
            if positive_power == "clamp":      # what the HIP kernels do; "skip": App. A.8 literally.  The
                power = torch.clamp(power, max=0.0)   # two differ only where ROUNDING makes the exponent positive (needles, float32)
            G = torch.exp(power)
            araw = opac[ids][None] * G
            if lod_w is not None:
                araw = lod_alpha(araw, lod_w[ids][None].to(dtype), lod_k[ids][None])
            alpha = araw + (torch.clamp(araw, max=ALPHA_MAX) - araw).detach()   # straight-through cap
            live = (power <= 0) & (alpha >= ALPHA_MIN)
            a_eff = torch.where(live, alpha, torch.zeros_like(alpha))
            with torch.no_grad():
                T_incl = torch.cumprod(1.0 - a_eff, dim=1)
                stop = live & (T_incl < T_EPS)
                dead = torch.cumsum(stop.to(torch.int32), dim=1) > 0
                keep = live & ~dead
                # The band around a threshold inside which a float32 evaluation may decide differently GROWS WITH THE
                # FOOTPRINT: the exponent is a sum of three products that cancel (|A dx^2|, |C dy^2|, |B dx dy| reach
                # thousands for a Gaussian hundreds of pixels wide while their sum stays at -5.5), so alpha carries a
                # relative error of about 2^-24 per unit of M = the sum of their magnitudes, times the handful of
                # roundings in the chain (FRAGILE_FP32_K); a fixed 1e-5 band missed flips at 1/255 on the trained 1080p
                # scene of round 5 (3 of 49 128 values, profiles/r05_config2_config3_scripts.log).  The transmittance
                # inherits the error of every alpha blended before it.
                M = 0.5 * (A[ids][None].abs() * dx * dx + C[ids][None].abs() * dy * dy) + (B[ids][None] * dx * dy).abs()
                band = fragile_tol + FRAGILE_FP32_K * M
                frag = ((alpha - ALPHA_MIN).abs() < band * ALPHA_MIN) & (power <= 0) & ~dead
                frag |= (power.abs() < 1e-12) & ~dead
                # (an alpha AT the 0.99 cap is a constant: it carries no rounding error into 1 - alpha)
                noise = torch.where(araw < ALPHA_MAX, a_eff / (1.0 - a_eff).clamp_min(1e-2) * (FRAGILE_FP32_K * M),
                                    torch.zeros_like(M))
                band_T = fragile_tol + torch.cumsum(noise, dim=1)
                frag |= ((T_incl - T_EPS).abs() < band_T * T_EPS) & live & \
                        (torch.cumsum(stop.to(torch.int32), dim=1) <= 1)
                fr = frag.any(dim=1)
                idx = torch.arange(1, e - s + 1)[None, :].expand_as(keep)
                nc = torch.where(keep, idx, torch.zeros_like(idx)).max(dim=1).values
            a_fin = torch.where(keep, alpha, torch.zeros_like(alpha))
            om = 1.0 - a_fin
            T_after = torch.cumprod(om, dim=1)
            T_before = torch.cat([torch.ones(npx, 1, dtype=dtype), T_after[:, :-1]], 1)
            w = a_fin * T_before
            ct = (w @ rgb[ids]).t() + T_after[:, -1][None, :] * bgc[:, None]
            dt = w @ invz[ids]
            yy_n, xx_n = yy.reshape(-1).numpy(), xx.reshape(-1).numpy()
            fragile[yy_n, xx_n] = fr.numpy()
            n_contrib[yy_n, xx_n] = nc.numpy().astype(np.int32)
            final_T[yy_n, xx_n] = T_after[:, -1].detach().numpy()
        col_tiles.append(ct)
        dep_tiles.append(dt)
        coords.append((yy.reshape(-1), xx.reshape(-1)))
    if col_tiles:
        ally = torch.cat([c[0] for c in coords])
        allx = torch.cat([c[1] for c in coords])
        flat = ally * W + allx
        color = color.reshape(3, H * W).index_copy(1, flat, torch.cat(col_tiles, 1)).reshape(3, H, W)
        invd = invd.reshape(1, H * W).index_copy(1, flat, torch.cat(dep_tiles)[None, :]).reshape(1, H, W)
    return OracleOut(color=color, radii=torch.from_numpy(geom.radii.copy()), invdepth=invd,
                     fragile=fragile, geom=geom, binning=binning, n_contrib=n_contrib, final_T=final_T)


def naive_per_pixel_blend(gxp, gyp, conic, opac, rgb, invz, point_list, ranges, W, H, bg):
    """Literal per-pixel front-to-back loop, pure Python, float64.
    Only for tiny cases: cross-checks the vectorised blend above."""
    gx = (W + TILE - 1) // TILE
    out = np.zeros((3, H, W))
    dep = np.zeros((H, W))
    for y in range(H):
        for x in range(W):
            t = (y // TILE) * gx + (x // TILE)
            s, e = ranges[t]
            T = 1.0
            C = np.zeros(3)
            D = 0.0
            for i in range(s, e):
                g = point_list[i]
                dx = gxp[g] - x
                dy = gyp[g] - y
                power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
                if power > 0:
                    continue
                alpha = min(ALPHA_MAX, opac[g] * math.exp(power))
                if alpha < ALPHA_MIN:
                    continue
                Tn = T * (1 - alpha)
                if Tn < T_EPS:
                    break
                C += rgb[g] * alpha * T
                D += invz[g] * alpha * T
                T = Tn
            out[:, y, x] = C + T * bg
            dep[y, x] = D
    return out, dep


def activate_raw(scaling_raw, rotation_raw, opacity_raw, opacity_activation="sigmoid"):
    """Specification of the raw-parameter path (include/hgs.h HGS_ACT_*): the activations of
    scene/gaussian_model.py:108-128 evaluated in float64 and rounded ONCE to float32.  Returns float64 tensors whose
    VALUES are those float32 numbers (so the discrete float32 spec sees exactly them) and whose autograd graph is
    the exact activation (straight-through over the rounding)."""
    def rounded(t):
        return t + (t.detach().float().double() - t.detach())
    s = rounded(torch.exp(scaling_raw.double()))
    q = rotation_raw.double()
    n = torch.sqrt(((q[:, 0] ** 2 + q[:, 1] ** 2) + q[:, 2] ** 2) + q[:, 3] ** 2).clamp_min(1e-12)
    r = rounded(q / n[:, None])
    x = opacity_raw.double()
    if opacity_activation == "sigmoid":
        o = rounded(1.0 / (1.0 + torch.exp(-x)))
    elif opacity_activation == "abs":
        o = x.abs()
    else:
        o = x
    return s, r, o
