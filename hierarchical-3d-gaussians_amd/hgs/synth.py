"""Seeded synthetic workloads for the rasterizer hot path.

Cameras follow the reference's conventions (scene/cameras.py:89-98,
utils/graphics_utils.py:38-77): matrices are stored transposed (row-vector
convention), ``full_proj = world_view @ projection``, +z forward, y down,
znear 0.01 / zfar 100.  Everything is generated on the CPU from a
``torch.Generator`` so the oracle and the GPU see identical bits.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class Camera:
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # [4,4] stored (transposed) form
    full_proj_transform: torch.Tensor    # [4,4]
    camera_center: torch.Tensor          # [3]
    znear: float = 0.01
    zfar: float = 100.0

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)

    def to(self, device):
        return Camera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                      self.world_view_transform.to(device), self.full_proj_transform.to(device),
                      self.camera_center.to(device), self.znear, self.zfar)


def projection_matrix(znear, zfar, fovx, fovy, primx=0.5, primy=0.5):
    """Standard-orientation perspective matrix with P[3,2]=1 (z forward);
    same contract as utils/graphics_utils.py:51-77."""
    ty = math.tan(fovy / 2) * znear
    tx = math.tan(fovx / 2) * znear
    top, bottom = primy * 2 * ty, (1 - primy) * 2 * -ty
    right, left = primx * 2 * tx, (1 - primx) * 2 * -tx
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def make_camera(width, height, fovy_deg=60.0, R=None, T=None, znear=0.01, zfar=100.0) -> Camera:
    """R: camera-to-world rotation (3x3, as scene/dataset_readers.py:89 stores it), T: world-to-camera
    translation.  Defaults: camera at the origin looking down +z."""
    fovy = math.radians(fovy_deg)
    fy = height / (2.0 * math.tan(fovy / 2))
    fovx = 2.0 * math.atan(width / (2.0 * fy))          # square pixels
    R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64)
    T = np.zeros(3) if T is None else np.asarray(T, dtype=np.float64)
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.T
    Rt[:3, 3] = T
    Rt[3, 3] = 1.0
    wv = torch.tensor(np.float32(Rt)).transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wv.inverse()[3, :3].contiguous()
    return Camera(width, height, fovx, fovy, wv, full, center, znear, zfar)


def orbit_camera(width, height, k, n, radius=0.6, fovy_deg=60.0, tilt=0.05) -> Camera:
    """k-th of n cameras on a small circle around the origin, all looking roughly down +z
    (per-view data-parallel workloads)."""
    ang = 2 * math.pi * k / max(n, 1)
    c = np.array([radius * math.cos(ang), radius * math.sin(ang), 0.0])
    yaw = tilt * math.cos(ang)
    pitch = tilt * math.sin(ang)
    Ry = np.array([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(pitch), -math.sin(pitch)], [0, math.sin(pitch), math.cos(pitch)]])
    Rc2w = Ry @ Rx
    T = -Rc2w.T @ c
    return make_camera(width, height, fovy_deg, R=Rc2w, T=T)


@dataclass
class Scene:
    means3D: torch.Tensor      # [P,3]
    scales: torch.Tensor       # [P,3] activated (exp)
    rotations: torch.Tensor    # [P,4] normalised (w,x,y,z)
    opacities: torch.Tensor    # [P,1] activated (sigmoid)
    shs: torch.Tensor          # [P,M,3]
    sh_degree: int

    def to(self, device):
        return Scene(*(t.to(device) for t in (self.means3D, self.scales, self.rotations,
                                               self.opacities, self.shs)), self.sh_degree)

    @property
    def P(self):
        return self.means3D.shape[0]


def make_scene(P, cam: Camera, seed=0, sh_degree=3, s_px=(0.5, 4.0), z_range=(2.0, 20.0)) -> Scene:
    """Frustum-filling Gaussians exactly as specified in SURVEY.md §8(d) / BASELINE.md §3."""
    g = torch.Generator().manual_seed(seed)
    U = lambda *s: torch.rand(*s, generator=g)
    N = lambda *s: torch.randn(*s, generator=g)
    z = z_range[0] + (z_range[1] - z_range[0]) * U(P)
    x = z * cam.tanfovx * (2 * U(P) - 1)
    y = z * cam.tanfovy * (2 * U(P) - 1)
    means = torch.stack([x, y, z], 1)
    fx = cam.image_width / (2.0 * cam.tanfovx)
    spx = torch.exp(math.log(s_px[0]) + (math.log(s_px[1]) - math.log(s_px[0])) * U(P))
    scales = (z * spx / fx)[:, None] * (0.3 + 0.7 * U(P, 3))
    q = N(P, 4)
    q = q / q.norm(dim=1, keepdim=True)
    opac = (0.05 + 0.9 * U(P))[:, None]
    M = (sh_degree + 1) ** 2
    shs = torch.empty(P, M, 3)
    shs[:, 0] = 0.5 * N(P, 3)
    if M > 1:
        shs[:, 1:] = 0.05 * N(P, M - 1, 3)
    # camera-space -> world (camera may not sit at the origin)
    wv = cam.world_view_transform.double()
    c2w = wv.inverse()
    mh = torch.cat([means.double(), torch.ones(P, 1, dtype=torch.float64)], 1) @ c2w
    return Scene(mh[:, :3].float().contiguous(), scales.contiguous(), q.contiguous(),
                 opac.contiguous(), shs.contiguous(), sh_degree)


def make_scene_trained_like(P, cam: Camera, seed=0, sh_degree=3, median_px=2.0, sigma=1.0, max_px=100.0, aniso=0.05,
                            z_range=(2.0, 20.0)) -> Scene:
    """A footprint distribution closer to an optimised scene than the §8(d) benchmark spec: screen-space sigma
    log-normal (median ``median_px``, log-sigma ``sigma``, clipped at ``max_px``: a few Gaussians cover hundreds of
    tiles), one principal axis at that size and the other two log-uniform in [aniso, 1] of it (needles and discs),
    opacities biased towards the ends of (0, 1).  Everything else as make_scene."""
    base = make_scene(P, cam, seed=seed, sh_degree=sh_degree, z_range=z_range)
    g = torch.Generator().manual_seed(seed + 7919)
    U = lambda *s: torch.rand(*s, generator=g)
    wv = cam.world_view_transform.double()
    z = (torch.cat([base.means3D.double(), torch.ones(P, 1, dtype=torch.float64)], 1) @ wv)[:, 2].float()
    fx = cam.image_width / (2.0 * cam.tanfovx)
    spx = torch.exp(math.log(median_px) + sigma * torch.randn(P, generator=g)).clamp(0.3, max_px)
    ratios = torch.exp(U(P, 3) * math.log(aniso))
    ratios[torch.arange(P), torch.randint(0, 3, (P,), generator=g)] = 1.0
    scales = (z * spx / fx)[:, None] * ratios
    u = U(P)
    opac = (0.02 + 0.97 * (0.5 - 0.5 * torch.cos(math.pi * u)))[:, None]      # mass near 0 and 1
    return Scene(base.means3D, scales.contiguous(), base.rotations, opac.contiguous(), base.shs, sh_degree)


def make_scene_trained_scale(P, cam: Camera, seed=0, sh_degree=3, median_px=3.5, sigma=1.2, max_px=300.0, aniso=0.05,
                             clustered_frac=0.5, n_clusters=10, cluster_px=100.0, order="index",
                             z_range=(2.0, 20.0)) -> Scene:
    """The statistics of a scene the reference's own scripts TRAINED at 1080p (profiles/r05_config2_config3_scripts.log:
    train_single.py, 375 k Gaussians: 28 tile instances per Gaussian, L = 10.7 M, lists of 1 300 on average and 3 155 at
    most) from a seed: make_scene_trained_like's footprints (log-normal screen sigma, needles and discs, bimodal opacity)
    at a larger median, and a NON-UNIFORM screen density -- ``clustered_frac`` of the Gaussians sit in ``n_clusters``
    blobs of ``cluster_px`` pixels (objects), the rest fills the frustum -- so that the longest tile list is well above
    twice the mean (defaults at P = 375 000, seed 0, 1920x1080: L = 10.43 M, 27.8 per Gaussian, lists of 1 278 on average
    and 3 740 at most, 1 149 Gaussians of more than 1 000 tiles).  ``order``:
      "index"      rows in generation order (a trained chunk: big and small Gaussians interleaved);
      "clustered"  what a hierarchy CUT hands to the op (train_post.py:91-119, render_hierarchy.py:58-92): the cut's big
                   (interior) nodes lie side by side -- rows sorted by footprint in blocks of 4 096, the biggest block
                   first, so that a few hundred consecutive rows emit hundreds of thousands of instances."""
    base = make_scene_trained_like(P, cam, seed=seed, sh_degree=sh_degree, median_px=median_px, sigma=sigma, max_px=max_px,
                                   aniso=aniso, z_range=z_range)
    g = torch.Generator().manual_seed(seed + 104729)
    wv = cam.world_view_transform.double()
    cs = torch.cat([base.means3D.double(), torch.ones(P, 1, dtype=torch.float64)], 1) @ wv          # camera space
    z = cs[:, 2]
    n_cl = int(P * clustered_frac)
    centres = torch.rand(n_clusters, 2, generator=g, dtype=torch.float64) * 1.6 - 0.8                # NDC
    which = torch.randint(0, n_clusters, (n_cl,), generator=g)
    fx = cam.image_width / (2.0 * cam.tanfovx)
    fy = cam.image_height / (2.0 * cam.tanfovy)
    off = torch.randn(n_cl, 2, generator=g, dtype=torch.float64) * cluster_px
    ndc = centres[which] + off / torch.tensor([0.5 * cam.image_width, 0.5 * cam.image_height], dtype=torch.float64)
    sel = torch.randperm(P, generator=g)[:n_cl]
    cs[sel, 0] = ndc[:, 0] * z[sel] * cam.tanfovx
    cs[sel, 1] = ndc[:, 1] * z[sel] * cam.tanfovy
    means = (cs @ wv.inverse())[:, :3].float().contiguous()
    sc = Scene(means, base.scales, base.rotations, base.opacities, base.shs, sh_degree)
    if order == "clustered":
        size = sc.scales.max(dim=1).values / z.float()                      # screen footprint, up to the focal length
        blocks = (P + 4095) // 4096
        key = torch.argsort(size, descending=True, stable=True)
        # the biggest block first, the others in generation order behind it (their rows keep their relative order)
        big = key[:4096 if blocks > 1 else P]
        rest_mask = torch.ones(P, dtype=torch.bool); rest_mask[big] = False
        perm = torch.cat([big, torch.nonzero(rest_mask).flatten()])
        sc = Scene(*(t[perm].contiguous() for t in (sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs)), sh_degree)
    elif order != "index":
        raise ValueError(order)
    return sc


def upstream_grads(H, W, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g)
