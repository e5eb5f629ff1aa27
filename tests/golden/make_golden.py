"""Generates tests/golden/ref_utils_golden.npz by IMPORTING the reference's own Python utilities
from /root/reference (read-only) -- the only in-tree numerical pins of the rasterizer's conventions
(SURVEY.md §8(c)).  Run in the build container (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

Pinned functions:
  utils/sh_utils.py:57-112        eval_sh, RGB2SH, SH2RGB
  utils/general_utils.py:82-114   build_rotation, build_scaling_rotation (+ strip_symmetric :68-80)
  utils/graphics_utils.py:38-77   getWorld2View2, getProjectionMatrix
  scene/cameras.py:95-98          world_view_transform / full_proj_transform / camera_center assembly
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from utils import sh_utils, graphics_utils  # noqa: E402
import utils.general_utils as gu  # noqa: E402

_zeros = torch.zeros


def _cpu_zeros(*a, **k):   # the reference hard-codes device="cuda" (utils/general_utils.py:87,105)
    k.pop("device", None)
    return _zeros(*a, **k)


def main():
    g = torch.Generator().manual_seed(1234)
    out = {}
    # --- SH ---------------------------------------------------------------------------
    P = 64
    sh = torch.randn(P, 3, 16, generator=g)                 # [..., C, (deg+1)^2] as the reference views it
    dirs = torch.randn(P, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out["sh_coeffs"] = sh.numpy()
    out["sh_dirs"] = dirs.numpy()
    for deg in range(4):
        out[f"sh_eval_deg{deg}"] = sh_utils.eval_sh(deg, sh, dirs).numpy()
    rgb = torch.rand(8, 3, generator=g)
    out["rgb"] = rgb.numpy()
    out["rgb2sh"] = sh_utils.RGB2SH(rgb).numpy()
    # --- rotation / covariance ---------------------------------------------------------
    q = torch.randn(P, 4, generator=g)                       # un-normalised: build_rotation normalises
    s = torch.rand(P, 3, generator=g) + 0.05
    torch.zeros = _cpu_zeros
    try:
        R = gu.build_rotation(q)
        Lm = gu.build_scaling_rotation(1.7 * s, q)
        cov = Lm @ Lm.transpose(1, 2)
        sym = gu.strip_symmetric(cov)
    finally:
        torch.zeros = _zeros
    out["quat"] = q.numpy()
    out["scale"] = s.numpy()
    out["rotmat"] = R.numpy()
    out["cov6_mod1p7"] = sym.numpy()
    # --- camera matrices ---------------------------------------------------------------
    Rc = np.array([[0.8, -0.6, 0.0], [0.6, 0.8, 0.0], [0.0, 0.0, 1.0]])
    T = np.array([0.3, -0.2, 2.0])
    fovx, fovy = 1.1, 0.7
    wv = torch.tensor(graphics_utils.getWorld2View2(Rc, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
    proj = graphics_utils.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy, primx=0.5, primy=0.5).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    out["cam_R"], out["cam_T"] = Rc, T
    out["cam_fov"] = np.array([fovx, fovy])
    out["world_view_transform"] = wv.numpy()
    out["projection_matrix"] = proj.numpy()
    out["full_proj_transform"] = full.numpy()
    out["camera_center"] = wv.inverse()[3, :3].numpy()
    pts = torch.randn(16, 3, generator=g) + torch.tensor([0.0, 0.0, 5.0])
    out["points"] = pts.numpy()
    out["points_ndc"] = graphics_utils.geom_transform_points(pts, full).numpy()
    # identity camera at 1080p / FoVy 60deg
    fy = 1080 / (2 * math.tan(math.radians(60) / 2))
    fovx_id = 2 * math.atan(1920 / (2 * fy))
    proj_id = graphics_utils.getProjectionMatrix(0.01, 100.0, fovx_id, math.radians(60), 0.5, 0.5).transpose(0, 1)
    out["proj_1080p_fovy60"] = proj_id.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_utils_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
