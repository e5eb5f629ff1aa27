"""Pins the oracle (and the host-side camera helper) to golden vectors generated from the REFERENCE'S OWN
Python utilities (tests/golden/make_golden.py imports them from /root/reference):
utils/sh_utils.py eval_sh, utils/general_utils.py build_rotation / build_scaling_rotation /
strip_symmetric, utils/graphics_utils.py getWorld2View2 / getProjectionMatrix / geom_transform_points.
These are the only numerical pins the reference offers for this path (it has no tests)."""
import os

import numpy as np
import torch

from hgs import synth
from oracle import raster_oracle as ro

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_utils_golden.npz"))


def test_sh_polynomial_matches_reference_eval_sh():
    sh = torch.from_numpy(G["sh_coeffs"]).permute(0, 2, 1).contiguous()      # op layout [P,M,3]
    dirs = torch.from_numpy(G["sh_dirs"])
    for deg in range(4):
        got = ro.eval_sh_torch(deg, sh.double(), dirs.double())
        assert np.allclose(got.numpy(), G[f"sh_eval_deg{deg}"], rtol=0, atol=2e-6), deg
    assert np.allclose((G["rgb"] - 0.5) / ro.SH_C0, G["rgb2sh"], atol=1e-6)


def test_covariance_matches_reference_build_scaling_rotation():
    q = G["quat"]
    qn = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)   # caller normalises (gaussian_model.py:113)
    R = ro._quat_to_rot_f32(qn)
    assert np.allclose(R, G["rotmat"], atol=2e-6)
    cov = ro.cov3d_spec(G["scale"], qn, 1.7)
    assert np.allclose(cov, G["cov6_mod1p7"], rtol=2e-5, atol=2e-6)
    cov_t = ro._cov3d_torch(torch.from_numpy(G["scale"]).double(), torch.from_numpy(qn).double(), 1.7)
    six = torch.stack([cov_t[:, 0, 0], cov_t[:, 0, 1], cov_t[:, 0, 2], cov_t[:, 1, 1], cov_t[:, 1, 2], cov_t[:, 2, 2]], 1)
    assert np.allclose(six.numpy(), G["cov6_mod1p7"], rtol=2e-5, atol=2e-6)


def test_camera_helper_matches_reference_matrices():
    fovx, fovy = G["cam_fov"]
    proj = synth.projection_matrix(0.01, 100.0, float(fovx), float(fovy)).transpose(0, 1)
    assert np.allclose(proj.numpy(), G["projection_matrix"], atol=1e-7)
    # make_camera derives FoVx from square pixels; rebuild the same camera through its pieces
    Rt = np.zeros((4, 4)); Rt[:3, :3] = G["cam_R"].T; Rt[:3, 3] = G["cam_T"]; Rt[3, 3] = 1
    wv = torch.tensor(np.float32(Rt)).transpose(0, 1)
    assert np.allclose(wv.numpy(), G["world_view_transform"], atol=1e-6)
    full = wv @ proj
    assert np.allclose(full.numpy(), G["full_proj_transform"], atol=1e-6)
    assert np.allclose(wv.inverse()[3, :3].numpy(), G["camera_center"], atol=1e-6)
    cam = synth.make_camera(1920, 1080, 60.0)
    assert np.allclose((cam.world_view_transform @ torch.from_numpy(G["proj_1080p_fovy60"])).numpy(),
                       cam.full_proj_transform.numpy(), atol=1e-6)
    assert abs(cam.tanfovx - 1.0264005) < 1e-6 and abs(cam.tanfovy - 0.5773503) < 1e-6


def test_projection_convention_matches_geom_transform_points():
    """Row-vector convention: p_hom = [p,1] @ full_proj; the oracle's flattened-matrix reads must agree."""
    full = G["full_proj_transform"].astype(np.float32).reshape(16)
    pts = G["points"].astype(np.float32)
    hx = ro._xform3(full, pts[:, 0], pts[:, 1], pts[:, 2], 0)
    hy = ro._xform3(full, pts[:, 0], pts[:, 1], pts[:, 2], 1)
    hw = ro._xform3(full, pts[:, 0], pts[:, 1], pts[:, 2], 3)
    ndc = np.stack([hx / (hw + 1e-7), hy / (hw + 1e-7)], 1)
    assert np.allclose(ndc, G["points_ndc"][:, :2], rtol=1e-5, atol=1e-6)
