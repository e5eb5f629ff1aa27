"""The PRODUCT's per-Gaussian math header (csrc/gaussian_math.h) compiled for the HOST and held against the oracle --
no GPU.  The header's float32 chain (cull, radius, tile rectangle, depth key, EWA clamp flags: everything that feeds a
discrete decision) is written with contraction off and explicit parentheses, so g++ on x86-64 reproduces the device's
results bit for bit; tests/harness/host_math/common.h supplies the qualifiers as no-ops and the header itself is copied
unchanged at test time.  What `-m gpu` checks through the kernels (`test_scale_parity_gpu.py`: radii / rectangles /
depth bits of whole frames) is checked here on the same arithmetic, so an edit of the header that breaks the
specification is caught before it reaches a GPU box."""
import ctypes as C
import math
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
from hgs import synth                          # noqa: E402
from oracle import raster_oracle as ro         # noqa: E402

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
HARNESS = os.path.join(ROOT, "tests", "harness", "host_math")


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    d = tmp_path_factory.mktemp("host_math")
    for f in (os.path.join(HARNESS, "common.h"), os.path.join(HARNESS, "host_math.cpp"),
              os.path.join(ROOT, "hierarchical-3d-gaussians_amd", "csrc", "gaussian_math.h")):
        shutil.copy(f, d)
    so = os.path.join(d, "libhostmath.so")
    r = subprocess.run(["g++", "-O2", "-ffp-contract=off", "-w", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                        "-I", str(d), os.path.join(d, "host_math.cpp"), "-o", so], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return C.CDLL(so)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def geometry(host, scene, cam, W, H, mod=1.0, cov3d=None):
    f32 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
    means, scales, rots = f32(scene.means3D), f32(scene.scales), f32(scene.rotations)
    vm, pm = f32(cam.world_view_transform).reshape(16), f32(cam.full_proj_transform).reshape(16)
    P = means.shape[0]
    out = dict(visible=np.zeros(P, np.uint8), radii=np.zeros(P, np.int32), depth=np.zeros(P, np.float32),
               rect=np.zeros((P, 4), np.int32), touched=np.zeros(P, np.uint32), conic=np.zeros((P, 3), np.float32),
               cov3d=np.zeros((P, 6), np.float32), pxpy=np.zeros((P, 2), np.float32), clamp=np.zeros(P, np.uint8),
               cont=np.zeros((P, 9), np.float64))
    host.host_geometry(_ptr(means), _ptr(scales) if cov3d is None else None, _ptr(rots) if cov3d is None else None,
                       _ptr(cov3d), _ptr(vm), _ptr(pm), W, H, C.c_float(cam.tanfovx), C.c_float(cam.tanfovy),
                       C.c_float(mod), P, *[_ptr(out[k]) for k in ("visible", "radii", "depth", "rect", "touched", "conic",
                                                                    "cov3d", "pxpy", "clamp", "cont")])
    spec = ro.geometry_spec(means, scales, rots, cov3d, vm.reshape(4, 4), pm.reshape(4, 4), W, H, cam.tanfovx,
                            cam.tanfovy, mod)
    return out, spec, (means, scales, rots, vm, pm)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


SCENES = {
    "metric": lambda cam: synth.make_scene(60_000, cam, seed=0),
    "heavy": lambda cam: synth.make_scene(40_000, cam, seed=1, s_px=(1.0, 8.0)),
    "close and wide": lambda cam: synth.make_scene(40_000, cam, seed=2, s_px=(0.05, 150.0), z_range=(0.15, 40.0)),
    "trained-like": lambda cam: synth.make_scene_trained_like(40_000, cam, seed=3),
}


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("size", [(1920, 1080), (637, 349)])
def test_discrete_geometry_of_the_header_equals_the_oracle_bit_for_bit(host, name, size):
    W, H = size
    cam = synth.make_camera(W, H, 75.0 if W < 1000 else 60.0)
    out, spec, _ = geometry(host, SCENES[name](cam), cam, W, H, mod=1.0 if name != "heavy" else 0.7)
    vis = out["visible"].astype(bool)
    assert 0.2 < vis.mean() <= 1.0
    assert np.array_equal(vis, spec.visible)
    assert np.array_equal(out["radii"], spec.radii)
    assert np.array_equal(bits(out["depth"]), bits(spec.depth))                    # the sort key's low word
    assert np.array_equal(out["touched"], spec.tiles_touched)
    assert np.array_equal(out["rect"][vis][:, :2], spec.rect_min[vis]) and np.array_equal(out["rect"][vis][:, 2:], spec.rect_max[vis])
    assert np.array_equal(bits(out["cov3d"]), bits(spec.cov3d))
    assert np.array_equal(bits(out["conic"][vis]), bits(spec.conic[vis]))
    assert np.array_equal(bits(out["pxpy"][vis, 0]), bits(spec.px[vis])) and np.array_equal(bits(out["pxpy"][vis, 1]), bits(spec.py[vis]))
    assert int(out["touched"].sum()) > 0


def test_precomputed_covariance_route(host):
    W, H = 800, 600
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(20_000, cam, seed=5)
    _, spec0, (means, scales, rots, vm, pm) = geometry(host, scene, cam, W, H)
    out, spec, _ = geometry(host, scene, cam, W, H, cov3d=np.ascontiguousarray(spec0.cov3d))
    assert np.array_equal(out["radii"], spec.radii) and np.array_equal(out["radii"], spec0.radii)
    assert np.array_equal(out["touched"], spec.tiles_touched)


def test_the_double_twin_agrees_with_a_float64_restatement(host):
    """pixel centre, conic, 2D covariance and 1 / z of project_gaussian_d against the same formulas in numpy float64
    (tests/tools/k8a_float_chain_study.py restates them): 1e-12 relative -- the device's reciprocal is a float seed + one
    Newton step (rcp_d), not an IEEE division."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import k8a_float_chain_study as st
    W, H = 1920, 1080
    cam = synth.make_camera(W, H)
    out, spec, (means, scales, rots, vm, pm) = geometry(host, synth.make_scene_trained_like(30_000, cam, seed=4), cam, W, H)
    pd = st.forward_double(means.astype(np.float64), scales.astype(np.float64), rots.astype(np.float64),
                           vm.astype(np.float64), pm.astype(np.float64), W, H, float(np.float32(cam.tanfovx)),
                           float(np.float32(cam.tanfovy)))
    vis = out["visible"].astype(bool)
    # the double twin takes its clamp decisions from the float32 chain: compare where both agree on them
    same = vis & (pd["clampx"] == ((out["clamp"] & 1) != 0)) & (pd["clampy"] == ((out["clamp"] & 2) != 0))
    assert same.sum() > 0.95 * vis.sum()
    px = ((pd["hx"] * pd["pw"] + 1.0) * W - 1.0) * 0.5
    py = ((pd["hy"] * pd["pw"] + 1.0) * H - 1.0) * 0.5
    ref = np.stack([px, py, pd["conA"], pd["conB"], pd["conC"], pd["a"], pd["b"], pd["c"], pd["itz"]], axis=1)[same]
    got = out["cont"][same]
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
    # (px, py: absolute 1e-9 px; the rest relative)
    assert np.abs(got[:, :2] - ref[:, :2]).max() < 1e-9
    assert rel[:, 2:].max() < 1e-11, rel[:, 2:].max(axis=0)


def test_sh_basis_and_its_gradient(host):
    """sh_basis against the reference's own table (utils/sh_utils.py via the oracle's eval), its gradient against
    central differences of the basis itself."""
    rng = np.random.default_rng(0)
    d = rng.standard_normal((2000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    b, dbx, dby, dbz = (np.zeros((2000, 16), np.float32) for _ in range(4))
    host.host_sh_basis(3, _ptr(d), 2000, _ptr(b), _ptr(dbx), _ptr(dby), _ptr(dbz))
    for deg in (0, 1, 2, 3):          # sum_k b_k sh_k against the oracle's evaluation of the reference's polynomial
        bd = np.zeros((2000, 16), np.float32)
        z = [np.zeros((2000, 16), np.float32) for _ in range(3)]
        host.host_sh_basis(deg, _ptr(d), 2000, _ptr(bd), _ptr(z[0]), _ptr(z[1]), _ptr(z[2]))
        sh = torch.from_numpy(rng.standard_normal((2000, 16, 3)))
        ref = ro.eval_sh_torch(deg, sh, torch.from_numpy(d.astype(np.float64))).numpy()
        got = np.einsum("nk,nkc->nc", bd.astype(np.float64), sh.numpy())
        assert np.abs(got - ref).max() < 2e-5, deg
        assert not bd[:, (deg + 1) ** 2:].any()          # nothing beyond the active degree
    eps = 1e-3

    def basis(dd):
        o = np.zeros((dd.shape[0], 16), np.float32)
        z = np.zeros_like(o)
        dd = np.ascontiguousarray(dd, dtype=np.float32)
        host.host_sh_basis(3, _ptr(dd), dd.shape[0], _ptr(o), _ptr(z.copy()), _ptr(z.copy()), _ptr(z.copy()))
        return o.astype(np.float64)

    for axis, g in enumerate((dbx, dby, dbz)):
        e = np.zeros(3, np.float32); e[axis] = eps
        fd = (basis(d + e) - basis(d - e)) / (2 * eps)
        assert np.abs(g - fd).max() < 2e-3, axis


def test_lod_opacity_of_the_header_equals_the_oracle(host):
    host.host_lod_opacity.restype = C.c_float
    for o in (0.05, 0.3, 0.9, 1.0, 1.3):
        for w in (0.0, 0.25, 1.0):
            for k in (1, 2, 4, 8):
                dd = C.c_float(0)
                got = host.host_lod_opacity(C.c_float(o), C.c_float(w), k, C.byref(dd))
                ref = float(ro.lod_opacity(torch.tensor([o], dtype=torch.float64), torch.tensor([w], dtype=torch.float64),
                                           torch.tensor([k], dtype=torch.int32))[0])
                assert abs(got - ref) <= 2e-6 * max(1.0, abs(ref)), (o, w, k, got, ref)
