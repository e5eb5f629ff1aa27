"""Runs the REFERENCE'S OWN render glue (gaussian_renderer/__init__.py: render, render_post,
render_coarse -- unmodified, imported from /root/reference) on top of this repo's
``diff_gaussian_rasterization`` package, on CPU, to prove the op surface is a drop-in: keyword
names, the 17 settings fields, empty CPU LOD tensors, 3-tuple return, autograd contract
(grads reach every parameter and ``viewspace_points``).

No GPU exists in the build container and /root/reference does not exist on the GPU box, so this is
the only place the reference's glue can be exercised.  For that purpose -- and only inside this test
-- the extension-module layer (`diff_gaussian_rasterization._C`) is swapped for an oracle-backed
stand-in and ``device="cuda"`` is mapped to the CPU.  The product package itself has no such path.
"""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gaussian_renderer")),
                                reason="reference checkout not present (GPU box)")


from harness.cpu_backends import OracleRasterC as _OracleC   # oracle-backed extension layer (test only)


@pytest.fixture()
def glue(monkeypatch):
    # (1) modules the reference imports at load time but that are absent from this image
    for name in ("plyfile", "cv2"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "plyfile":
                m.PlyData = type("PlyData", (), {})
                m.PlyElement = type("PlyElement", (), {})
            monkeypatch.setitem(sys.modules, name, m)
    # (2) "cuda" -> cpu for the factory calls the glue makes
    def remap(fn):
        def wrapped(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return wrapped
    for fname in ("zeros_like", "zeros", "empty", "range", "ones", "tensor"):
        monkeypatch.setattr(torch, fname, remap(getattr(torch, fname)))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    # (3) oracle-backed extension layer (test only)
    import diff_gaussian_rasterization as dgr
    monkeypatch.setattr(dgr, "_C", _OracleC)
    monkeypatch.syspath_prepend(REF)
    for mod in [m for m in sys.modules if m.split(".")[0] in ("gaussian_renderer", "scene", "utils", "arguments")]:
        monkeypatch.delitem(sys.modules, mod)
    import gaussian_renderer
    yield gaussian_renderer
    for mod in [m for m in sys.modules if m.split(".")[0] in ("gaussian_renderer", "scene", "utils", "arguments")]:
        sys.modules.pop(mod, None)


class _PC:
    """Duck-typed GaussianModel: only what gaussian_renderer touches (scene/gaussian_model.py:108-139)."""

    def __init__(self, scene, max_sh_degree=3, skybox_points=0):
        p = lambda t: torch.nn.Parameter(t.clone())
        self._xyz = p(scene.means3D)
        self._scaling = p(torch.log(scene.scales))
        self._rotation = p(scene.rotations)
        self._opacity = p(torch.logit(scene.opacities.clamp(1e-4, 1 - 1e-4)))
        self._features_dc = p(scene.shs[:, :1])
        self._features_rest = p(scene.shs[:, 1:])
        self.max_sh_degree = max_sh_degree
        self.active_sh_degree = max_sh_degree
        self.skybox_points = skybox_points
        self.pretrained_exposures = None
        self._exposure = torch.nn.Parameter(torch.eye(3, 4)[None])
        self.exposure_mapping = {"img0": 0}

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def get_exposure_from_name(self, name):
        return self._exposure[self.exposure_mapping[name]]

    def params(self):
        return [self._xyz, self._scaling, self._rotation, self._opacity, self._features_dc, self._features_rest]


def _viewpoint(cam):
    return types.SimpleNamespace(FoVx=cam.FoVx, FoVy=cam.FoVy, image_height=cam.image_height,
                                 image_width=cam.image_width, world_view_transform=cam.world_view_transform,
                                 full_proj_transform=cam.full_proj_transform, camera_center=cam.camera_center,
                                 image_name="img0")


def test_render_render_post_render_coarse_run_unmodified(glue):
    from hgs import synth
    cam = synth.make_camera(48, 32)
    scene = synth.make_scene(40, cam, seed=3)
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([0.1, 0.2, 0.3])

    # --- render (train_single.py:97): do_depth=True, 4 empty "cuda" LOD tensors, trained exposure
    pc = _PC(scene)
    pkg = glue.render(_viewpoint(cam), pc, pipe, bg, use_trained_exp=True)
    assert set(pkg) == {"render", "depth", "viewspace_points", "visibility_filter", "radii"}
    assert pkg["render"].shape == (3, 32, 48) and pkg["depth"].shape == (1, 32, 48)
    assert pkg["visibility_filter"].dtype == torch.int64
    (pkg["render"].sum() + pkg["depth"].sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in pc.params())
    assert pkg["viewspace_points"].grad is not None and pkg["viewspace_points"].grad.shape == (40, 3)
    assert pkg["viewspace_points"].grad[:, :2].abs().sum() > 0          # densification statistic is alive

    # every gradient the glue produces against the oracle chained (float64) through the activations it applied
    # (scene/gaussian_model.py:108-128); the twin of tests/test_reference_on_gpu.py, where the op is the HIP one
    import parity as pa
    pc = _PC(scene)
    with torch.no_grad():
        act = synth.Scene(pc.get_xyz.clone(), pc.get_scaling.clone(), pc.get_rotation.clone(), pc.get_opacity.clone(),
                          pc.get_features.clone(), 3)
    gc, gd = synth.upstream_grads(32, 48)
    oo, og = pa.run_oracle(act, cam, bg, gc, gd, mask_fragile=False)
    pkg = glue.render(_viewpoint(cam), pc, pipe, bg)
    ((pkg["render"] * gc).sum() + (pkg["depth"] * gd).sum()).backward()
    raw = dict(xyz=pc._xyz, scaling=pc._scaling, rotation=pc._rotation, opacity=pc._opacity,
               features_dc=pc._features_dc, features_rest=pc._features_rest)
    want = pa.chain_to_raw({k: v.detach() for k, v in raw.items()}, og)
    for k, v in raw.items():
        st = pa.err_stats(v.grad, want[k])
        assert st["maxrel"] <= 1e-5 and st["l2"] <= 1e-5, (k, st)

    # --- render_coarse (train_coarse.py:94): debug forced True, do_depth False
    pc = _PC(scene)
    pkg = glue.render_coarse(_viewpoint(cam), pc, pipe, bg)
    assert pkg["visibility_filter"].dtype == torch.bool and pkg["render"].shape == (3, 32, 48)

    # --- render_post (train_post.py:119-129): python-side LOD lerp, EMPTY CPU index tensors + non-empty
    #     weights / kids reach the op
    pc = _PC(scene)
    pc._opacity = torch.nn.Parameter(scene.opacities.clone())
    type(pc).get_opacity = property(lambda s: torch.abs(s._opacity))        # hierarchy mode (gaussian_model.py:393)
    n = 25
    render_indices = torch.arange(n, dtype=torch.int32)
    parent_indices = torch.zeros(40, dtype=torch.int32); parent_indices[:n] = torch.arange(n).flip(0).int()
    weights = torch.rand(40, generator=torch.Generator().manual_seed(1))
    kids = torch.full((40,), 2, dtype=torch.int32)
    pkg = glue.render_post(_viewpoint(cam), pc, pipe, bg, render_indices=render_indices,
                           parent_indices=parent_indices, interpolation_weights=weights, num_node_kids=kids,
                           use_trained_exp=True)
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii"}
    assert pkg["visibility_filter"].shape == (n,) and pkg["visibility_filter"].dtype == torch.bool
    pkg["render"].sum().backward()
    assert pc._xyz.grad is not None and pc._xyz.grad[:n].abs().sum() > 0
    type(pc).get_opacity = property(lambda s: torch.sigmoid(s._opacity))

    # --- the two "python twin" pipeline flags route through colors_precomp / cov3D_precomp
    pipe2 = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=True)
    pc = _PC(scene)
    a = glue.render(_viewpoint(cam), pc, pipe2, bg)["render"]
    b = glue.render(_viewpoint(cam), _PC(scene), pipe, bg)["render"]
    assert (a - b).abs().max() < 1e-5
