#!/usr/bin/env python
"""PIN KIT -- run this where the UPSTREAM extensions are installed (an NVIDIA machine with the reference's
``pip install submodules/hierarchy-rasterizer submodules/gaussianhierarchy``, /root/reference/requirements.txt:10-12):

    python tests/golden/make_upstream_golden.py            # writes tests/golden/upstream_*.npz
    git add tests/golden/upstream_*.npz                    # tests/test_upstream_pins.py then consumes them

It feeds the committed, seeded cases of this repository to the UPSTREAM ``diff_gaussian_rasterization`` and
``gaussian_hierarchy._C`` and stores inputs AND outputs, which pins the three semantics that could only be restated
from memory here (DESIGN.md sections 3 and 4; their reference call sites in brackets):

  upstream_raster_config1.npz   BASELINE configs[0] (1 k Gaussians, 128x128): color, radii, invdepth and all gradients
                                [gaussian_renderer/__init__.py:44-64,105-113; train_single.py:123]
  upstream_raster_post.npz      a render_post-shaped call with NON-EMPTY interpolation_weights / num_node_kids -- pins
                                how the kernel uses them (``lod_opacity``)  [gaussian_renderer/__init__.py:247-277]
  upstream_lod_cut.npz          expand_to_size + get_interpolation_weights on the synthetic hierarchy at three
                                thresholds -- pins the cut rule and the weight formula  [train_post.py:91-113,
                                render_hierarchy.py:55-80]
  upstream_raster_needles.npz   20 needles (3 000 px x 0.3-0.6 px at every angle, 128x128): a float32 conic of such a needle loses
                                positive definiteness (a c - b^2 cancels to its last bits; below ~2 500 px it never does), the
                                exponent A dx^2 + C dy^2 + 2 B dx dy turns POSITIVE on whole stripes -- the reference lineage then
                                SKIPS the Gaussian there ("power > 0"), this library keeps the conic positive definite in K1 and
                                clamps the exponent at 0.  Says which of the two upstream is nearer to.
  upstream_hier_file.npz        the bytes of a .hier written by upstream write_hierarchy and what upstream load_hierarchy
                                returns for it -- pins the file layout  [scene/gaussian_model.py:329,420-427]

Nothing of this repository's HIP / ctypes side is imported (the package directory that carries the same import names as
the upstream extensions is NOT put on sys.path); only the pure-Python scene / camera / hierarchy generators are loaded,
by file path.  A case that the upstream extension rejects is reported and skipped, the others are still written."""
import importlib.util
import os
import sys
import tempfile
import traceback

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HGS = os.path.join(ROOT, "hierarchical-3d-gaussians_amd", "hgs")


def _load(name):
    spec = importlib.util.spec_from_file_location(f"_pin_{name}", os.path.join(HGS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


synth = _load("synth")
hierarchy = _load("hierarchy")
DEV = "cuda"


def settings_kwargs(cam, bg, sh_degree, do_depth, interpolation_weights=None, num_node_kids=None):
    e_i = torch.empty(0, dtype=torch.int32, device=DEV)
    e_f = torch.empty(0, dtype=torch.float32, device=DEV)
    return dict(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                bg=bg.to(DEV), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(DEV),
                projmatrix=cam.full_proj_transform.to(DEV), sh_degree=sh_degree, campos=cam.camera_center.to(DEV),
                prefiltered=False, debug=False, do_depth=do_depth, render_indices=e_i,
                parent_indices=torch.empty(0, dtype=torch.int32),           # render_post passes CPU empties (:244-245)
                interpolation_weights=e_f if interpolation_weights is None else interpolation_weights,
                num_node_kids=e_i if num_node_kids is None else num_node_kids)


def cam_arrays(cam):
    return dict(W=cam.image_width, H=cam.image_height, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                campos=cam.camera_center.numpy())


def run_raster(scene, cam, bg, gc, gd, do_depth, weights=None, kids=None):
    import diff_gaussian_rasterization as dgr          # UPSTREAM
    req = lambda t: t.clone().to(DEV).requires_grad_(True)
    m3, sc, rot, op, sh = map(req, (scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs))
    m2 = torch.zeros(scene.P, 3, device=DEV, requires_grad=True)
    m2.retain_grad()
    kw = settings_kwargs(cam, bg, scene.sh_degree, do_depth,
                         None if weights is None else weights.to(DEV), None if kids is None else kids.to(DEV))
    try:
        rs = dgr.GaussianRasterizationSettings(**kw)
    except TypeError:                                     # an upstream build whose settings have no do_depth field
        kw.pop("do_depth")
        rs = dgr.GaussianRasterizationSettings(**kw)
    out = dgr.GaussianRasterizer(raster_settings=rs)(means3D=m3, means2D=m2, shs=sh, colors_precomp=None, opacities=op,
                                                     scales=sc, rotations=rot, cov3D_precomp=None)
    color, radii = out[0], out[1]
    invd = out[2] if len(out) > 2 else None
    loss = (color * gc.to(DEV)).sum()
    if do_depth and invd is not None:
        loss = loss + (invd * gd.to(DEV)).sum()
    loss.backward()
    res = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy().astype(np.int32),
               d_means3D=m3.grad.cpu().numpy(), d_means2D=m2.grad.cpu().numpy(), d_shs=sh.grad.cpu().numpy(),
               d_opacities=op.grad.cpu().numpy(), d_scales=sc.grad.cpu().numpy(), d_rotations=rot.grad.cpu().numpy())
    if invd is not None:
        res["invdepth"] = invd.detach().cpu().numpy()
    return res


def scene_arrays(scene):
    return dict(means3D=scene.means3D.numpy(), scales=scene.scales.numpy(), rotations=scene.rotations.numpy(),
                opacities=scene.opacities.numpy(), shs=scene.shs.numpy(), sh_degree=scene.sh_degree)


def case_config1():
    cam = synth.make_camera(128, 128)
    scene = synth.make_scene(1000, cam, seed=0)
    gc, gd = synth.upstream_grads(128, 128)
    bg = torch.tensor([0.1, 0.2, 0.3])
    out = run_raster(scene, cam, bg, gc, gd, True)
    return dict(**{"in_" + k: v for k, v in scene_arrays(scene).items()}, **{"cam_" + k: v for k, v in cam_arrays(cam).items()},
                bg=bg.numpy(), gc=gc.numpy(), gd=gd.numpy(), do_depth=True, **{"out_" + k: v for k, v in out.items()})


def needle_scene(cam, n=20, seed=9):
    """Needles through the frustum: one axis 3 000 px long on screen, the other two 0.3 .. 0.6 px, random orientation."""
    scene = synth.make_scene(n, cam, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    fx = cam.image_width / (2.0 * cam.tanfovx)
    z = scene.means3D[:, 2]
    thin = (0.3 + 0.3 * torch.rand(n, 2, generator=g)) * (z / fx)[:, None]
    scene.scales = torch.cat([(3000.0 * z / fx)[:, None], thin], 1).contiguous()
    scene.opacities = (0.3 + 0.6 * torch.rand(n, 1, generator=g)).contiguous()
    return scene


def case_needles():
    cam = synth.make_camera(128, 128)
    scene = needle_scene(cam)
    gc, gd = synth.upstream_grads(128, 128, seed=9)
    bg = torch.tensor([0.0, 0.0, 0.0])
    out = run_raster(scene, cam, bg, gc, gd, True)
    return dict(**{"in_" + k: v for k, v in scene_arrays(scene).items()}, **{"cam_" + k: v for k, v in cam_arrays(cam).items()},
                bg=bg.numpy(), gc=gc.numpy(), gd=gd.numpy(), do_depth=True, **{"out_" + k: v for k, v in out.items()})


def _small_hierarchy():
    cam = synth.make_camera(320, 208)
    h = hierarchy.build_hierarchy(synth.make_scene(3000, cam, seed=5, s_px=(0.7, 3.0)))
    return cam, h


TAUS_PX = (0.0, 3.0, 15.0)


def case_lod_cut():
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights     # UPSTREAM
    cam, h = _small_hierarchy()
    nodes, boxes = h.nodes.to(DEV), h.boxes.to(DEV)
    G = h.xyz.shape[0]
    out = dict(nodes=h.nodes.numpy(), boxes=h.boxes.numpy(), viewpoint=cam.camera_center.numpy(), taus=[])
    for i, tau_px in enumerate(TAUS_PX):
        tau = (2 * tau_px + 1) * cam.tanfovx / (0.5 * cam.image_width)          # render_hierarchy.py:55-56
        ri = torch.zeros(G, dtype=torch.int32, device=DEV); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
        w = torch.zeros(G, device=DEV); ns = torch.zeros(G, dtype=torch.int32, device=DEV)
        n = expand_to_size(nodes, boxes, tau, cam.camera_center.to(DEV), torch.zeros(3), ri, pi, ni)
        get_interpolation_weights(ni[:n], tau, nodes, boxes, cam.camera_center.cpu(), torch.zeros(3), w, ns)
        out["taus"].append(tau)
        out[f"n_{i}"] = int(n)
        out[f"render_indices_{i}"] = ri[:n].cpu().numpy()
        out[f"parent_indices_{i}"] = pi[:n].cpu().numpy()
        out[f"nodes_for_render_indices_{i}"] = ni[:n].cpu().numpy()
        out[f"weights_{i}"] = w[:n].cpu().numpy()
        out[f"num_siblings_{i}"] = ns[:n].cpu().numpy()
    out["taus"] = np.asarray(out["taus"], dtype=np.float64)
    return out


def case_raster_post():
    """The rows render_post would hand to the op for the 3 px cut of the small hierarchy, lerped exactly as
    gaussian_renderer/__init__.py:204-218 does, with the cut's weights / sibling counts in the settings."""
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights     # UPSTREAM
    cam, h = _small_hierarchy()
    nodes, boxes = h.nodes.to(DEV), h.boxes.to(DEV)
    G = h.xyz.shape[0]
    tau = (2 * 3.0 + 1) * cam.tanfovx / (0.5 * cam.image_width)
    ri = torch.zeros(G, dtype=torch.int32, device=DEV); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
    w = torch.zeros(G, device=DEV); ns = torch.zeros(G, dtype=torch.int32, device=DEV)
    n = expand_to_size(nodes, boxes, tau, cam.camera_center.to(DEV), torch.zeros(3), ri, pi, ni)
    get_interpolation_weights(ni[:n], tau, nodes, boxes, cam.camera_center.cpu(), torch.zeros(3), w, ns)
    r, p = ri[:n].long().cpu(), pi[:n].long().cpu()
    t = w[:n].cpu().unsqueeze(1); ti = 1 - t
    rots = torch.nn.functional.normalize(h.rots)
    rn, rp = rots[r], rots[p]
    rp = torch.where(((rn * rp).sum(1) < 0)[:, None], -rp, rp)
    sc = torch.exp(h.log_scales)
    rows = synth.Scene((t * h.xyz[r] + ti * h.xyz[p]).contiguous(), (t * sc[r] + ti * sc[p]).contiguous(),
                       (t * rn + ti * rp).contiguous(), (t * h.alpha.abs()[r] + ti * h.alpha.abs()[p]).contiguous(),
                       (t.unsqueeze(2) * h.shs[r] + ti.unsqueeze(2) * h.shs[p]).contiguous(), 3)
    gc, gd = synth.upstream_grads(cam.image_height, cam.image_width, seed=21)
    bg = torch.zeros(3)
    weights, kids = w.cpu().clone(), ns.cpu().clone()          # [P_total] arrays, first n valid (as render_post passes them)
    out = run_raster(rows, cam, bg, gc, gd, False, weights, kids)
    return dict(**{"in_" + k: v for k, v in scene_arrays(rows).items()}, **{"cam_" + k: v for k, v in cam_arrays(cam).items()},
                bg=bg.numpy(), gc=gc.numpy(), gd=gd.numpy(), do_depth=False, interpolation_weights=weights.numpy(),
                num_node_kids=kids.numpy(), n=int(n), **{"out_" + k: v for k, v in out.items()})


def case_hier_file():
    from gaussian_hierarchy._C import load_hierarchy, write_hierarchy               # UPSTREAM
    cam = synth.make_camera(320, 208)
    h = hierarchy.build_hierarchy(synth.make_scene(200, cam, seed=6))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "pin.hier")
        write_hierarchy(path, h.xyz.to(DEV), h.shs.to(DEV), h.alpha.to(DEV), h.log_scales.to(DEV), h.rots.to(DEV),
                        h.nodes.to(DEV), h.boxes.to(DEV))                            # scene/gaussian_model.py:420-427
        raw = np.fromfile(path, dtype=np.uint8)
        back = load_hierarchy(path)                                                   # scene/gaussian_model.py:329
    names = ("xyz", "shs", "alpha", "log_scales", "rots", "nodes", "boxes")
    out = dict(file_bytes=raw, **{"in_" + k: getattr(h, k).numpy() for k in names})
    for k, v in zip(names, back):
        out["loaded_" + k] = v.cpu().numpy()
    return out


def main():
    if not torch.cuda.is_available():
        raise SystemExit("needs the upstream CUDA extensions and a GPU they run on")
    import diff_gaussian_rasterization as dgr
    if os.path.commonpath([os.path.abspath(dgr.__file__), ROOT]) == ROOT:
        raise SystemExit("diff_gaussian_rasterization resolves to THIS repository -- the goldens must come from the "
                         "upstream extension; remove hierarchical-3d-gaussians_amd from PYTHONPATH")
    done = []
    for name, fn in (("raster_config1", case_config1), ("lod_cut", case_lod_cut), ("raster_post", case_raster_post),
                     ("raster_needles", case_needles), ("hier_file", case_hier_file)):
        try:
            data = fn()
            np.savez_compressed(os.path.join(HERE, f"upstream_{name}.npz"), **data)
            done.append(name)
        except Exception:
            print(f"[pin kit] case {name} failed against the upstream extension:", file=sys.stderr)
            traceback.print_exc()
    print("wrote:", ", ".join(f"tests/golden/upstream_{n}.npz" for n in done) or "nothing")


if __name__ == "__main__":
    main()
