#!/usr/bin/env python
"""Tuning aid (GPU box): a few fwd+bwd frames of the trained-scale scene alone (for rocprofv3 passes).
    python scripts/trained_loop.py [frames] [index|clustered] [metric]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
import torch
import diff_gaussian_rasterization as dgr
from hgs import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
order = sys.argv[2] if len(sys.argv) > 2 else "index"
W, H = 1920, 1080
dev = torch.device("cuda:0")
cam = synth.make_camera(W, H)
if len(sys.argv) > 3 and sys.argv[3] == "metric":
    scene = synth.make_scene(1_000_000, cam, seed=0).to(dev)
else:
    scene = synth.make_scene_trained_scale(375_000, cam, seed=0, order=order).to(dev)
gc, gd = (t.to(dev) for t in synth.upstream_grads(H, W, seed=1))
c = cam.to(dev)
e_i = torch.empty(0, dtype=torch.int32, device=dev); e_f = torch.empty(0, device=dev)
rs = dgr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy,
                                       bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=c.world_view_transform,
                                       projmatrix=c.full_proj_transform, sh_degree=3, campos=c.camera_center, prefiltered=False,
                                       debug=False, do_depth=True, render_indices=e_i, parent_indices=e_i,
                                       interpolation_weights=e_f, num_node_kids=e_i)
params = [t.requires_grad_(True) for t in (scene.means3D, scene.shs, scene.opacities, scene.scales, scene.rotations)]
for _ in range(n):
    for t in params:
        t.grad = None
    m2 = torch.zeros(scene.P, 3, device=dev, requires_grad=True)
    color, radii, invd = dgr.GaussianRasterizer(rs)(means3D=params[0], means2D=m2, shs=params[1], colors_precomp=None,
                                                   opacities=params[2], scales=params[3], rotations=params[4], cov3D_precomp=None)
    torch.autograd.backward([color, invd], [gc, gd])
torch.cuda.synchronize()
print("L", color.grad_fn.num_rendered)
