#!/usr/bin/env python
"""Tuning aid (GPU box): what is K1 (preprocess forward) made of?  Stage time of `preprocess_fwd` at the metric
configuration for the call shapes that switch parts of it off:
  grad      shs, a backward announced (stores d(rgb)/d(direction): the benchmark's K1)
  no_grad   shs, forward only (no Jacobian store)
  colors    colors_precomp instead of shs (no SH block through LDS at all: the geometry chain alone)
and the same three for K8 where they apply."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
import torch

import bench
import diff_gaussian_rasterization as dgr
from hgs import _lib, synth

dev = torch.device("cuda:0")
W, H, P = 1920, 1080, 1_000_000
cam = synth.make_camera(W, H)
scene = synth.make_scene(P, cam, seed=0).to(dev)
rs = bench._settings(dgr, cam, dev)
rast = dgr.GaussianRasterizer(rs)
gc, gd = (t.to(dev) for t in synth.upstream_grads(H, W))
colors = torch.rand(P, 3, device=dev)


def run(mode, n=12):
    def once():
        if mode == "no_grad":
            with torch.no_grad():
                rast(means3D=scene.means3D, means2D=None, shs=scene.shs, opacities=scene.opacities, scales=scene.scales,
                     rotations=scene.rotations)
            return
        prm = [scene.means3D, scene.opacities, scene.scales, scene.rotations, scene.shs, colors]
        for t in prm:
            t.requires_grad_(True); t.grad = None
        kw = dict(colors_precomp=colors) if mode == "colors" else dict(shs=scene.shs)
        c, r, d = rast(means3D=scene.means3D, means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
                       opacities=scene.opacities, scales=scene.scales, rotations=scene.rotations, **kw)
        torch.autograd.backward([c, d], [gc, gd])
    for _ in range(4):
        once()
    torch.cuda.synchronize()
    _lib.timing_read(True)
    _lib.timing_enable(True)
    for _ in range(n):
        once()
    torch.cuda.synchronize()
    _lib.timing_enable(False)
    tm = {k: ms / c for k, (ms, c) in _lib.timing_read(True).items() if c}
    print(f"{mode:8s}", {k: round(v, 4) for k, v in tm.items() if k in ("preprocess_fwd", "preprocess_bwd", "render_fwd", "render_bwd")})


for m in ("grad", "no_grad", "colors"):
    run(m)
