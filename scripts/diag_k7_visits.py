#!/usr/bin/env python
"""Diagnostic (GPU box, library built from scripts/diag/k7_visit_stamps.patch): cycle stamps of the first 16 waves of
render_bwd_quad_kernel at every visit (mode 1) or at five points of every visit (mode 2); HGS_GRID_LIMIT limits the launch
(1 024 = one wave per SIMD).  Prints cycles per visit / per section.
    HGS_GRID_LIMIT=1024 python scripts/diag_k7_visits.py <stamps per visit: 1|5>"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hierarchical-3d-gaussians_amd"))
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from hgs import _lib, synth  # noqa: E402

per = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
W, H = 1920, 1080
cam0 = synth.make_camera(W, H)
scene = synth.make_scene(1_000_000, cam0, seed=0).to(dev)
gc, gd = (t.to(dev) for t in synth.upstream_grads(H, W, seed=1))
params = dict(means3D=scene.means3D, shs=scene.shs, opacities=scene.opacities, scales=scene.scales, rotations=scene.rotations)
for t in params.values():
    t.requires_grad_(True)
cams = [synth.orbit_camera(W, H, j, 8, radius=0.05, tilt=0.004) for j in range(8)]
d = bench.DropIn(dgr, params, scene.sh_degree, [bench._settings(dgr, c, dev) for c in cams], gc, gd, dev)
for _ in range(4):
    d.step()
torch.cuda.synchronize()
lib = _lib.lib()
lib.hgs_debug_set_trace.argtypes = [C.c_void_p]
buf = torch.zeros(16 * 4096, dtype=torch.int64, device=dev)
assert lib.hgs_debug_set_trace(C.c_void_p(buf.data_ptr())) == 0
d.step()
torch.cuda.synchronize()
assert lib.hgs_debug_set_trace(C.c_void_p(0)) == 0
tr = buf.cpu().numpy().reshape(16, 4096)
print(f"grid limit {os.environ.get('HGS_GRID_LIMIT')}, {per} stamp(s) per visit")
for b in range(16):
    n = int(tr[b, 0])
    if n <= 2 + per:
        continue
    st = tr[b, 2:n].astype(np.int64)
    nv = len(st) // per
    st = st[:nv * per].reshape(nv, per)
    dv = np.diff(st[:, 0])                          # visit start to next visit start (includes batch boundaries)
    inner = dv[dv < np.percentile(dv, 90)]
    line = f"wave {b:2d}: {nv} visits, list {int(tr[b, 1])}; cycles per visit median {np.median(dv):.0f} mean(<p90) {inner.mean():.0f} p90 {np.percentile(dv, 90):.0f} max {dv.max()}"
    if per > 1:
        sec = np.diff(st, axis=1)
        line += "; sections median " + " ".join(f"{np.median(sec[:, k]):.0f}" for k in range(per - 1))
        line += f" | tail->next {np.median(st[1:, 0] - st[:-1, -1]):.0f}"
    print(line)
