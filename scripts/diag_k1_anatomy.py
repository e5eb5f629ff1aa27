#!/usr/bin/env python
"""Tuning aid (GPU box): what is K1 made of?  Times `preprocess_fwd` alone (hgs_raster_fwd_stage1: K1 + scan + wait) at
the metric configuration, for the product library and for every ab_variants/libhgs_<name>.so given on the command line
(each in a process of its own: the library is loaded once per process).  The variants are builds with -DHGS_K1X=<mask>
(preprocess.hip: parts of K1 left out) or other -D switches (scripts/ab_build.sh).
    python scripts/diag_k1_anatomy.py [name ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys
root, libpath = sys.argv[1], sys.argv[2]
for p in (root, os.path.join(root, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
import ctypes as C
import torch
from hgs import _lib, synth
if libpath != "product":
    _lib.LIB_PATH = libpath
import bench
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C as dC
dev = torch.device("cuda:0")
W, H, P = 1920, 1080, 1_000_000
cam = synth.make_camera(W, H)
scene = synth.make_scene(P, cam, seed=0).to(dev)
rs = bench._settings(dgr, cam, dev)
lib = _lib.lib()
a, keep, P, M = dC._build_args(rs.bg, scene.means3D, None, scene.opacities, scene.scales, scene.rotations, 1.0, None,
                               rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, H, W, scene.shs, 3, rs.campos, False,
                               rs.interpolation_weights, rs.num_node_kids, True)
a.prepare_backward = 1
pl = dC._plan(lib, P, W, H, 0)
geom = torch.empty(pl["geom"], dtype=torch.uint8, device=dev)
radii = torch.empty(P, dtype=torch.int32, device=dev)
L = C.c_uint32(0)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def once():
    _lib.check(lib.hgs_raster_fwd_stage1(C.byref(a), geom.data_ptr(), _lib.ptr(radii), C.byref(L), st, 0), "stage1")
for _ in range(5): once()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    _lib.timing_read(True); _lib.timing_enable(True, ["preprocess_fwd"])
    for _ in range(20): once()
    torch.cuda.synchronize()
    _lib.timing_enable(False)
    ms, c = _lib.timing_read(True)["preprocess_fwd"]
    best = min(best, ms / c)
print("K1_MS", round(best * 1000, 1), "us  L", L.value)
"""
names = ["product"] + sys.argv[1:]
for n in names:
    path = "product" if n == "product" else os.path.join(ROOT, "ab_variants", f"libhgs_{n}.so")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, path], capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("K1_MS")]
    print(f"{n:16s}", line[-1] if line else ("FAILED " + r.stderr[-400:]))
