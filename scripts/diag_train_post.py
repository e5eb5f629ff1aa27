#!/usr/bin/env python
"""Tuning aid (GPU box): host time per phase of the train_post-shaped step (no device syncs added) next to the step time:
is the step bound by the GPU or by the Python / launch path between its two host waits?"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
import torch
import bench
import diff_gaussian_rasterization as dgr
from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
from hgs import hierarchy, synth
from hgs.optim import Adam

dev = torch.device("cuda:0")
W, H = 1920, 1080
cam = synth.make_camera(W, H)
full = synth.make_scene(500_000, cam, seed=0)
left = full.means3D[:, 0] < 0
h = hierarchy.merge_hierarchies([hierarchy.build_hierarchy(
    synth.Scene(full.means3D[m], full.scales[m], full.rotations[m], full.opacities[m], full.shs[m], 3)) for m in (left, ~left)])
nodes, boxes = h.nodes.to(dev), h.boxes.to(dev)
G = h.xyz.shape[0]
ri = torch.zeros(G, dtype=torch.int32, device=dev); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
w = torch.zeros(G, device=dev); ns = torch.zeros(G, dtype=torch.int32, device=dev)
attrs = dict(xyz=h.xyz, shs=h.shs, op=h.alpha.abs().reshape(-1, 1), sc=torch.exp(h.log_scales), rot=torch.nn.functional.normalize(h.rots))
params = {kk: torch.nn.Parameter(v.to(dev).contiguous()) for kk, v in attrs.items()}
lrs = dict(xyz=1.6e-5, shs=2.5e-3, op=1e-3, sc=1e-6, rot=1e-5)
opt = Adam([dict(params=[params[kk]], lr=lrs[kk], name=kk) for kk in params], lr=0.0, eps=1e-15)
target = torch.rand(3, H, W, device=dev)
mfull = torch.zeros(G, 3, device=dev, requires_grad=True)
g = torch.Generator().manual_seed(3)
vp_gpu, vp_cpu, zero3 = cam.camera_center.to(dev), cam.camera_center.cpu(), torch.zeros(3)
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
def step():
    t = time.perf_counter()
    limit = math.pow(2, torch.rand(1, generator=g).item() * (math.log2(0.1) - math.log2(0.005)) + math.log2(0.005))
    n = expand_to_size(nodes, boxes, limit, vp_gpu, zero3, ri, pi, ni); t = tick("expand (incl. its wait)", t)
    get_interpolation_weights(ni[:n], limit, nodes, boxes, vp_cpu, zero3, w, ns); t = tick("weights", t)
    rs = bench._settings(dgr, cam, dev, do_depth=False, interpolation_weights=w, num_node_kids=ns, render_indices=ri[:n], parent_indices=pi); t = tick("settings", t)
    color, _, _ = dgr.GaussianRasterizer(rs)(means3D=params["xyz"], means2D=mfull, shs=params["shs"], opacities=params["op"], scales=params["sc"], rotations=params["rot"]); t = tick("forward (incl. its wait)", t)
    loss = (color - target).abs().mean(); t = tick("loss", t)
    opt.zero_grad(set_to_none=True); t = tick("zero_grad", t)
    loss.backward(); t = tick("backward", t)
    opt.step(None); t = tick("adam", t)
for _ in range(8): step()
torch.cuda.synchronize(); acc.clear()
N = 60
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / N * 1e3
print(f"step {el:.3f} ms; host time per phase (ms):", {k: round(v / N * 1e3, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()) / N * 1e3, 3))
