#!/usr/bin/env python
"""Tuning aid (GPU box): where a frame of the configs[4] loop (cut + weights + 4K render, 50 M nodes) spends its time,
phase by phase with a device sync after each, next to the free-running frame rate and the allocator's counters."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
import torch
import bench
import diff_gaussian_rasterization as dgr
from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
from hgs import hierarchy, synth

dev = torch.device("cuda:0")
W, H = 3840, 2160
cam = synth.make_camera(W, H)
h = hierarchy.build_hierarchy_on_device(int(os.environ.get("LEAVES", 25_000_000)), cam, dev, seed=0)
G = h.nodes.shape[0]
ri = torch.zeros(G, dtype=torch.int32, device=dev); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
w = torch.zeros(G, device=dev); ns = torch.zeros(G, dtype=torch.int32, device=dev)
cams = [synth.orbit_camera(W, H, j, 8, radius=0.05, tilt=0.004) for j in range(8)]
vps = [(c.camera_center.to(dev), c.camera_center.cpu()) for c in cams]
tau = (2 * 3.0 + 1) * cam.tanfovx / (0.5 * W)
m2 = torch.zeros(G, 3, device=dev)
sc, zero3 = torch.exp(h.log_scales), torch.zeros(3)
sync = torch.cuda.synchronize


def frame(j, timed):
    t = [time.perf_counter()]
    n = expand_to_size(h.nodes, h.boxes, tau, vps[j][0], zero3, ri, pi, ni)
    if timed: sync()
    t.append(time.perf_counter())
    get_interpolation_weights(ni[:n], tau, h.nodes, h.boxes, vps[j][1], zero3, w, ns)
    if timed: sync()
    t.append(time.perf_counter())
    rs = bench._settings(dgr, cams[j], dev, do_depth=False, interpolation_weights=w, num_node_kids=ns,
                         render_indices=ri[:n], parent_indices=pi)
    t.append(time.perf_counter())
    with torch.no_grad():
        color, radii, _ = dgr.GaussianRasterizer(rs)(means3D=h.xyz, means2D=m2, shs=h.shs, opacities=h.alpha,
                                                     scales=sc, rotations=h.rots)
    t.append(time.perf_counter())
    if timed: sync()
    t.append(time.perf_counter())
    return [b - a for a, b in zip(t, t[1:])]


for i in range(8):
    frame(i % 8, False)
sync()
ms0 = torch.cuda.memory_stats(dev)
t0 = time.perf_counter()
for i in range(16):
    frame(i % 8, False)
sync()
free_run = (time.perf_counter() - t0) / 16
ms1 = torch.cuda.memory_stats(dev)
rows = [frame(i % 8, True) for i in range(16)]
names = ["expand(+sync)", "weights(+sync)", "settings", "render call (host)", "render (sync)"]
print("free-running ms/frame", round(free_run * 1e3, 3))
for k, nm in enumerate(names):
    print(f"  {nm:22s} median {statistics.median(r[k] for r in rows) * 1e3:8.3f} ms   max {max(r[k] for r in rows) * 1e3:8.3f}")
for key in ("segment.all.allocated", "segment.all.freed", "num_alloc_retries", "allocation.all.allocated"):
    print(key, ms1.get(key, 0) - ms0.get(key, 0))
print("reserved GB", ms1.get("reserved_bytes.all.current", 0) / 1e9, "allocated GB", ms1.get("allocated_bytes.all.current", 0) / 1e9)
print("stats", dgr._C.stats)
free, total = torch.cuda.mem_get_info()
print("mem free/total GB", free / 1e9, total / 1e9)
os.system("rocm-smi --showclocks 2>/dev/null | grep -i 'sclk\\|mclk\\|fclk' | head -6; nproc; cat /proc/loadavg")
