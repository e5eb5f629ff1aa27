#!/usr/bin/env python
"""Wall time of each of the first 60 drop-in steps of a fresh process (device-synchronised per step): is there a ramp?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hierarchical-3d-gaussians_amd"))
import torch
import bench
import diff_gaussian_rasterization as dgr
from hgs import synth
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
if os.environ.get("PREWARM"):        # another scene first: are the slow first steps the GPU's clocks or this scene's first touches?
    sc2 = synth.make_scene(300_000, cam0, seed=3).to(dev)
    p2 = dict(means3D=sc2.means3D, shs=sc2.shs, opacities=sc2.opacities, scales=sc2.scales, rotations=sc2.rotations)
    for t in p2.values():
        t.requires_grad_(True)
    d2 = bench.DropIn(dgr, p2, sc2.sh_degree, [bench._settings(dgr, c, dev) for c in cams], gc, gd, dev)
    for _ in range(int(os.environ["PREWARM"])):
        d2.step()
torch.cuda.synchronize()
ts = []
for i in range(80):
    t0 = time.perf_counter(); d.step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("ms per step (synchronised):", [round(x, 3) for x in ts[:32]])
print("mean of steps 5..24: %.4f   25..44: %.4f   45..79: %.4f" % (sum(ts[5:25]) / 20, sum(ts[25:45]) / 20, sum(ts[45:]) / 35))
# unsynchronised blocks of 20, as the bench times them
for k in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        d.step()
    torch.cuda.synchronize(); print("block of 20 steps: %.4f ms per step" % ((time.perf_counter() - t0) * 1e3 / 20))
