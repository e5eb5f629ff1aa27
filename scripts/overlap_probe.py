#!/usr/bin/env python
"""Probe: do the HBM-bound stages of one view (preprocess, binning, preprocess backward) overlap with the ALU-bound
compositing kernels of another when forward and backward run on two HIP streams?  Same work as bench.py's default
step (4 views, batched SH ends); prints frames/s for the single-stream and the two-stream schedule."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C
from hgs import dp, synth


def main():
    dev = torch.device("cuda:0")
    W, H, P, k = 1920, 1080, 1_000_000, 4
    base = synth.make_camera(W, H)
    scene = synth.make_scene(P, base, seed=0).to(dev)
    cams = [synth.orbit_camera(W, H, j, k).to(dev) for j in range(k)]
    gc, gd = (t.to(dev) for t in synth.upstream_grads(H, W))
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    params = {n: getattr(scene, n) for n in names}
    bucket = dp.GradBucket({n: tuple(v.shape) for n, v in params.items()}, dev)
    e_i = torch.empty(0, dtype=torch.int32, device=dev)
    e_f = torch.empty(0, dtype=torch.float32, device=dev)
    bg = torch.zeros(3, device=dev)
    campos = [c.camera_center for c in cams]
    prio = {"nn": (0, 0), "hn": (-1, 0), "nh": (0, -1)}[os.environ.get("PROBE_PRIO", "nn")]
    sA, sB = torch.cuda.Stream(device=dev, priority=prio[0]), torch.cuda.Stream(device=dev, priority=prio[1])

    def forward(j, rgb):
        c = cams[j]
        return _C.rasterize_gaussians(bg, params["means3D"], rgb, params["opacities"], params["scales"],
                                      params["rotations"], 1.0, None, c.world_view_transform, c.full_proj_transform,
                                      c.tanfovx, c.tanfovy, H, W, None, 3, c.camera_center, False, False, e_i, e_i,
                                      e_f, e_i, True, prepare_backward=True)

    def backward(call, color, invd, j):
        return _C.rasterize_gaussians_backward(call, color, invd, gc, gd, out=bucket.views, accumulate=j > 0)

    def step(two_streams):
        fa = sA if two_streams else torch.cuda.current_stream(dev)
        fb = sB if two_streams else torch.cuda.current_stream(dev)
        with torch.cuda.stream(fa):
            rgbs, clamps = _C.sh_colors_batched(params["means3D"], params["shs"], 3, campos)
        d_rgbs, keep = [], []
        for j in range(k):
            with torch.cuda.stream(fa):
                out = forward(j, rgbs[j])
                color, invd, call = out[1], out[6], out[7]
                ev = torch.cuda.Event()
                ev.record(fa)
            with torch.cuda.stream(fb):
                fb.wait_event(ev)
                if two_streams:
                    for t in (color, invd, call.geom, call.binb, call.img, call.scratch, rgbs[j], clamps[j], out[2]):
                        t.record_stream(fb)
                g = backward(call, color, invd, j)
                d_rgbs.append(g[1])
            keep.append(out)
        with torch.cuda.stream(fb):
            _C.sh_colors_batched_backward(params["means3D"], params["shs"], 3, campos, clamps, d_rgbs,
                                          bucket.views["shs"], bucket.views["means3D"])
        if two_streams:
            torch.cuda.current_stream(dev).wait_stream(fa)
            torch.cuda.current_stream(dev).wait_stream(fb)
            if os.environ.get("PROBE_STEP_BARRIER", "1") == "1":
                fa.wait_stream(fb)      # an optimiser update would sit here: the next step's forwards wait for this
                                        # step's backwards (without it the probe overlaps across the step boundary)

    res = {}
    for mode in (False, True, False, True):
        for _ in range(5):
            step(mode)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 30
        for _ in range(n):
            step(mode)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        res.setdefault("two_streams" if mode else "one_stream", []).append(k / dt)
    res["prio_fwd_bwd"] = prio
    print(json.dumps(res))


if __name__ == "__main__":
    main()
