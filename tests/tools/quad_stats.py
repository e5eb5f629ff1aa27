#!/usr/bin/env python
"""Work model of the compositing kernels' two decompositions, computed on the CPU from the float32 geometry
specification of a scene (test infrastructure; imports oracle/):
  packed   one wave per tile, a visited instance costs one pass per reached 16x8 HALF (box test of K1's alpha >= 1/255
           extents against the halves), wave-uniform
  quad     one wave per tile, the four 16-lane rows of the wave own the four 8x8 QUADRANTS and walk their OWN lists of
           the 64-instance batch: an iteration serves up to four (instance, quadrant) pairs; iterations per batch =
           the longest of the four lists
Also reported: the iterations 128-instance batches would need (quad_iterations_b128) and the bound without any batch
boundary (quad_iterations_unbatched: the longest of a tile's four lists) -- what a record ring could win at most.
Early termination is ignored.   python tests/tools/quad_stats.py [heavy]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
from hgs import synth                       # noqa: E402
from oracle import raster_oracle as ro      # noqa: E402


def main(P=1_000_000, W=1920, H=1080, n_tiles=400, s_px=(0.5, 4.0), batch=64):
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=0, s_px=s_px)
    geom = ro.geometry_spec(scene.means3D.numpy(), scene.scales.numpy(), scene.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            float(np.float32(cam.tanfovx)), float(np.float32(cam.tanfovy)), 1.0)
    b = ro.binning_spec(geom)
    gx = geom.grid[0]
    tiles = np.random.default_rng(0).choice(gx * geom.grid[1], size=n_tiles, replace=False)
    op = scene.opacities.numpy().reshape(-1).astype(np.float64)
    A, B, C = (geom.conic[:, i].astype(np.float64) for i in range(3))
    T2 = 2.0 * (np.log(np.maximum(255.0 * op, 1e-300)) + 1e-3 * 0.6931471805599453)
    det = A * C - B * B
    ok = (T2 > 0) & (det > 0)
    ex = np.where(ok, np.sqrt(np.maximum(T2 * C / np.where(ok, det, 1), 0)) * 1.0001 + 5e-3, -1.0)
    ey = np.where(ok, np.sqrt(np.maximum(T2 * A / np.where(ok, det, 1), 0)) * 1.0001 + 5e-3, -1.0)
    tot = dict(instances=0, visited=0, half_visits=0, quad_visits=0, quad_iterations=0, batches=0,
               quad_iterations_b128=0, quad_iterations_unbatched=0)
    for t in tiles:
        s, e = b.ranges[t]
        if e <= s:
            continue
        ids = b.point_list[s:e]
        x = geom.px[ids].astype(np.float64) - (t % gx) * 16
        y = geom.py[ids].astype(np.float64) - (t // gx) * 16
        hx, hy = ex[ids], ey[ids]
        xr = [(x - hx <= 7) & (x + hx >= 0), (x - hx <= 15) & (x + hx >= 8)]
        yr = [(y - hy <= 7) & (y + hy >= 0), (y - hy <= 15) & (y + hy >= 8)]
        xany = (x - hx <= 15) & (x + hx >= 0)
        halves = np.stack([xany & yr[0], xany & yr[1]], 1)
        quads = np.stack([xr[0] & yr[0], xr[1] & yr[0], xr[0] & yr[1], xr[1] & yr[1]], 1)
        n = len(ids)
        tot["instances"] += n
        tot["visited"] += int(halves.any(1).sum())
        tot["half_visits"] += int(halves.sum())
        tot["quad_visits"] += int(quads.sum())
        for b0 in range(0, n, batch):
            q = quads[b0:b0 + batch].sum(0)
            tot["quad_iterations"] += int(q.max())
            tot["batches"] += 1
        for b0 in range(0, n, 2 * batch):
            tot["quad_iterations_b128"] += int(quads[b0:b0 + 2 * batch].sum(0).max())
        # rows that never wait for each other at batch boundaries: the longest of the tile's four lists
        tot["quad_iterations_unbatched"] += int(quads.sum(0).max())
    out = dict(scene=f"{P} Gaussians, {W}x{H}, s_px {s_px}", tiles=n_tiles, batch=batch, totals=tot,
               per_instance={k: v / tot["instances"] for k, v in tot.items() if k != "instances"},
               quad_list_balance=tot["quad_visits"] / (4.0 * tot["quad_iterations"]),
               iterations_per_half_visit=tot["quad_iterations"] / tot["half_visits"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "heavy":
        main(s_px=(1.0, 8.0))
    else:
        main()
