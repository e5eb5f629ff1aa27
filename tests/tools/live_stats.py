#!/usr/bin/env python
"""Lane utilisation of the compositing kernels, computed on the CPU from the float32 geometry specification
(oracle/raster_oracle.py) of the benchmark scene: for a seeded sample of tiles, how many of a tile's instances have
ANY pixel with alpha >= 1/255, how many 16x8 tile halves (the skip unit of K6 / K7: one packed strip pair), 16x4 strips
and 8x8 quadrants that is, and how many pixels.  Early termination is ignored (slightly overestimates the live work).

    python tests/tools/live_stats.py > profiles/r02_live_lane_stats.json

Reading: a live half costs one pass of the packed live path over 128 pixel slots; live pixels / (live halves x 128) is
the fraction of those slots doing useful work."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
from hgs import synth                       # noqa: E402
from oracle import raster_oracle as ro      # noqa: E402  (analysis script: test infrastructure, not product)


def main(P=1_000_000, W=1920, H=1080, n_tiles=300):
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=0)
    geom = ro.geometry_spec(scene.means3D.numpy(), scene.scales.numpy(), scene.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            float(np.float32(cam.tanfovx)), float(np.float32(cam.tanfovy)), 1.0)
    b = ro.binning_spec(geom)
    gx = geom.grid[0]
    tiles = np.random.default_rng(0).choice(gx * geom.grid[1], size=n_tiles, replace=False)
    op = scene.opacities.numpy().reshape(-1)
    tot = dict(instances=0, live_instances=0, live_halves=0, live_strips=0, live_quadrants=0, live_pixels=0)
    for t in tiles:
        s, e = b.ranges[t]
        if e <= s:
            continue
        ids = b.point_list[s:e]
        ty0, tx0 = (t // gx) * 16, (t % gx) * 16
        ys, xs = np.mgrid[ty0:ty0 + 16, tx0:tx0 + 16]
        dx = geom.px[ids][:, None, None] - xs[None]
        dy = geom.py[ids][:, None, None] - ys[None]
        A, B, C = geom.conic[ids, 0], geom.conic[ids, 1], geom.conic[ids, 2]
        pw = -0.5 * (A[:, None, None] * dx * dx + C[:, None, None] * dy * dy) - B[:, None, None] * dx * dy
        live = (np.minimum(0.99, op[ids][:, None, None] * np.exp(np.minimum(pw, 0))) >= 1 / 255)
        n = len(ids)
        tot["instances"] += n
        tot["live_instances"] += int(live.any(axis=(1, 2)).sum())
        tot["live_halves"] += int(live.reshape(n, 2, 8, 16).any(axis=(2, 3)).sum())
        tot["live_strips"] += int(live.reshape(n, 4, 4, 16).any(axis=(2, 3)).sum())
        tot["live_quadrants"] += int(live.reshape(n, 2, 8, 2, 8).any(axis=(2, 4)).sum())
        tot["live_pixels"] += int(live.sum())
    out = dict(scene=f"{P} Gaussians, {W}x{H}, seed 0", tiles_sampled=int(n_tiles), totals=tot,
               per_instance={k: v / tot["instances"] for k, v in tot.items() if k != "instances"},
               lane_utilisation_of_executed_halves=tot["live_pixels"] / (128.0 * tot["live_halves"]),
               instances_without_any_live_pixel=1.0 - tot["live_instances"] / tot["instances"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
