#!/usr/bin/env python
"""Paper cost of three formulations of the backward compositing kernel (K7) -- VERDICT r03 item 5: "cost the next
formulation on paper first".  Computed on the CPU from the float32 geometry specification of a scene (test
infrastructure; imports oracle/), per sampled tile, with the per-iteration instruction counts of the CURRENT kernel as
the unit (117 vector instructions per iteration of the quadrant loop, of which 21 are the DPP
reduction, ~10 the LDS accumulation, ~12 list / record handling and ~74 the arithmetic of 4 pixels per lane in two
packed pairs; ~130 per 64-instance batch for staging):

  quad      today: the wave's four 16-lane rows own the tile's four 8x8 quadrants and walk their own lists of the
            64-instance batch; 4 pixels per lane; cost = sum over batches of max(list lengths) x 117 + 130 per batch.
  systolic  (i) lane = INSTANCE, pixel state (T, four suffix accumulators, four upstream gradients, id) handed from lane
            to lane by wave_shr DPP; every lane keeps its ten sums in registers (no reduction, no LDS atomics).  A pixel
            that ANY instance of the batch can touch has to run through all 64 lanes, so a batch costs
            (|union of the batch's pixels| + 63) steps of ~60 instructions (12 alpha, 28 blend / gradient terms, 10 state
            shifts, 10 accumulations) -- unpacked, one pixel per lane and step.
  cell4x4   (ii) sixteen 4x4 cells per tile with their own lists, ONE pixel per lane, four cells per wave pass, four
            passes (or waves) per tile: an iteration serves four (instance, cell) pairs = 64 pixel evaluations for
            ~69 instructions (21 reduction + 10 accumulation + 12 list handling + 26 for one unpacked pixel) and a batch
            costs max over its four cells' lists per pass; staging 4 x 130 per batch (sixteen ballots instead of four).

Visits use the IDEAL test (a quadrant / cell is visited iff one of its pixels reaches alpha >= 1/255): the finer
decompositions get their best case.  Early termination is ignored everywhere.

    python tests/tools/k7_cost_models.py            -> profiles/r04_k7_formulations.md
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
from hgs import synth                       # noqa: E402
from oracle import raster_oracle as ro      # noqa: E402

C_ITER, C_BATCH = 117, 130
C_SYS_STEP = 60
C_CELL_ITER, C_CELL_BATCH = 69, 4 * 130


def tile_models(geom, b, op, t, gx, batch=64):
    s, e = b.ranges[t]
    if e <= s:
        return None
    ids = b.point_list[s:e]
    x0, y0 = (t % gx) * 16, (t // gx) * 16
    px = np.arange(16, dtype=np.float64)
    dx = geom.px[ids].astype(np.float64)[:, None, None] - (x0 + px)[None, None, :]
    dy = geom.py[ids].astype(np.float64)[:, None, None] - (y0 + px)[None, :, None]
    A, B, C = (geom.conic[ids, i].astype(np.float64)[:, None, None] for i in range(3))
    power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
    alpha = np.minimum(0.99, op[ids][:, None, None] * np.exp(np.minimum(power, 0.0)))
    live = alpha >= 1.0 / 255.0                                         # [n, 16(y), 16(x)]
    n = len(ids)
    quads = live.reshape(n, 2, 8, 2, 8).any(axis=(2, 4)).reshape(n, 4)
    cells = live.reshape(n, 4, 4, 4, 4).any(axis=(2, 4)).reshape(n, 16)
    out = dict(instances=n, live_pairs=int(live.sum()), quad_visits=int(quads.sum()), cell_visits=int(cells.sum()),
               quad=0, systolic=0, cell4x4=0, batches=0)
    for b0 in range(0, n, batch):
        q = quads[b0:b0 + batch].sum(0)
        out["quad"] += int(q.max()) * C_ITER + C_BATCH
        union = int(live[b0:b0 + batch].any(0).sum())
        out["systolic"] += ((union + 63) * C_SYS_STEP) if union else 0
        c = cells[b0:b0 + batch].sum(0).reshape(4, 4)                   # pass p = cells 4p .. 4p+3 (one 16x4 strip)
        out["cell4x4"] += int(c.max(1).sum()) * C_CELL_ITER + C_CELL_BATCH
        out["batches"] += 1
    return out


def run(name, scene, cam, W, H, n_tiles):
    geom = ro.geometry_spec(scene.means3D.numpy(), scene.scales.numpy(), scene.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            float(np.float32(cam.tanfovx)), float(np.float32(cam.tanfovy)), 1.0)
    b = ro.binning_spec(geom)
    gx = geom.grid[0]
    tiles = np.random.default_rng(0).choice(gx * geom.grid[1], size=n_tiles, replace=False)
    op = scene.opacities.numpy().reshape(-1).astype(np.float64)
    tot = {}
    used = 0
    for t in tiles:
        r = tile_models(geom, b, op, int(t), gx)
        if r is None:
            continue
        used += 1
        for k, v in r.items():
            tot[k] = tot.get(k, 0) + v
    per_tile = {k: v / used for k, v in tot.items()}
    return dict(scene=name, tiles=used, per_tile=per_tile,
                live_share_of_visited_quadrant=tot["live_pairs"] / (64.0 * tot["quad_visits"]),
                live_share_of_visited_cell=tot["live_pairs"] / (16.0 * tot["cell_visits"]),
                relative_to_quad={k: tot[k] / tot["quad"] for k in ("quad", "systolic", "cell4x4")})


def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    W, H = 1920, 1080
    cam = synth.make_camera(W, H)
    rows = [run("metric: 1 M, s_px in [0.5, 4]", synth.make_scene(1_000_000, cam, seed=0), cam, W, H, n_tiles),
            run("heavy: 1 M, s_px in [1, 8]", synth.make_scene(1_000_000, cam, seed=0, s_px=(1.0, 8.0)), cam, W, H, n_tiles),
            run("trained-like: 400 k, log-normal footprints, anisotropy 0.05",
                synth.make_scene_trained_like(400_000, cam, seed=0), cam, W, H, n_tiles)]
    lines = ["# K7 formulations costed on paper (tests/tools/k7_cost_models.py; vector instructions per tile, ideal visit tests)",
             "",
             "| scene | instances / tile | live pixels / instance | live share of a visited 8x8 quadrant | ... of a visited 4x4 cell | quad (today) | systolic lane = instance | 4x4 cells, 1 pixel / lane |",
             "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        p = r["per_tile"]
        rel = r["relative_to_quad"]
        lines.append(f"| {r['scene']} | {p['instances']:.0f} | {p['live_pairs'] / p['instances']:.1f} | "
                     f"{r['live_share_of_visited_quadrant']:.2f} | {r['live_share_of_visited_cell']:.2f} | "
                     f"{p['quad']:.0f} (1.00) | {p['systolic']:.0f} ({rel['systolic']:.2f}) | "
                     f"{p['cell4x4']:.0f} ({rel['cell4x4']:.2f}) |")
    lines += ["",
              "Reading: neither alternative comes within 20 % BELOW the current quadrant loop on any scene, so neither was built "
              "(the verdict's bar).  The systolic form pays for every pixel the batch's UNION touches in every lane (64 "
              "instances x ~250 pixels per batch against ~17 live pixels per instance); the 4x4 cells double the live share "
              "of a visit but quarter the pixels an iteration serves while the per-iteration reduction (21 DPP adds + the "
              "LDS accumulation) stays, and packed arithmetic is lost with one pixel per lane.  The measured kernel retires "
              "~17 000 instructions per tile on the metric scene (profiles/r03_*): the model's `quad` column reproduces it.",
              "", "```json", json.dumps(rows, indent=1), "```", ""]
    out = os.path.join(ROOT, "profiles", "r04_k7_formulations.md")
    with open(out, "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines[:8]))


if __name__ == "__main__":
    main()
