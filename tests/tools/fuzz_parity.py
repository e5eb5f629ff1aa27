#!/usr/bin/env python
"""Soak test (GPU): many random small frames through the HIP op and the float64 oracle -- random Gaussian counts, image
sizes (not multiples of the tile), fields of view, SH degrees, footprint ranges (sub-pixel to image-filling), opacity
ranges (up to fully opaque stacks), backgrounds, scale modifiers, with / without the depth channel, SH or precomputed
colours.  Prints the worst error per quantity and every case above tolerance.

    python tests/tools/fuzz_parity.py [n_cases] [first_seed]            (test infrastructure: imports oracle/)"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import parity as pa                      # noqa: E402
from hgs import synth                    # noqa: E402

TOL = 1e-5


def make_case(seed):
    """-> (scene, cam, bg, gc, gd, kwargs of run_oracle / run_hip, description)"""
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(17, 300)), int(rng.integers(17, 200))
    P = int(rng.choice([1, 7, 64, 300, 1500, 4000]))
    deg = int(rng.integers(0, 4))
    fov = float(rng.uniform(25.0, 100.0))
    s_hi = float(rng.choice([2.0, 6.0, 30.0, 150.0]))
    s_lo = float(rng.choice([0.05, 0.3, 1.0]))
    cam = synth.make_camera(W, H, fov)
    scene = synth.make_scene(P, cam, seed=seed, sh_degree=deg, s_px=(s_lo, s_hi),
                             z_range=(float(rng.uniform(0.15, 2.0)), float(rng.uniform(3.0, 40.0))))
    opaque = bool(rng.random() < 0.3)
    if opaque:
        scene.opacities[:] = torch.clamp(scene.opacities * 1.6, max=1.0)      # opaque stacks: saturation paths
    depth = bool(rng.random() < 0.6)
    sm = float(rng.choice([1.0, 1.0, 0.5, 1.7]))
    bg = torch.tensor(rng.uniform(0, 1, 3), dtype=torch.float32)
    gc, gd = synth.upstream_grads(H, W, seed=seed)
    kw = dict(scale_modifier=sm, do_depth=depth)
    if rng.random() < 0.25:
        kw["colors_precomp"] = torch.tensor(rng.uniform(0, 1, (P, 3)), dtype=torch.float32)
    desc = dict(W=W, H=H, P=P, deg=deg, fov=fov, s=(s_lo, s_hi), depth=depth, sm=sm, opaque=opaque,
                precomp="colors_precomp" in kw)
    return scene, cam, bg, gc, gd, kw, desc


def run_cases(n_cases, seed0, dev):
    """Returns the report dict ({"worst": ..., "above_tolerance": [...], "index_mismatches": [...]})."""
    worst, worst_elem, bad, idx_bad = {}, {}, [], []
    for c in range(n_cases):
        scene, cam, bg, gc, gd, kw, desc = make_case(seed0 + c)
        depth = kw["do_depth"]
        oo, og = pa.run_oracle(scene, cam, bg, gc, gd, **kw)
        hip = pa.run_hip(scene, cam, bg, gc, gd, dev, **kw)
        mism = pa.check_indices(hip, oo)
        if any(mism.values()):
            idx_bad.append((seed0 + c, {k: v for k, v in mism.items() if v}))
        st = pa.compare(hip, oo, og, do_depth=depth)
        for k, v in st.items():
            if isinstance(v, dict):
                e = max(v["maxrel"], v["l2"]) if v["scale"] > 0 else v["maxrel"]
                if e > worst.get(k, (0, None))[0]:
                    worst[k] = (e, seed0 + c)
                for fig in ("mixed", "p999_rel"):                    # element-wise figures (tests/parity.py::err_stats)
                    if v[fig] > worst_elem.get((k, fig), (0, None))[0]:
                        worst_elem[(k, fig)] = (v[fig], seed0 + c)
                # fewer than 8 Gaussians: "relative to the tensor's maximum" degenerates into "relative to the entry
                # itself", and a single float32 sum with cancellation (terms 100 x the result, seed 1027) shows 7e-5
                # against the float64 oracle; the reference lineage's float32 atomics are no better there
                if e > (TOL if desc["P"] >= 8 else 10 * TOL):
                    bad.append((seed0 + c, k, e, dict(desc, fragile=st["fragile_frac"])))
    return {"cases": n_cases, "first_seed": seed0, "tolerance": TOL,
            "worst": {k: {"err": v[0], "seed": v[1]} for k, v in worst.items()},
            "worst_elementwise": {f"{k}:{fig}": {"value": v[0], "seed": v[1]} for (k, fig), v in worst_elem.items()},
            "above_tolerance": bad, "index_mismatches": idx_bad}


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rep = run_cases(n_cases, seed0, torch.device("cuda:0"))
    print(json.dumps(rep, default=str))
    return 1 if rep["above_tolerance"] or rep["index_mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
