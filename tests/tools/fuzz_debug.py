#!/usr/bin/env python
"""Diagnose one case of fuzz_parity.py: python tests/tools/fuzz_debug.py <seed> -- prints, per gradient tensor, the rows
with the largest error together with the Gaussian's footprint / opacity / tile rectangle (test infrastructure)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd"), os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, p)
import parity as pa                      # noqa: E402
import fuzz_parity                       # noqa: E402


def main():
    seed = int(sys.argv[1])
    dev = torch.device("cuda:0")
    case = fuzz_parity.make_case(seed)
    scene, cam, bg, gc, gd, kw, desc = case
    print("case", desc)
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd, **kw)
    hip = pa.run_hip(scene, cam, bg, gc, gd, dev, **kw)
    print("indices", pa.check_indices(hip, oo))
    st = pa.compare(hip, oo, og, do_depth=kw["do_depth"])
    for k, v in st.items():
        print(k, v)
    geom = oo.geom
    per_tile = oo.binning.ranges[:, 1] - oo.binning.ranges[:, 0]
    print("tiles", geom.grid, "L", oo.binning.num_rendered, "longest list", per_tile.max())
    for k, g in og.items():
        h = hip["grads"][k].double().reshape(g.shape[0], -1)
        o = g.double().reshape(g.shape[0], -1)
        err = (h - o).abs().max(dim=1).values
        scale = o.abs().max()
        top = torch.argsort(err, descending=True)[:4]
        print(f"--- d_{k}: scale {scale:.4e}")
        for i in top.tolist():
            print(f"   row {i}: err {err[i]:.3e} ({err[i] / scale:.2e} of max) hip {h[i][:4].tolist()} oracle {o[i][:4].tolist()} "
                  f"radius {int(geom.radii[i])} opacity {float(scene.opacities[i]):.4f} depth {float(geom.depth[i]):.3f} "
                  f"rect {geom.rect_min[i].tolist()}-{geom.rect_max[i].tolist()} scales {scene.scales[i].tolist()}")
    # how many pixels saturate / what n_contrib looks like
    nc = hip["views"]["n_contrib"].numpy()
    fT = hip["views"]["final_T"].numpy()
    print("final_T min", fT.min(), "pixels with T < 1e-3:", int((fT < 1e-3).sum()), "of", fT.size, "; n_contrib max", nc.max())
    bad_nc = int((nc != oo.n_contrib).sum())
    print("n_contrib mismatches vs oracle:", bad_nc, "fragile pixels", int(oo.fragile.sum()))
    dT = np.abs(fT - oo.final_T) if hasattr(oo, "final_T") else None
    if dT is not None:
        print("final_T max abs diff", dT.max())


if __name__ == "__main__":
    main()
