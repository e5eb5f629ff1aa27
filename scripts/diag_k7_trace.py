#!/usr/bin/env python
"""Diagnostic (GPU box, a library built with the K7 timeline patch -- hgs_debug_set_trace exported): one drop-in step of
the metric frame with every wave of render_bwd_quad_kernel writing 100 MHz timestamps (kernel start, prologue done, per
batch: staged / inner loop done / stored, end) + its hardware id.  Prints where a wave's life goes, by round of the launch.
    python scripts/diag_k7_trace.py [scene: metric|trained] [out.npz]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hierarchical-3d-gaussians_amd"))
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from hgs import _lib, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "metric"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"k7_trace_{which}.npz")
dev = torch.device("cuda", 0)
W, H = 1920, 1080
cam0 = synth.make_camera(W, H)
scene = (synth.make_scene(1_000_000, cam0, seed=0) if which == "metric" else synth.make_scene_trained_scale(375_000, cam0, seed=0)).to(dev)
gc, gd = (t.to(dev) for t in synth.upstream_grads(H, W, seed=1))
params = dict(means3D=scene.means3D, shs=scene.shs, opacities=scene.opacities, scales=scene.scales, rotations=scene.rotations)
for t in params.values():
    t.requires_grad_(True)
cams = [synth.orbit_camera(W, H, j, 8, radius=0.05, tilt=0.004) for j in range(8)]
d = bench.DropIn(dgr, params, scene.sh_degree, [bench._settings(dgr, c, dev) for c in cams], gc, gd, dev)
for _ in range(10):
    d.step()
torch.cuda.synchronize()
lib = _lib.lib()
lib.hgs_debug_set_trace.argtypes = [C.c_void_p]
NT = ((W + 15) // 16) * ((H + 15) // 16)
NB = ((NT + 7) // 8) * 8
buf = torch.zeros(NB * 64, dtype=torch.int64, device=dev)
buf6 = torch.zeros(NB * 64, dtype=torch.int64, device=dev)
assert lib.hgs_debug_set_trace(C.c_void_p(buf.data_ptr())) == 0
if hasattr(lib, "hgs_debug_set_trace6"):
    lib.hgs_debug_set_trace6.argtypes = [C.c_void_p]
    assert lib.hgs_debug_set_trace6(C.c_void_p(buf6.data_ptr())) == 0
d.step()
torch.cuda.synchronize()
assert lib.hgs_debug_set_trace(C.c_void_p(0)) == 0
if hasattr(lib, "hgs_debug_set_trace6"):
    assert lib.hgs_debug_set_trace6(C.c_void_p(0)) == 0
np.savez_compressed(out, trace=buf.cpu().numpy().reshape(NB, 64), trace6=buf6.cpu().numpy().reshape(NB, 64))


def report(tr, which):

    ok = tr[:, 1] > 0
    tr = tr[ok]
    if not len(tr):
        return
    t0 = tr[:, 1].min()
    nev = tr[:, 62].astype(int)                       # index of the last timestamp slot
    start = (tr[:, 1] - t0) / 100.0                   # us
    end = np.array([tr[i, min(nev[i], 61)] for i in range(len(tr))])
    end = (end - t0) / 100.0
    total = tr[:, 63] >> 32
    print(f"{which}: {len(tr)} waves, launch spans {end.max():.1f} us; list entries per tile mean {total.mean():.0f} max {total.max()}")
    hw = tr[:, 0] & 0xffffffff
    xcc = (tr[:, 0] >> 32) & 0xf
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print("distinct (xcc, se, sh, cu):", len(np.unique(key)), " waves per CU min/mean/max:", np.bincount(np.unique(key, return_inverse=True)[1]).min(),
          len(tr) / len(np.unique(key)), np.bincount(np.unique(key, return_inverse=True)[1]).max())
    ks = key * 4 + simd
    cnt = np.bincount(np.unique(ks, return_inverse=True)[1])
    print("distinct SIMDs:", len(cnt), " tiles per SIMD min/mean/max:", cnt.min(), round(cnt.mean(), 2), cnt.max(), " histogram:", np.bincount(cnt))
    first = start < 20.0
    print(f"waves starting in the first 20 us: {first.sum()};  their lifetime: mean {np.mean(end[first] - start[first]):.1f} us "
          f"(p10 {np.percentile(end[first] - start[first], 10):.1f}, p90 {np.percentile(end[first] - start[first], 90):.1f});  "
          f"later waves: {np.sum(~first)}, lifetime mean {np.mean(end[~first] - start[~first]):.1f} us")
    print("start time percentiles (us):", [round(float(np.percentile(start, p)), 1) for p in (0, 10, 25, 50, 75, 90, 100)])
    print("end   time percentiles (us):", [round(float(np.percentile(end, p)), 1) for p in (0, 10, 25, 50, 75, 90, 100)])
    # phases: prologue = slot2 - slot1; per batch: staging = s[3+3b+1] - s[3+3b], loop = s[3+3b+2] - s[3+3b+1], store+sync = next - s[3+3b+2]
    for name, sel in (("first-round waves", first), ("later waves", ~first)):
        pro, stg, loop, sto, nb = [], [], [], [], []
        for i in np.flatnonzero(sel):
            r = tr[i]
            n_b = (nev[i] - 4) // 3
            if n_b <= 0:
                continue
            pro.append((r[2] - r[1]) / 100.0)
            b = r[3:3 + 3 * n_b + 1].astype(np.int64)
            stg.append(sum(b[1 + 3 * j] - b[3 * j] for j in range(n_b)) / 100.0)
            loop.append(sum(b[2 + 3 * j] - b[1 + 3 * j] for j in range(n_b)) / 100.0)
            sto.append(sum(b[3 + 3 * j] - b[2 + 3 * j] for j in range(n_b)) / 100.0)
            nb.append(n_b)
        if pro:
            print(f"{name}: prologue {np.mean(pro):.1f} us; per wave over {np.mean(nb):.1f} batches: staging {np.mean(stg):.1f}, "
                  f"inner loops {np.mean(loop):.1f}, store + barrier {np.mean(sto):.1f} us")
    # occupancy over time: waves alive per 10 us
    edges = np.arange(0, end.max() + 10, 10.0)
    alive = [(np.sum((start <= e) & (end > e))) for e in edges]
    print("waves alive at t = 0, 10, 20 ... us:", alive)


report(buf.cpu().numpy().reshape(NB, 64), which + " K7")
report(buf6.cpu().numpy().reshape(NB, 64), which + " K6")
