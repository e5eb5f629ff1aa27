#!/usr/bin/env python
"""Tuning aid (GPU box): where does the HOST time of one drop-in fwd+bwd go?

    python scripts/diag_host_floor.py [P] [W] [H]

(1) wall time per step of bench.DropIn at a size whose GPU time is negligible (default 2 000 Gaussians at 256x256): the
    host floor of the call path -- Python, ctypes, torch allocator, autograd engine, kernel launches;
(2) the same loop under cProfile, top functions by own time;
(3) the 300 k / 1080p configuration (BASELINE configs[1]): step time next to the sum of its GPU stage times."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
import torch

import bench
import diff_gaussian_rasterization as dgr
from hgs import _lib, synth

dev = torch.device("cuda:0")


def make(P, W, H, n_cams=8):
    cams = [synth.orbit_camera(W, H, k, n_cams) for k in range(n_cams)]
    scene = synth.make_scene(P, cams[0], seed=0)
    params = {k: getattr(scene, k).to(dev).contiguous().requires_grad_(True)
              for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    gc, gd = synth.upstream_grads(H, W)
    return bench.DropIn(dgr, params, 3, [bench._settings(dgr, c, dev) for c in cams], gc.to(dev), gd.to(dev), dev)


def wall(step, n, warm=20):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def host_only(step, n):
    """Host time of the step itself: perf_counter around each call, no device sync inside the loop."""
    torch.cuda.synchronize()
    acc = 0.0
    for _ in range(n):
        t0 = time.perf_counter()
        step()
        acc += time.perf_counter() - t0
    torch.cuda.synchronize()
    return acc / n * 1e3


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    d = make(P, W, H)
    print(f"[tiny {P} @ {W}x{H}] wall per fwd+bwd: {wall(d.step, 400):.4f} ms   (host floor of the call path)")
    print(f"[tiny] host time inside step(): {host_only(d.step, 400):.4f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        d.step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
    # forward alone / backward alone
    rast, pr_ = d.rasts[0], d.params

    def fwd_only():
        with torch.no_grad():
            rast(means3D=pr_["means3D"], means2D=None, shs=pr_["shs"], colors_precomp=None, opacities=pr_["opacities"],
                 scales=pr_["scales"], rotations=pr_["rotations"], cov3D_precomp=None)
    print(f"[tiny] forward only (no_grad), wall: {wall(fwd_only, 400):.4f} ms")
    lib = _lib.lib()
    sz = [__import__('ctypes').c_size_t() for _ in range(4)]
    t0 = time.perf_counter()
    for _ in range(10000):
        lib.hgs_raster_ws_sizes(P, W, H, 0, *[__import__('ctypes').byref(x) for x in sz])
    print(f"one ctypes call (hgs_raster_ws_sizes): {(time.perf_counter() - t0) / 10000 * 1e6:.2f} us")
    t0 = time.perf_counter()
    for _ in range(10000):
        torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    print(f"one torch.empty (1 MiB, cached): {(time.perf_counter() - t0) / 10000 * 1e6:.2f} us")
    d2 = make(300_000, 1920, 1080)
    _lib.timing_enable(True)
    for _ in range(10):
        d2.step()
    torch.cuda.synchronize()
    _lib.timing_read(True)
    for _ in range(40):
        d2.step()
    torch.cuda.synchronize()
    tm = _lib.timing_read(True)
    _lib.timing_enable(False)
    stage_sum = sum(ms / calls for ms, calls in tm.values() if calls)
    w = wall(d2.step, 100)
    print(f"[300 k @ 1080p] wall per fwd+bwd {w:.4f} ms; host inside step() {host_only(d2.step, 100):.4f} ms; "
          f"GPU stage sum {stage_sum:.4f} ms")


if __name__ == "__main__":
    main()
