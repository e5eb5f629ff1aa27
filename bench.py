#!/usr/bin/env python
"""bench.py -- fwd+bwd frames/s of the rasterizer hot path at 1080p on 1 M synthetic Gaussians
(BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one optimizer step's worth of rasterizer work on every rank: `--views-per-step` (default 8)
views, each a full GaussianRasterizer forward + backward through the C ABI (preprocess, binning, sorts,
compositing, compositing backward, preprocess backward) with inputs resident in HBM, their gradients
accumulated in place into one flat 59*P-float bucket.  The SH-dependent ends of the k views are batched: their
colours come from ONE pass over the coefficients (hgs_sh_colors_batched -- the HIP form of the reference's
convert_SHs_python route, gaussian_renderer/__init__.py:84-89; the rasterizer then takes colors_precomp) and dL/dSH
from ONE pass (hgs_sh_colors_batched_bwd); `--no-batched-sh-forward` evaluates SH inside every rasterizer call (then
only the backward is batched), `--no-deferred-sh` nothing.  The backwards are enqueued on a second HIP stream
(`--no-stream-overlap`: on the forwards' stream), so that the HBM-bound kernels of one view run next to the ALU-bound
compositing kernels of the next; the step ends when both streams have drained.  With N > 1 every rank renders different views of
the same replicated Gaussians and the step ends with ONE RCCL all-reduce of that bucket (per-view data
parallelism with gradient accumulation).  The per-rank work is the same for every N
(weak scaling); value = N * views_per_step * steps / max-over-ranks time.  The 236 MB all-reduce costs
about as much as one view's compute on xGMI, hence the accumulation window (k = 1 is available; k = 8 keeps the
exposed all-reduce near 13 % of a step at 8 GPUs and gives the two-stream schedule 7 overlapped view pairs per step).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "hierarchical-3d-gaussians_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DOMINANT = "render_bwd"  # the kernel the roofline object describes (largest share of the frame, profiles/)


def algorithmic_bytes(P, V, L, N, T, M, depth=True, k=1, deferred_sh=False, sh_forward=False):
    """SURVEY.md §8(d) byte model, per stage, for the measured P (Gaussians), V (visible), L (tile
    instances), N (pixels), T (tiles), M (SH coefficients).  Each boundary tensor is counted once
    read / once written; irreducible intermediates once written + once read; the sort as one pass."""
    rec, inst = 64, 48
    ch = 4 if depth else 3
    b = {}
    b["preprocess_fwd"] = P * 44 + V * 12 * M + 4 * P + V * (rec + 12) + 8 * P
    b["scan"] = 8 * (P // 256 + 1)
    b["duplicate_keys"] = 12 * P + 8 * L
    b["tile_sort"] = 8 * L + 4 * L + 8 * T        # (tile id, Gaussian id) pairs in, ids grouped by tile out, ranges
    b["tile_depth_sort"] = 8 * T + 4 * L + 4 * L + 4 * L   # ids in, depth gather, ids out
    b["tile_ranges"] = 0                         # ranges come out of the tile binning (radix fallback only)
    b["render_fwd"] = 8 * T + 4 * L + rec * L + 4 * ch * N + 8 * N
    b["memset_bwd"] = inst * L
    b["render_bwd"] = 8 * T + 4 * L + rec * L + 4 * ch * N * 2 + inst * L
    b["preprocess_bwd"] = inst * L + P * 44 + V * 12 * M + 12 * P + P * (56 + 12 * M)
    b["memset_bwd"] = 0                          # the forward's compositing kernel clears the scratch on the side
    if deferred_sh:
        # per view: geometry chain only (SH neither read nor written); per STEP: one pass over the coefficients
        b["preprocess_bwd"] = inst * L + P * 44 + 12 * P + P * 56 + 12 * P
        b["sh_bwd_batched"] = P * 12 * M * 2 + k * (16 * P) + 24 * P
    if sh_forward:
        # the colours come from one batched pass per step; K1 reads 12 B of colour instead of 12 M B of coefficients
        b["preprocess_fwd"] = P * 44 + V * 12 + 4 * P + V * (rec + 12) + 8 * P
        b["sh_colors_batched"] = P * 12 * M + 12 * P + k * (13 * P)
    return b


def cpu_baseline(scene, cam, bg, gc, gd, L_total, seed=3, n_tiles=1024):
    """Naive PyTorch-CPU per-pixel alpha blend (= the oracle, float32) timed on the host cores on a
    bounded sample: the per-Gaussian stage for the whole scene + dense blending fwd+bwd of `n_tiles`
    randomly chosen tiles; the blend time is scaled by tile-instance count to a full frame."""
    import numpy as np
    from oracle import raster_oracle as ro
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    kw = dict(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
              bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
              sh_degree=scene.sh_degree, campos=cam.camera_center, dtype=torch.float32)

    def run(tiles):
        req = lambda t: t.clone().requires_grad_(True)
        m3, sc, rot, op, sh = map(req, (scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs))
        m2 = torch.zeros(scene.P, 3, requires_grad=True)
        t0 = time.perf_counter()
        out = ro.rasterize(m3, m2, sh, None, op, sc, rot, None, tiles=tiles, **kw)
        loss = (out.color * gc).sum() + (out.invdepth * gd).sum()
        loss.backward()
        return time.perf_counter() - t0, out

    T = ((cam.image_width + 15) // 16) * ((cam.image_height + 15) // 16)
    rng = np.random.default_rng(seed)
    tiles = rng.choice(T, size=min(n_tiles, T), replace=False).tolist()
    small = sorted(tiles[:max(1, len(tiles) // 4)])
    tiles = sorted(tiles)
    # two sample sizes -> t = a + b * instances: a = per-Gaussian stage (fwd+bwd, whole scene),
    # b = blend cost per tile instance
    t1, out1 = run(small)
    t2, out2 = run(tiles)
    rg = out2.binning.ranges
    L1 = int((rg[small, 1] - rg[small, 0]).sum())
    L2 = int((rg[tiles, 1] - rg[tiles, 0]).sum())
    b = max((t2 - t1) / max(L2 - L1, 1), 0.0)
    a = max(t1 - b * L1, 0.0)
    t_frame = a + b * L_total
    return {"value": 1.0 / t_frame, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (naive PyTorch-CPU dense per-pixel blend, float32, {torch.get_num_threads()} threads): "
                      f"fwd+bwd of the per-Gaussian stage for the whole scene plus {len(small)} and {len(tiles)} "
                      f"of {T} tiles ({L1} / {L2} of {L_total} tile instances) in {t1:.2f} s / {t2:.2f} s; "
                      f"linear fit {a:.2f} s + {b * 1e6:.3f} us/instance extrapolated to the full frame"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--variant", type=int, default=0, help="render-kernel strip layout (0 = library default)")
    ap.add_argument("--views-per-step", type=int, default=8,
                    help="views rendered (fwd+bwd) per rank between two gradient all-reduces")
    ap.add_argument("--no-batched-sh-forward", action="store_true",
                    help="k > 1: evaluate the SH colours inside every view's rasterizer call instead of once per step "
                         "for all k views (then only the SH backward is batched, see --no-deferred-sh)")
    ap.add_argument("--no-stream-overlap", action="store_true",
                    help="enqueue the backwards on the forwards' stream (default: a second HIP stream, so that the "
                         "HBM-bound stages of one view overlap with the ALU-bound compositing of the next)")
    ap.add_argument("--no-deferred-sh", action="store_true",
                    help="per-view SH backward (accumulating) instead of one batched pass per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--cpu-baseline-only", type=int, default=0, metavar="L",
                    help="internal: time the CPU oracle for a frame with L tile instances, print JSON, exit")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        from hgs import synth
        cam = synth.make_camera(args.width, args.height)
        scene = synth.make_scene(args.gaussians, cam, seed=0)
        gc, gd = synth.upstream_grads(args.height, args.width, seed=1)
        print(json.dumps(cpu_baseline(scene, cam, torch.zeros(3), gc, gd, args.cpu_baseline_only)))
        return

    from hgs import _lib, dp, synth
    import diff_gaussian_rasterization as dgr

    rank, local, world = dp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rasterizer has no CPU path (see oracle/ for the checker)")
    if _lib.lib().hgs_device_count() < 1:
        raise SystemExit("libhgs.so sees no HIP device")
    dev = torch.device("cuda", local if torch.cuda.device_count() > local else 0)
    torch.cuda.set_device(dev)
    W, H, P = args.width, args.height, args.gaussians

    base_cam = synth.make_camera(W, H)
    scene_cpu = synth.make_scene(P, base_cam, seed=0)            # same Gaussians on every rank
    k = max(1, args.views_per_step)
    # every (rank, slot) gets its own camera: the canonical pose perturbed by a few centimetres / milliradians,
    # so each view still sees (almost) all of the 1 M Gaussians -- the BASELINE workload -- but no two views agree
    n_views = world * k
    cams_cpu = [base_cam if n_views == 1 else synth.orbit_camera(W, H, rank * k + j, n_views, radius=0.05, tilt=0.004)
                for j in range(k)]
    gc_cpu, gd_cpu = synth.upstream_grads(H, W, seed=1)
    bg_cpu = torch.zeros(3)
    scene = scene_cpu.to(dev)
    gc, gd, bg = gc_cpu.to(dev), gd_cpu.to(dev), bg_cpu.to(dev)
    e_i = torch.empty(0, dtype=torch.int32, device=dev)
    e_f = torch.empty(0, dtype=torch.float32, device=dev)
    rasts = []
    for cam_c in cams_cpu:
        cam = cam_c.to(dev)
        rs = dgr.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0,
            viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=scene.sh_degree,
            campos=cam.camera_center, prefiltered=False, debug=False, do_depth=True, render_indices=e_i,
            parent_indices=e_i, interpolation_weights=e_f, num_node_kids=e_i)
        rasts.append(dgr.GaussianRasterizer(rs))
    dgr._RasterizeGaussians.variant = args.variant
    params = dict(means3D=scene.means3D, shs=scene.shs, opacities=scene.opacities, scales=scene.scales,
                  rotations=scene.rotations)
    for t in params.values():
        t.requires_grad_(True)
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    # the backward writes (first view) / accumulates (later views) straight into the flat bucket
    bucket = dp.GradBucket({kk: tuple(v.shape) for kk, v in params.items()}, dev)
    dgr._RasterizeGaussians.grad_buffers = bucket.views
    # forwards on the current stream, backwards on a second one: view j's backward (and its HBM-bound preprocess /
    # gradient kernels) runs next to view j+1's forward
    overlap = not args.no_stream_overlap and k > 1      # one view per step: nothing to run next to
    dgr._RasterizeGaussians.backward_stream = torch.cuda.Stream(device=dev) if overlap else None
    info = {"L": 0, "V": 0}

    # k > 1: the SH part of the k backwards (81 % of the gradient bytes) is left pending and done for all k views in one
    # pass over the coefficients at the end of the step (hgs_raster_sh_bwd_batched)
    dgr._RasterizeGaussians.defer_sh_backward = k > 1 and not args.no_deferred_sh and args.no_batched_sh_forward

    sh_fwd = bool(k > 1 and not args.no_batched_sh_forward)
    campos = [r.raster_settings.campos for r in rasts]
    raster_names = [kk for kk in params if kk != "shs"]

    def step_sh_forward():
        # colours of all k views in ONE pass over the coefficients (the reference's convert_SHs_python route, in HIP),
        # k rasterizations with colors_precomp, then ONE pass for dL/dSH and the view-direction part of dL/dmeans3D
        with torch.no_grad():
            rgbs, clamps = dgr.sh_colors_batched(params["means3D"], params["shs"], scene.sh_degree, campos)
        d_rgbs = []
        for j, rast in enumerate(rasts):
            dgr._RasterizeGaussians.grad_accumulate = j > 0
            rgb = rgbs[j].requires_grad_(True)
            color, radii, invd = rast(means3D=params["means3D"], means2D=means2D, colors_precomp=rgb,
                                      opacities=params["opacities"], scales=params["scales"],
                                      rotations=params["rotations"])
            info["L"] = color.grad_fn.num_rendered
            info["radii"] = radii
            g = torch.autograd.grad([color, invd], [params[kk] for kk in raster_names] + [means2D, rgb], [gc, gd])
            d_rgbs.append(g[-1])
        dgr.sh_colors_batched_backward(params["means3D"], params["shs"], scene.sh_degree, campos, clamps, d_rgbs,
                                       bucket.views["shs"], bucket.views["means3D"])
        dgr.wait_backward_stream()
        if world > 1:
            bucket.all_reduce()

    def step():
        if sh_fwd:
            return step_sh_forward()
        for j, rast in enumerate(rasts):
            dgr._RasterizeGaussians.grad_accumulate = j > 0
            color, radii, invd = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"],
                                      opacities=params["opacities"], scales=params["scales"],
                                      rotations=params["rotations"])
            info["L"] = color.grad_fn.num_rendered
            info["radii"] = radii
            torch.autograd.grad([color, invd], [params[kk] for kk in params] + [means2D], [gc, gd])
        if dgr._RasterizeGaussians.defer_sh_backward:
            dgr.finish_deferred_sh_backward()
        dgr.wait_backward_stream()
        if world > 1:
            bucket.all_reduce()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timing = (not args.no_stage_timing)
    barrier()
    if timing:
        # inside the timed region only the dominant kernel is bracketed by hipEvents (the roofline figure must
        # come from the timed steps themselves); the other stages are timed in a short extra pass afterwards so
        # that their 20 event records per view do not sit in the measured stream
        _lib.timing_read(reset=True)
        _lib.timing_enable(True, stages=[DOMINANT])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    stages = {}
    if timing:
        _lib.timing_enable(False)
        dom_ms = {k: (ms / max(c, 1)) for k, (ms, c) in _lib.timing_read(reset=True).items() if c}
        _lib.timing_enable(True)
        for _ in range(max(2, min(args.steps, 5))):
            step()
        barrier()
        _lib.timing_enable(False)
        stages = {k: (ms / max(c, 1)) for k, (ms, c) in _lib.timing_read(reset=True).items() if c}
        stages.update(dom_ms)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * k * args.steps / elapsed
        L = int(info["L"])
        V = int((info["radii"] > 0).sum().item())
        N, T, M = W * H, ((W + 15) // 16) * ((H + 15) // 16), scene.shs.shape[1]
        deferred = bool(dgr._RasterizeGaussians.defer_sh_backward) or sh_fwd
        ab = algorithmic_bytes(P, V, L, N, T, M, k=k, deferred_sh=deferred, sh_forward=sh_fwd)
        # per frame: the batched SH passes run once per step of k views
        per_step = ("sh_bwd_batched", "sh_colors_batched")
        total_bytes = sum(v for kk_, v in ab.items() if kk_ not in per_step) + sum(ab.get(kk_, 0) for kk_ in per_step) / k
        result = {
            "metric": "fwd+bwd frames/s @1080p, 1M Gaussians", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{P} frustum-filling synthetic Gaussians (SURVEY §8(d) spec, seed 0), "
                                   f"{W}x{H}, SH degree 3, depth channel on, fwd+bwd through GaussianRasterizer",
                       "gaussians": P, "visible": V, "tile_instances": L, "width": W, "height": H,
                       "views_per_step_per_gpu": k,
                       "parallelism": f"per-view dp{world}, {k} views per rank per step accumulated in place" +
                                      (", SH backward batched over the views" if dgr._RasterizeGaussians.defer_sh_backward else "") +
                                      (", SH colours and their backward batched over the views" if sh_fwd else "") +
                                      (", backwards on a second HIP stream" if overlap else "") +
                                      (", one RCCL all-reduce of the 59P-float grad bucket per step" if world > 1 else ""),
                       "render_variant": args.variant},
            "algorithmic_bytes_per_frame": total_bytes,
            "ms_per_frame_per_gpu": ms_per_step / k,
            "frame_hbm_frac": total_bytes * (k * args.steps / elapsed) / 1e9 / HBM_PEAK_GBS,
        }
        if stages:
            dom = DOMINANT if DOMINANT in stages else max(stages, key=stages.get)
            achieved = ab[dom] / (stages[dom] * 1e-3) / 1e9
            traffic = None
            pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc_path):
                try:
                    traffic = json.load(open(pmc_path)).get(dom)
                except Exception:
                    traffic = None
            result["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                                  "avg_ms": stages[dom], "algorithmic_bytes": ab[dom],
                                  "note": "compositing kernels are VALU/LDS-bound (gather/blend, no MFMA); "
                                          "the HBM fraction is reported because the metric mandates it"}
            # The compositing kernels are VALU-issue-bound: add the vector-ALU view next to the mandated HBM one.
            # Peak: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles per SIMD at 2.4 GHz (packed
            # v_pk_*_f32 instructions count once and do two flops-lanes of work).
            valu_path = os.path.join(ROOT, "profiles", "pmc_valu.json")
            if os.path.exists(valu_path):
                try:
                    pv = json.load(open(valu_path)).get(dom)
                    if pv:
                        peak = 256 * 4 * 2.4e9 / 4
                        rate = pv["valu_insts_per_launch"] / (stages[dom] * 1e-3)
                        result["roofline"]["valu"] = {
                            "insts_per_launch": pv["valu_insts_per_launch"], "achieved_ginst_s": rate / 1e9,
                            "peak_ginst_s": peak / 1e9, "frac": rate / peak,
                            "source": "SQ_INSTS_VALU from rocprofv3 --pmc (profiles/pmc_valu.json), time from this run"}
                except Exception:
                    pass
            result["stages_ms"] = stages
            result["stages_gbs"] = {k: ab[k] / (v * 1e-3) / 1e9 for k, v in stages.items() if k in ab}
        if world == 1 and not args.no_cpu_baseline:
            # separate process + hard time limit: the baseline must never take the GPU number down with it
            import subprocess
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", str(L),
                                     "--gaussians", str(P), "--width", str(W), "--height", str(H)],
                                    capture_output=True, text=True, timeout=420)
                result["cpu_baseline"] = json.loads(cp.stdout.strip().splitlines()[-1])
                result["cpu_baseline"]["sample"] += " (canonical camera)"
            except Exception as e:
                result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count() or 1,
                                          "kind": "port", "sample": f"not measured: {e!r}"}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
