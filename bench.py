#!/usr/bin/env python
"""bench.py -- fwd+bwd frames/s of the rasterizer hot path at 1080p on 1 M synthetic Gaussians
(BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Two schedules are measured; both run every kernel of the path (preprocess, binning, sorts, compositing, compositing
backward, preprocess backward) through the C ABI with inputs resident in HBM:

* DROP-IN (the headline at N = 1): exactly the call train_single.py makes -- one view per step,
  ``GaussianRasterizer(raster_settings)(means3D=, means2D=, shs=, opacities=, scales=, rotations=)`` with the kwargs
  of gaussian_renderer/__init__.py:105-113, ``loss.backward()`` semantics (fresh ``.grad`` tensors every step, as after
  ``optimizer.zero_grad(set_to_none=True)``), ONE stream, no opt-in API.  A step = one view.
* BATCHED (``"batched"`` in the JSON line; the schedule of the N > 1 runs): `--views-per-step` (default 8) views per
  rank per optimizer step through a ``RasterContext``: gradients accumulated in place into one flat 59*P-float
  bucket, SH colours of the k views from ONE pass over the coefficients (hgs_sh_colors_batched -- the HIP form of the
  reference's convert_SHs_python route, gaussian_renderer/__init__.py:84-89) and dL/dSH from ONE pass
  (hgs_sh_colors_batched_bwd), backwards on a second HIP stream next to the following view's forward.  With N > 1
  every rank renders different views of the same replicated Gaussians and the step ends with the RCCL all-reduce of
  that bucket, issued in two parts ((opacity, scale, rotation) while the batched SH backward still runs, then
  (position, SH)) -- per-view data parallelism with gradient accumulation, SURVEY.md §8(e).  Per-rank work is the same for
  every N (weak scaling); value = N * views_per_step * steps / max-over-ranks time.

At N = 1 ``value`` is the DROP-IN number and ``batched.value`` the other one; at N > 1 ``value`` == ``batched.value``
(the data-parallel schedule).  To compute scaling efficiency compare ``batched.value`` across N.

Both schedules cycle through 8 orbit cameras per rank (the instance count L changes from step to step, as in training:
the speculative single-call forward sizes its workspace from the previous view of the shape; ``capacity_misses``
counts the calls that overflowed it and were finished on the two-stage path).

At N = 1 the same JSON line carries ``extra``: the other BASELINE.json configurations timed by the same process with
the same schema (value, ms_per_step, stages_ms, roofline from THAT run's P / V / L):
  config2_300k            300 k Gaussians at 1080p, the train_single.py call shape (configs[1])
  heavy_1m                1 M Gaussians with s_px in [1, 8] (SURVEY App. C "heavy": L ~ 4.9 M, ~600 instances per tile)
  trained_like_10m        375 k Gaussians with the statistics of a scene train_single.py trained at 1080p (L ~ 10 M, lists > 3 000)
  trained_cut_10m         the same in the row order of a hierarchy cut (big nodes side by side)
  config3_train_post      train_post.py-shaped step on a merged 2-chunk hierarchy (configs[2])
  config5_50m_4k_render   50 M-node hierarchy, cut + weights + 3840x2160 render per frame as render_hierarchy.py (configs[4])
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
VALU_GUIDE_GINST_S = 256 * 4 * 2.4 / 2.0   # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD


def algorithmic_bytes(P, V, L, N, T, M, depth=True, k=1, deferred_sh=False, sh_forward=False):
    """SURVEY.md §8(d) byte model, per stage, for the measured P (Gaussians), V (visible), L (tile
    instances), N (pixels), T (tiles), M (SH coefficients).  Each boundary tensor is counted once
    read / once written; irreducible intermediates once written + once read; the sort as one pass."""
    rec, inst = 64, 40
    ch = 4 if depth else 3
    b = {}
    # K1: inputs, record, per-Gaussian arrays + SH block in, the record's three colour floats and the Jacobian row out
    b["preprocess_fwd"] = P * 44 + 4 * P + V * rec + 20 * P + V * 12 * M + 16 * P + V * (12 + 48)
    b["scan"] = 8 * (P // 256 + 1)
    b["duplicate_keys"] = 12 * P + 8 * L
    b["tile_sort"] = 8 * L + 4 * L + 8 * T        # (tile id, Gaussian id) pairs in, ids grouped by tile out, ranges
    b["tile_depth_sort"] = 8 * T + 4 * L + 4 * L + 4 * L   # ids in, depth gather, ids out
    b["tile_ranges"] = 0                         # ranges come out of the tile binning (radix fallback only)
    b["render_fwd"] = 8 * T + 4 * L + rec * L + 4 * ch * N + 8 * N
    b["render_bwd"] = 8 * T + 4 * L + rec * L + 4 * ch * N * 2 + inst * L
    b["preprocess_bwd"] = inst * L + P * 44 + V * 12 * M + 12 * P + P * (56 + 12 * M)
    b["memset_bwd"] = 0                          # no clearing: the backward's compositing kernel writes every record
    if deferred_sh:
        # per view: geometry chain only (SH neither read nor written); per STEP: one pass over the coefficients
        b["preprocess_bwd"] = inst * L + P * 44 + 12 * P + P * 56 + 12 * P
        b["sh_bwd_batched"] = P * 12 * M * 2 + k * (16 * P) + 24 * P
    if sh_forward:
        # the colours come from one batched pass per step; K1 reads 12 B of colour instead of 12 M B of coefficients
        b["preprocess_fwd"] = P * 44 + V * 12 + 4 * P + V * (rec + 12) + 8 * P
        b["sh_colors_batched"] = P * 12 * M + 12 * P + k * (13 * P)
    return b


def survey_bytes(P, V, L, N, T, M):
    """SURVEY.md §8(d) ALGORITHMIC bytes, split per kernel exactly as the survey's fwd / bwd sums are written
    (boundary tensors once, intermediates once written + once read, 40-byte 2D record, 40-byte per-Gaussian 2D
    gradient, the sort as one pass).  This is the model ``roofline.achieved`` / ``roofline.frac`` use;
    ``algorithmic_bytes`` above additionally charges this implementation's 64-byte record and its 48-byte
    per-instance scratch and is reported under ``roofline.impl_*``."""
    b = {}
    b["preprocess_fwd"] = 44 * P + 4 * P + 28 * V + 12 * M * V + 12 * V     # inputs, radii, the 40-byte 2D record; SH block in
    b["duplicate_keys"] = 12 * L
    b["tile_sort"] = 12 * L + 4 * L                      # sort read + sorted ids written
    b["render_fwd"] = 4 * L + 40 * L + 16 * N + 8 * N + 8 * T
    b["render_bwd"] = 16 * N + 8 * N + 4 * L + 40 * L + 40 * V
    b["preprocess_bwd"] = 40 * V + 44 * P + 12 * M * V + P * (56 + 12 * M)
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
    return {"value": 1.0 / t_frame, "unit": "frames/s", "cores": torch.get_num_threads(),
            "host_cores": os.cpu_count(), "kind": "port",
            "cores_note": "threads capped at 32: the oracle issues thousands of small per-tile tensor operations and "
                          "torch's intra-op pool turns each into a fork-join over every thread -- on the 256-thread host "
                          "more threads make it SLOWER (tests/conftest.py measured 8x between 16 and 256 threads)",
            "sample": f"oracle (naive PyTorch-CPU dense per-pixel blend, float32, {torch.get_num_threads()} threads): "
                      f"fwd+bwd of the per-Gaussian stage for the whole scene plus {len(small)} and {len(tiles)} "
                      f"of {T} tiles ({L1} / {L2} of {L_total} tile instances) in {t1:.2f} s / {t2:.2f} s; "
                      f"linear fit {a:.2f} s + {b * 1e6:.3f} us/instance extrapolated to the full frame"}


def kernel_source_sha():
    """Hash of the sources libhgs.so is built from: committed PMC summaries carry the hash of the build they were
    collected on, and a summary from another build is not mixed into this run's line."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.h")) +
                   glob.glob(os.path.join(PKG, "csrc", "*.cpp")) + [os.path.join(PKG, "csrc", "Makefile"),
                                                                    os.path.join(ROOT, "include", "hgs.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "r", errors="replace") as fh:
            h.update(_code_only(fh.read()).encode())
    return h.hexdigest()[:16]


def _code_only(src):
    """The source without comments and blank lines (a reworded comment is not another build).  String literals are kept
    as they are; a comment marker inside one is rare enough in these sources that the simple scan is left alone: it
    could only make the hash change too often, never too rarely."""
    import re
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = []
    for line in src.splitlines():
        line = re.sub(r"(?<!:)//.*$", "", line) if "#" not in line[:1] else line
        line = line.rstrip()
        if line.strip():
            out.append(line)
    return "\n".join(out)


def _profile_json(name):
    """(content, path relative to the repo) of a committed rocprofv3 PMC summary, or (None, None)."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return json.load(f), os.path.join("profiles", name)
    except Exception:
        return None, None


PMC_PASSES = {       # one rocprofv3 run per counter group, each with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3)
    "SQ": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
           "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"],
    "FETCH": ["FETCH_SIZE"],
    "WRITE": ["WRITE_SIZE"],
}


def measure_pmc_live(args, N, L, time_limit=75):
    """HBM traffic and vector-ALU counters of THIS build on THIS box, collected while this command runs: one
    `rocprofv3 --pmc <group> --kernel-trace` child per counter group around a 3-step drop-in run of this very script at
    the same configuration (this process is idle meanwhile: nothing else touches HBM), summarised per kernel by
    scripts/pmc_summary.py and turned into bytes per launch by scripts/make_pmc_json.py (gfx950 corrections there).
    Returns (traffic, valu, note); (None, None, why) when rocprofv3 is absent, fails or exceeds its time limit -- the
    caller then falls back to the committed summary of the same build, if there is one."""
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, None, "this run is itself under a profiler (rocprofv3 environment inherited): no nested counter passes"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import make_pmc_json
        import pmc_summary
    except Exception as e:
        return None, None, f"scripts/ not importable: {e!r}"
    tmp = tempfile.mkdtemp(prefix="hgs_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
             "--no-stage-timing", "--no-secondary", "--schedule", "dropin", "--gaussians", str(args.gaussians),
             "--width", str(args.width), "--height", str(args.height)]
    t0 = time.perf_counter()
    merged = {}
    try:
        for name, counters in PMC_PASSES.items():
            out = os.path.join(tmp, name)
            cmd = [exe, "--pmc", *counters, "--kernel-trace", "-d", out, "-o", "pmc", "--", *child]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout=time_limit)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)       # the group this call started, nothing else
                proc.wait()
                return None, None, f"rocprofv3 --pmc {name} pass exceeded {time_limit} s"
            if proc.returncode != 0:
                return None, None, f"rocprofv3 --pmc {name} pass exited with {proc.returncode}"
            dbs = sorted(glob.glob(os.path.join(out, "**", "*.db"), recursive=True))
            if not dbs:
                return None, None, f"rocprofv3 --pmc {name} pass wrote no database"
            for kernel, cs in pmc_summary.load(dbs[0]).items():
                merged.setdefault(pmc_summary.short_name(kernel), {}).update(cs)
        traffic, valu = make_pmc_json.derive(merged, N, L)
        if DOMINANT not in traffic:
            return None, None, "the counter passes saw no launch of the dominant kernel"
        return traffic, valu, (f"measured in this run: rocprofv3 --pmc passes ({' | '.join(' '.join(c) for c in PMC_PASSES.values())}; "
                               f"--kernel-trace only) of 3 drop-in steps of this command in a child process, "
                               f"{time.perf_counter() - t0:.0f} s")
    except Exception as e:
        return None, None, f"live PMC collection failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class DropIn:
    """The reference's call shape, one view per step: ``GaussianRasterizer(raster_settings)(means3D, means2D, shs,
    colors_precomp=None, opacities, scales, rotations, cov3D_precomp=None)`` + ``backward`` with fresh ``.grad`` tensors
    (train_single.py:97,123; gaussian_renderer/__init__.py:105-113), cycling through the given views."""

    def __init__(self, dgr, params, sh_degree, settings_list, gc, gd, dev, dp_bucket=None, do_depth=True):
        self.params, self.gc, self.gd, self.dev, self.bucket, self.do_depth = params, gc, gd, dev, dp_bucket, do_depth
        self.rasts = [dgr.GaussianRasterizer(raster_settings=rs) for rs in settings_list]
        self.P = params["means3D"].shape[0]
        self.i = 0
        self.reset_stats()

    def reset_stats(self):
        self.L_sum, self.n, self.radii = 0, 0, {}

    def step(self):
        params = self.params
        for t in params.values():
            t.grad = None                                       # optimizer.zero_grad(set_to_none=True)
        j = self.i % len(self.rasts)
        self.i += 1
        screenspace_points = torch.zeros(self.P, 3, device=self.dev, requires_grad=True)   # gaussian_renderer/__init__.py:29
        color, radii, invd = self.rasts[j](means3D=params["means3D"], means2D=screenspace_points, shs=params["shs"],
                                           colors_precomp=None, opacities=params["opacities"], scales=params["scales"],
                                           rotations=params["rotations"], cov3D_precomp=None)
        self.L_sum += color.grad_fn.num_rendered
        self.n += 1
        self.radii[j] = radii
        if self.do_depth:
            torch.autograd.backward([color, invd], [self.gc, self.gd])   # loss.backward() with dL/dcolor, dL/dinvdepth given
        else:
            torch.autograd.backward([color], [self.gc])
        if self.bucket is not None:
            self.bucket.fill({kk: v.grad for kk, v in params.items()})
            self.bucket.all_reduce()

    def mean_L(self):
        return self.L_sum / max(self.n, 1)

    def mean_V(self):
        vs = [int((r > 0).sum().item()) for r in self.radii.values()]
        return sum(vs) / max(len(vs), 1)


def _settings(dgr, cam, dev, sh_degree=3, **over):
    # the camera's tensors live on the GPU, as the reference's Camera objects do (scene/cameras.py): building the settings
    # of a frame must not cost host-to-device copies (each is a blocking call; on a loaded / virtualised host their
    # wake-ups were seen to add milliseconds to a 10 ms frame)
    cache = cam.__dict__.setdefault("_bench_dev", {})
    if dev not in cache:
        cache[dev] = dict(bg=torch.zeros(3, device=dev), viewmatrix=cam.world_view_transform.to(dev),
                          projmatrix=cam.full_proj_transform.to(dev), campos=cam.camera_center.to(dev),
                          e_i=torch.empty(0, dtype=torch.int32, device=dev),
                          e_f=torch.empty(0, dtype=torch.float32, device=dev))
    c = cache[dev]
    kw = dict(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
              bg=c["bg"], scale_modifier=1.0, viewmatrix=c["viewmatrix"], projmatrix=c["projmatrix"], sh_degree=sh_degree,
              campos=c["campos"], prefiltered=False, debug=False, do_depth=True, render_indices=c["e_i"],
              parent_indices=c["e_i"], interpolation_weights=c["e_f"], num_node_kids=c["e_i"])
    kw.update(over)
    return dgr.GaussianRasterizationSettings(**kw)


def extra_dropin(name, scene_cpu, W, H, dev, measure, steps, warmup, what):
    """One more scene through the drop-in schedule (same call, same 8 cycled cameras, same timing)."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C as dgrC
    from hgs import synth
    scene = scene_cpu.to(dev)
    params = dict(means3D=scene.means3D, shs=scene.shs, opacities=scene.opacities, scales=scene.scales,
                  rotations=scene.rotations)
    for t in params.values():
        t.requires_grad_(True)
    gc, gd = (t.to(dev) for t in synth.upstream_grads(H, W, seed=1))
    cams = [synth.orbit_camera(W, H, j, 8, radius=0.05, tilt=0.004) for j in range(8)]
    d = DropIn(dgr, params, scene.sh_degree, [_settings(dgr, c, dev, scene.sh_degree) for c in cams], gc, gd, dev)
    miss0 = dgrC.stats["capacity_misses"]
    elapsed, stages = measure(d.step, steps, warmup, True, reset=d.reset_stats)
    P, M = scene.P, scene.shs.shape[1]
    L, V = int(round(d.mean_L())), int(round(d.mean_V()))
    N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    sb = survey_bytes(P, V, L, N, T, M)
    out = {"what": what, "metric": "fwd+bwd frames/s", "value": steps / elapsed, "unit": "frames/s", "steps": steps,
           "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
           "host_ms_per_step": getattr(measure, "last_host_ms", None),
           "stage_sum_ms": sum(stages.values()) if stages else None,
           "config": {"gaussians": P, "visible": V, "tile_instances": L, "width": W, "height": H, "schedule": "dropin",
                      "capacity_misses": dgrC.stats["capacity_misses"] - miss0},
           "algorithmic_bytes_per_frame": sum(sb.values()), "stages_ms": stages}
    if stages.get(DOMINANT):
        out["roofline"] = roofline_object(sb, stages, DOMINANT, "SURVEY.md §8(d): render_bwd = 24 N + 44 L + 40 V of this run")
    out["frame_hbm_frac"] = sum(sb.values()) * out["value"] / 1e9 / HBM_PEAK_GBS
    return out


def extra_train_post(dev, measure, steps, warmup, leaves=500_000):
    """BASELINE configs[2]: a train_post.py-shaped step (train_post.py:66-191) on a merged 2-chunk hierarchy at 1080p:
    threshold log-uniform in [0.005, 0.1] (:66-74), expand_to_size + get_interpolation_weights (:91-113), render_post
    with the interpolation done in the op (non-empty render_indices / parent_indices), L1 loss, backward, dense Adam over
    all hierarchy Gaussians (:37,191)."""
    import math
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C as dgrC
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
    from hgs import hierarchy, synth
    from hgs.optim import Adam
    W, H = 1920, 1080
    cam = synth.make_camera(W, H)
    full = synth.make_scene(leaves, cam, seed=0)
    left = full.means3D[:, 0] < 0
    h = hierarchy.merge_hierarchies([hierarchy.build_hierarchy(
        synth.Scene(full.means3D[m], full.scales[m], full.rotations[m], full.opacities[m], full.shs[m], 3))
        for m in (left, ~left)])
    nodes, boxes = h.nodes.to(dev), h.boxes.to(dev)
    G = h.xyz.shape[0]
    ri = torch.zeros(G, dtype=torch.int32, device=dev); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
    w = torch.zeros(G, device=dev); ns = torch.zeros(G, dtype=torch.int32, device=dev)
    attrs = dict(xyz=h.xyz, shs=h.shs, op=h.alpha.abs().reshape(-1, 1), sc=torch.exp(h.log_scales),
                 rot=torch.nn.functional.normalize(h.rots))
    params = {kk: torch.nn.Parameter(v.to(dev).contiguous()) for kk, v in attrs.items()}
    lrs = dict(xyz=1.6e-5, shs=2.5e-3, op=1e-3, sc=1e-6, rot=1e-5)
    opt = Adam([dict(params=[params[kk]], lr=lrs[kk], name=kk) for kk in params], lr=0.0, eps=1e-15)
    target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    mfull = torch.zeros(G, 3, device=dev, requires_grad=True)
    g = torch.Generator().manual_seed(3)
    vp_gpu, vp_cpu, zero3 = cam.camera_center.to(dev), cam.camera_center.cpu(), torch.zeros(3)
    st = {"cuts": [], "L": []}

    def step():
        limit = math.pow(2, torch.rand(1, generator=g).item() * (math.log2(0.1) - math.log2(0.005)) + math.log2(0.005))
        n = expand_to_size(nodes, boxes, limit, vp_gpu, zero3, ri, pi, ni)
        get_interpolation_weights(ni[:n], limit, nodes, boxes, vp_cpu, zero3, w, ns)
        rs = _settings(dgr, cam, dev, do_depth=False, interpolation_weights=w, num_node_kids=ns,
                       render_indices=ri[:n], parent_indices=pi)
        color, _, _ = dgr.GaussianRasterizer(rs)(means3D=params["xyz"], means2D=mfull, shs=params["shs"],
                                                 opacities=params["op"], scales=params["sc"], rotations=params["rot"])
        st["cuts"].append(n)
        st["L"].append(color.grad_fn.num_rendered)
        loss = (color - target).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step(None)                       # dense, as torch.optim.Adam in train_post.py:191

    miss0 = dgrC.stats["capacity_misses"]
    elapsed, stages = measure(step, steps, warmup, True, reset=lambda: (st["cuts"].clear(), st["L"].clear()))
    cuts, Ls = st["cuts"], st["L"]
    out = {"what": "BASELINE configs[2]: train_post.py-shaped step (log-uniform tau, cut + weights, in-op LOD render at "
                   "1080p, L1, backward, dense fused Adam) on a merged 2-chunk hierarchy",
           "metric": "optimiser steps/s", "value": steps / elapsed, "unit": "steps/s", "steps": steps, "warmup": warmup,
           "ms_per_step": elapsed / steps * 1e3,
           "config": {"hierarchy_nodes": G, "chunks": 2, "width": W, "height": H, "mean_cut": sum(cuts) / len(cuts),
                      "min_cut": min(cuts), "max_cut": max(cuts), "mean_tile_instances": sum(Ls) / len(Ls),
                      "capacity_misses": dgrC.stats["capacity_misses"] - miss0},
           "stages_ms": stages}
    if stages.get(DOMINANT):
        N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        Pm, Lm = int(sum(cuts) / len(cuts)), int(sum(Ls) / len(Ls))
        out["roofline"] = roofline_object(survey_bytes(Pm, Pm, Lm, N, T, 16), stages, DOMINANT,
                                          "SURVEY.md §8(d) render_bwd bytes at the mean cut / mean L of the timed steps "
                                          "(no depth channel: 4 N bytes fewer per image plane are not subtracted)")
    return out


def extra_config5(dev, steps, warmup, leaves=25_000_000, tau_px=3.0):
    """BASELINE configs[4]: a 50 M-node hierarchy (25 M leaves) resident in HBM, per frame what render_hierarchy.py:55-92
    does: expand_to_size at tau, get_interpolation_weights, render (in-op LOD path) at 3840x2160, forward only."""
    import diff_gaussian_rasterization as dgr
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
    from hgs import _lib, hierarchy, synth
    W, H = 3840, 2160
    cam = synth.make_camera(W, H)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = hierarchy.build_hierarchy_on_device(leaves, cam, dev, seed=0)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    G = h.nodes.shape[0]
    ri = torch.zeros(G, dtype=torch.int32, device=dev); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
    w = torch.zeros(G, device=dev); ns = torch.zeros(G, dtype=torch.int32, device=dev)
    cams = [synth.orbit_camera(W, H, j, 8, radius=0.05, tilt=0.004) for j in range(8)]
    vps = [(c.camera_center.to(dev), c.camera_center.cpu()) for c in cams]
    tau = (2 * tau_px + 1) * cam.tanfovx / (0.5 * W)                   # render_hierarchy.py:55-56
    m2 = torch.zeros(G, 3, device=dev)
    sc, zero3 = torch.exp(h.log_scales), torch.zeros(3)
    st = {"i": 0, "n": [], "L": [], "cut_s": 0.0}
    from diff_gaussian_rasterization import _C as dgrC
    from gaussian_hierarchy import _C as ghC
    ghC.set_viewpoint_cache(True)       # a viewer loop that keeps its Camera objects (restored by run_extras)

    def frame():
        j = st["i"] % len(cams); st["i"] += 1
        t_a = time.perf_counter()
        n = expand_to_size(h.nodes, h.boxes, tau, vps[j][0], zero3, ri, pi, ni)     # returns the count: host sync
        st["cut_s"] += time.perf_counter() - t_a
        get_interpolation_weights(ni[:n], tau, h.nodes, h.boxes, vps[j][1], zero3, w, ns)
        rs = _settings(dgr, cams[j], dev, do_depth=False, interpolation_weights=w, num_node_kids=ns,
                       render_indices=ri[:n], parent_indices=pi)
        with torch.no_grad():
            color, radii, _ = dgr.GaussianRasterizer(rs)(means3D=h.xyz, means2D=m2, shs=h.shs, opacities=h.alpha,
                                                         scales=sc, rotations=h.rots)
        st["n"].append(n)
        st["L"].append(dgr._C.stats["last_L"])
        return color

    for _ in range(warmup):
        frame()
    torch.cuda.synchronize()
    st["n"].clear(); st["L"].clear(); st["cut_s"] = 0.0
    miss0 = dgrC.stats["capacity_misses"]
    seg0 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)
    diag = bool(os.environ.get("HGS_BENCH_DIAG"))
    marks = []
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
        if diag:
            marks.append((time.perf_counter() - t0, torch.cuda.memory_stats(dev).get("segment.all.allocated", 0) - seg0,
                          torch.cuda.memory_stats(dev).get("reserved_bytes.all.current", 0) / 1e9))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if diag:
        import sys
        print("diag host time after each frame (ms), segments allocated so far, reserved GB:",
              [(round(a * 1e3, 2), b, round(c, 2)) for a, b, c in marks], "n", st["n"], file=sys.stderr)
    cut_ms, misses = st["cut_s"] / steps * 1e3, dgrC.stats["capacity_misses"] - miss0
    seg_allocs = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0) - seg0
    if os.environ.get("HGS_BENCH_DIAG"):      # phase by phase, a device sync after each
        import sys
        for it in range(8):
            j = it % len(cams)
            tt = [time.perf_counter()]
            n = expand_to_size(h.nodes, h.boxes, tau, vps[j][0], zero3, ri, pi, ni); torch.cuda.synchronize(); tt.append(time.perf_counter())
            get_interpolation_weights(ni[:n], tau, h.nodes, h.boxes, vps[j][1], zero3, w, ns); torch.cuda.synchronize(); tt.append(time.perf_counter())
            rs = _settings(dgr, cams[j], dev, do_depth=False, interpolation_weights=w, num_node_kids=ns, render_indices=ri[:n], parent_indices=pi)
            with torch.no_grad():
                dgr.GaussianRasterizer(rs)(means3D=h.xyz, means2D=m2, shs=h.shs, opacities=h.alpha, scales=sc, rotations=h.rots)
            tt.append(time.perf_counter()); torch.cuda.synchronize(); tt.append(time.perf_counter())
            print("diag frame", it, "expand/weights/render-host/render-sync ms", [round((b - a) * 1e3, 3) for a, b in zip(tt, tt[1:])],
                  "segments", torch.cuda.memory_stats(dev).get("segment.all.allocated", 0), file=sys.stderr)
        print("diag free-running ms/frame", elapsed / steps * 1e3, "timed-region device allocations", seg_allocs,
              "reserved GB", torch.cuda.memory_stats(dev).get("reserved_bytes.all.current", 0) / 1e9, file=sys.stderr)
    _lib.timing_read(reset=True)
    _lib.timing_enable(True)
    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    _lib.timing_enable(False)
    stages = {kk: (ms / max(c, 1)) for kk, (ms, c) in _lib.timing_read(reset=True).items() if c}
    nm = sum(st["n"][:steps]) / steps
    out = {"what": "BASELINE configs[4]: 50 M-node hierarchy resident in HBM (no streaming needed at 15 GB of 288 GB), per "
                   "frame expand_to_size + get_interpolation_weights + 3840x2160 render through the in-op LOD path "
                   "(forward only), as render_hierarchy.py:55-92",
           "metric": "rendered frames/s @ 3840x2160", "value": steps / elapsed, "unit": "frames/s", "steps": steps,
           "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
           "config": {"hierarchy_nodes": G, "leaves": leaves, "tau_px": tau_px, "mean_cut": nm, "width": W, "height": H,
                      "hierarchy_build_s": t_build, "resident_bytes": int(G * (59 * 4 + 28 + 32)),
                      "capacity_misses": misses,
                      # the camera centres' host copies are remembered on the tensor objects here (a viewer loop
                      # that keeps its Camera objects); a stock script pays one blocking 12-byte read per frame instead
                      "viewpoint_cache": True,
                      "device_allocations_in_timed_region": seg_allocs,   # hipMalloc calls of the caching allocator
                      "expand_to_size_ms": cut_ms},     # host time of the cut incl. the wait for everything enqueued before it
           "stages_ms": stages}
    if stages.get("render_fwd") and stages.get("preprocess_fwd"):
        N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        Pm, Lm = int(nm), int(sum(st["L"][:steps]) / steps)
        out["config"]["mean_tile_instances"] = Lm
        # forward-only bytes in the manner of SURVEY §8(d): the in-op LOD preprocess gathers node AND parent rows
        # (2 x 236 B) plus indices / weight / sibling count (16 B) per cut row; compositing without the depth plane
        sbk = {"preprocess_fwd": Pm * (2 * 236 + 16) + 4 * Pm + 40 * Pm, "render_fwd": 44 * Lm + 12 * N + 8 * N + 8 * T}
        dom = max(sbk, key=lambda kk: stages[kk])
        out["roofline"] = roofline_object(sbk, stages, dom, "forward only: preprocess_fwd = (2 x 236 + 16 + 44) B per cut "
                                          "row (node + parent rows gathered), render_fwd = 44 L + 20 N + 8 T")
    return out


def extra_config5_budgeted(dev, steps, warmup, leaves=25_000_000, tau_px=3.0, budget_mb=6000.0):
    """BASELINE configs[4] with its "VRAM-budgeted streaming LOD": the same 50 M-node hierarchy, but the attribute rows
    (11.8 GB) live in pinned HOST memory and the GPU holds `budget_mb` of them (hgs/residency.py; the reference viewer's
    --budget, README.md:233-235).  Per frame: cut + weights, the cut's rows made resident (misses fetched over PCIe by a
    kernel that reads the host arrays), 3840x2160 render through the in-op LOD path on the slot arrays.

    The camera FLIES: forward through the scene by 0.08 units per frame (the cut changes on every frame: rows are
    fetched and slots recycled continuously), with one jump sideways halfway (a burst).  A requested granularity that does
    not fit the budget settles at the finest one that does, as the reference's viewer.  For comparison the SAME path is
    then rendered from the fully resident hierarchy at the granularity each budgeted frame settled at
    (``resident_same_path``), and the last frame of both is compared bit for bit."""
    import numpy as np
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C as dgrC
    from gaussian_hierarchy import _C as ghC
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
    from hgs import hierarchy, synth
    from hgs.residency import BudgetedHierarchy
    W, H = 3840, 2160
    cam = synth.make_camera(W, H)
    h = hierarchy.build_hierarchy_on_device(leaves, cam, dev, seed=0)
    G = h.nodes.shape[0]
    full = dict(means3D=h.xyz, shs=h.shs, opacities=h.alpha, scales=torch.exp(h.log_scales), rotations=h.rots)
    t0 = time.perf_counter()
    bh = BudgetedHierarchy(full["means3D"].cpu(), full["shs"].cpu(), full["opacities"].cpu(), full["scales"].cpu(),
                           full["rotations"].cpu(), dev, budget_mb=budget_mb)
    t_host = time.perf_counter() - t0
    nodes, boxes = h.nodes, h.boxes
    total = warmup + steps
    jump = warmup + steps // 2

    def pose(k):            # world-to-camera translation of frame k: forward 0.08 per frame, 2 units sideways at the jump
        return np.array([-(2.0 if k >= jump else 0.0), 0.0, -0.08 * k])
    cams = [synth.make_camera(W, H, T=pose(k)) for k in range(total)]
    for c in cams:
        _settings(dgr, c, dev)                                  # camera tensors onto the GPU before the clock runs
    vps = [(c.camera_center.to(dev), c.camera_center.cpu()) for c in cams]
    tau = (2 * tau_px + 1) * cam.tanfovx / (0.5 * W)
    m2 = torch.zeros(max(bh.B, G), 3, device=dev)
    st = {"sel": []}

    def frame(k):
        sel = bh.select(nodes, boxes, tau, vps[k][0], vps[k][1])
        rs = _settings(dgr, cams[k], dev, do_depth=False, interpolation_weights=sel.weights, num_node_kids=sel.kids,
                       render_indices=sel.render_indices, parent_indices=sel.parent_indices)
        with torch.no_grad():
            color, radii, _ = dgr.GaussianRasterizer(rs)(means3D=bh.means3D, means2D=m2[:bh.B], shs=bh.shs,
                                                         opacities=bh.opacities, scales=bh.scales, rotations=bh.rotations)
        st["sel"].append((sel.n, sel.tau, sel.misses, sel.attempts))
        if k + 1 < total:       # the next pose is known here (a viewer extrapolates its camera): its rows cross PCIe under this render
            bh.prefetch(nodes, boxes, tau, vps[k + 1][0], vps[k + 1][1])
        return color

    prev_cache = ghC.set_viewpoint_cache(True)
    try:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        frame(0)                                                # cold start: the whole working set crosses PCIe
        torch.cuda.synchronize(); t_cold = time.perf_counter() - t0
        cold = dict(bh.stats)
        for k in range(1, warmup):
            frame(k)
        torch.cuda.synchronize()
        st["sel"].clear()
        f0, e0 = bh.stats["rows_fetched"], bh.stats["evictions"]
        miss0 = dgrC.stats["capacity_misses"]
        bh.profile_fetch, bh.fetch_events = True, []
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ends[0].record()
        t0 = time.perf_counter()
        for i in range(steps):
            color_b = frame(warmup + i)
            ends[i + 1].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        bh.profile_fetch = False
        sels = list(st["sel"])
        frame_ms = sorted(ends[i].elapsed_time(ends[i + 1]) for i in range(steps))
        fetch_rows = sum(m for m, _, _ in bh.fetch_events)
        fetch_ms = sum(a.elapsed_time(b) for _, a, b in bh.fetch_events)
        # ---- the same path from the fully resident hierarchy, at the granularity each budgeted frame settled at ----------
        ri = torch.zeros(G, dtype=torch.int32, device=dev); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
        w = torch.zeros(G, device=dev); ns = torch.zeros(G, dtype=torch.int32, device=dev)

        def resident(k, t):
            n = expand_to_size(nodes, boxes, t, vps[k][0], torch.zeros(3), ri, pi, ni)
            get_interpolation_weights(ni[:n], t, nodes, boxes, vps[k][1], torch.zeros(3), w, ns)
            rs = _settings(dgr, cams[k], dev, do_depth=False, interpolation_weights=w, num_node_kids=ns,
                           render_indices=ri[:n], parent_indices=pi)
            with torch.no_grad():
                return dgr.GaussianRasterizer(rs)(means3D=full["means3D"], means2D=m2[:G], shs=full["shs"],
                                                  opacities=full["opacities"], scales=full["scales"],
                                                  rotations=full["rotations"])[0]
        for i in range(min(3, steps)):
            resident(warmup + i, sels[i][1])
        # how tight the regulation is (untimed): rows a view NEEDS at the granularity it settled at = its cut's node rows
        # + the distinct parent rows of entries of weight < 1 (make_resident fetches nothing else), against the budget --
        # and, for comparison, at one escalation step finer (tau / 1.2), the step the regulator could not take
        mark = torch.zeros(G, dtype=torch.bool, device=dev)

        def rows_needed(k, t):
            n = expand_to_size(nodes, boxes, t, vps[k][0], torch.zeros(3), ri, pi, ni)
            get_interpolation_weights(ni[:n], t, nodes, boxes, vps[k][1], torch.zeros(3), w, ns)
            mark.zero_()
            mark[ri[:n].long()] = True
            lt1 = w[:n] < 1.0
            mark[pi[:n].long()[lt1]] = True
            return int(mark.sum().item()), n
        need = [rows_needed(warmup + i, sels[i][1]) for i in range(steps)]
        finer = [rows_needed(warmup + i, sels[i][1] / 1.2) for i in range(0, steps, max(1, steps // 4))]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            color_r = resident(warmup + i, sels[i][1])
        torch.cuda.synchronize()
        elapsed_res = time.perf_counter() - t0
        same_bits = bool(torch.equal(color_b, color_r))
    finally:
        ghC.set_viewpoint_cache(prev_cache)
    px = lambda t: (t * (0.5 * W) / cam.tanfovx - 1) / 2
    rows = [s[2] for s in sels]
    return {"what": "BASELINE configs[4] with VRAM-budgeted streaming LOD: the 50 M-node hierarchy's attribute rows in pinned "
                    "host memory, a budget of them on the GPU (hgs/residency.py), per frame cut + weights + residency + "
                    "3840x2160 render on the slot arrays (forward only); camera flying forward 0.08 units per frame with one "
                    "2-unit jump sideways halfway",
            "metric": "rendered frames/s @ 3840x2160", "value": steps / elapsed, "unit": "frames/s", "steps": steps,
            "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "frame_ms": {"p50": frame_ms[len(frame_ms) // 2], "p99": frame_ms[min(len(frame_ms) - 1, int(0.99 * len(frame_ms)))],
                         "max": frame_ms[-1], "min": frame_ms[0]},
            "resident_same_path": {"ms_per_step": elapsed_res / steps * 1e3, "value": steps / elapsed_res,
                                   "last_frame_bit_identical": same_bits,
                                   "what": "the same camera path from the fully resident hierarchy at the granularity "
                                           "each budgeted frame settled at"},
            "config": {"hierarchy_nodes": G, "attribute_bytes_on_host": int(G * bh.row_bytes), "host_row_bytes": 256, "budget_mb": budget_mb,
                       "budget_rows": bh.B, "requested_tau_px": tau_px,
                       "rendered_tau_px": sum(px(s[1]) for s in sels) / len(sels),
                       "rendered_tau_px_range": [px(min(s[1] for s in sels)), px(max(s[1] for s in sels))],
                       "mean_cut": sum(s[0] for s in sels) / len(sels),
                       "rows_needed_per_frame_min_median_max": [min(r for r, _ in need), sorted(r for r, _ in need)[len(need) // 2],
                                                                max(r for r, _ in need)],
                       "budget_occupancy_min_median_max": [min(r for r, _ in need) / bh.B, sorted(r for r, _ in need)[len(need) // 2] / bh.B,
                                                           max(r for r, _ in need) / bh.B],
                       "rows_needed_one_step_finer_over_budget": [r / bh.B for r, _ in finer],
                       "occupancy_note": "rows_needed = node rows of the cut + distinct parent rows of entries of weight < 1 at "
                                         "the granularity the frame settled at, over budget_rows; one_step_finer = the same at "
                                         "tau / 1.2 (sampled frames): above 1 means the regulator's next finer step cannot fit",
                       "viewpoint_cache": True,
                       "cuts_per_frame": sum(s[3] for s in sels) / len(sels),
                       "rows_fetched_per_frame": (bh.stats["rows_fetched"] - f0) / steps,
                       "rows_fetched_per_frame_min_median_max": [min(rows), sorted(rows)[len(rows) // 2], max(rows)],
                       "rows_fetched_share_of_cut_median": sorted(r / max(s[0], 1) for r, s in zip(rows, sels))[len(rows) // 2],
                       "fetch_kernel": {"launches": len(bh.fetch_events), "rows": fetch_rows, "ms": fetch_ms,
                                        "pcie_GBps": fetch_rows * bh.row_bytes / max(fetch_ms, 1e-9) / 1e6},
                       "evictions": bh.stats["evictions"] - e0,
                       "prefetch": "the next frame's cut, weights and residency run on a second stream behind this frame's "
                                   "render; rows_fetched_per_frame counts them, rows_fetched_* min/median/max what select "
                                   "still had to fetch itself",
                       "prefetched_rows": bh.stats.get("prefetched_rows", 0),
                       "cold_start": {"seconds": t_cold, "rows": cold["rows_fetched"],
                                      "pcie_GBps": cold["bytes_fetched"] / t_cold / 1e9},
                       "host_copy_s": t_host,
                       "capacity_misses": dgrC.stats["capacity_misses"] - miss0, "width": W, "height": H}}


C5_STEPS = (16, 8)      # (steps, warmup) of the configs[4] extra: the warm-up sees each of the 8 cameras once


def run_extras(args, dev, measure):
    from hgs import synth
    out = {}
    wanted = [x for x in args.extras.split(",") if x]
    W, H = 1920, 1080
    cam = synth.make_camera(W, H)
    jobs = {
        "config2_300k": lambda: extra_dropin("config2_300k", synth.make_scene(300_000, cam, seed=0), W, H, dev, measure, 60, 10,
                                             "BASELINE configs[1]: ~300 k Gaussians, 1080p, train_single.py fwd+bwd call shape"),
        "heavy_1m": lambda: extra_dropin("heavy_1m", synth.make_scene(1_000_000, cam, seed=0, s_px=(1.0, 8.0)), W, H, dev,
                                         measure, 40, 8, "the metric configuration with heavier footprints: 1 M Gaussians, "
                                         "s_px in [1, 8] (SURVEY App. C 'heavy 1 M'), 1080p, fwd+bwd"),
        # what the reference's scripts hand to the op after training at 1080p (profiles/r05_config2_config3_scripts.log)
        "trained_like_10m": lambda: extra_dropin("trained_like_10m", synth.make_scene_trained_scale(375_000, cam, seed=0), W, H,
                                                 dev, measure, 30, 8, "the statistics of a scene train_single.py trained at "
                                                 "1080p (train_single.py:97-176): 375 k Gaussians, ~28 tile instances per "
                                                 "Gaussian, L ~ 10 M, lists of ~1 300 on average and > 3 000 at most, fwd+bwd"),
        "trained_cut_10m": lambda: extra_dropin("trained_cut_10m", synth.make_scene_trained_scale(375_000, cam, seed=0, order="clustered"),
                                                W, H, dev, measure, 30, 8, "the same scene in the row order of a hierarchy cut "
                                                "(train_post.py:91-142): the cut's big nodes side by side, one K1 workgroup's "
                                                "rows emitting 840 k instances against a mean of 7 k, fwd+bwd"),
        "config3_train_post": lambda: extra_train_post(dev, measure, 40, 8),
        "config5_50m_4k_render": lambda: extra_config5(dev, *C5_STEPS),
        "config5_budgeted_6gb": lambda: extra_config5_budgeted(dev, 32, 8),
    }
    for name in wanted:
        if name not in jobs:
            continue
        t0 = time.perf_counter()
        try:
            out[name] = jobs[name]()
        except Exception as e:          # an extra must never take the headline line down with it
            out[name] = {"error": repr(e)}
        out[name]["wall_s"] = time.perf_counter() - t0
        from gaussian_hierarchy import _C as ghC
        ghC.set_viewpoint_cache(False)
        torch.cuda.empty_cache()
    return out


def roofline_object(sb, stages, dom, model_text, extra=None):
    sec = stages[dom] * 1e-3
    achieved = sb[dom] / sec / 1e9
    r = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "avg_ms": stages[dom], "algorithmic_bytes": sb[dom], "bytes_model": model_text,
         "traffic": None}
    # the whole frame next to the dominant kernel: every stage's §8(d) bytes over the sum of the stage times -- the figure
    # to read when the dominant kernel is ALU-bound and its own HBM fraction says little
    known = [s_ for s_ in stages if s_ in sb]
    if known:
        tot_b = sum(sb[s_] for s_ in known)
        tot_s = sum(stages[s_] for s_ in known) * 1e-3
        r["frame"] = {"achieved": tot_b / tot_s / 1e9, "frac": tot_b / tot_s / 1e9 / HBM_PEAK_GBS, "bytes": tot_b,
                      "ms": tot_s * 1e3, "what": "all stages of the frame: sum of their §8(d) bytes / sum of their times "
                                                 "(frame_hbm_frac uses the wall clock)"}
    if extra:
        r.update(extra)
    return r


def measure_host_floor(dgr, synth, dev, P=2000, W=256, H=256, steps=300):
    """Wall time per drop-in fwd+bwd at a size whose GPU time is negligible: what the call path costs on THIS host
    (Python + ctypes + torch allocator + autograd engine + ~17 kernel launches + the wait for the instance count)."""
    cams = [synth.orbit_camera(W, H, j, 8) for j in range(8)]
    sc = synth.make_scene(P, cams[0], seed=0)
    prm = {kk: getattr(sc, kk).to(dev).contiguous().requires_grad_(True)
           for kk in ("means3D", "shs", "opacities", "scales", "rotations")}
    gc, gd = synth.upstream_grads(H, W)
    d = DropIn(dgr, prm, 3, [_settings(dgr, c, dev) for c in cams], gc.to(dev), gd.to(dev), dev)
    for _ in range(30):
        d.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        d.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": ms, "what": f"wall time per drop-in fwd+bwd of {P} Gaussians at {W}x{H} (GPU work ~0.05 ms): the "
                                       "floor of the call path on this host; a configuration whose GPU stage sum is below it "
                                       "is host-bound"}


def measure_exchange(dp, params, dev, rank, world, rccl_log, reps=8):
    """N > 1: the gradient exchange of one step ALONE (the two spans of the bucket, back to back, nothing to overlap
    with), timed per rank with events; what RCCL says about the algorithm / protocol it uses for these sizes."""
    bucket = dp.GradBucket({kk: tuple(v.shape) for kk, v in params.items()}, dev)
    spans = [bucket.span(dp.DataParallelStep.EARLY), bucket.span(dp.DataParallelStep.LATE)]
    for _ in range(3):
        for sp in spans:
            dist.all_reduce(sp)
    torch.cuda.synchronize()
    dist.barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        for sp in spans:
            dist.all_reduce(sp)
        ev[i + 1].record()
    torch.cuda.synchronize()
    mine = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]
    per_rank = [None] * world
    dist.all_gather_object(per_rank, float(mine))
    nbytes = int(sum(sp.numel() for sp in spans) * 4)
    out = {"backend": dist.get_backend(), "bytes_per_step": nbytes, "spans_bytes": [int(sp.numel() * 4) for sp in spans],
           "exchange_ms_per_rank": per_rank,
           # all-reduce bus bandwidth in nccl-tests' convention: 2 (N - 1) / N x bytes / time
           "busbw_GBps": 2.0 * (world - 1) / world * nbytes / (max(per_rank) * 1e-3) / 1e9,
           "env": {kk: os.environ[kk] for kk in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                 "HSA_ENABLE_IPC_MODE_LEGACY", "HGS_DP_ALLREDUCE") if kk in os.environ}}
    try:
        out["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        pass
    if rank == 0 and rccl_log:
        try:
            path = rccl_log.replace("%p", str(os.getpid())).replace("%h", os.uname().nodename)
            lines = [ln.strip() for ln in open(path, errors="replace")]
            keep, seen = [], set()
            for ln in lines:
                low = ln.lower()
                if ("algo" in low and "proto" in low) or "channels" in low or "nccl version" in low or "rccl version" in low:
                    key = ln.split("]")[-1].strip()[:160]
                    if key not in seen:
                        seen.add(key)
                        keep.append(key)
            out["rccl_trace"] = keep[:24]
            # RCCL's own count of the communicator ("... nranks 8 ..." on its INIT lines) and the GPUs it names there
            import re
            nr = sorted({int(m.group(1)) for ln in lines for m in [re.search(r"nranks (\d+)", ln)] if m})
            bus = sorted({m.group(1) for ln in lines for m in [re.search(r"busId ([0-9a-fA-F]+)", ln)] if m})
            out["ranks_seen"] = {"nranks": nr, "bus_ids": bus[:16]}
            out["rccl_trace_note"] = ("unique lines of RCCL's INFO log (INIT, TUNING) that name an algorithm / protocol / "
                                      "channel count; algo 0 Tree 1 Ring, proto 0 LL 1 LL128 2 Simple in RCCL 2.2x")
        except OSError as e:
            out["rccl_trace"] = [f"no log: {e!r}"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--views-per-step", type=int, default=8,
                    help="BATCHED schedule: views rendered (fwd+bwd) per rank between two gradient all-reduces")
    ap.add_argument("--schedule", choices=("auto", "dropin", "batched"), default="auto",
                    help="which schedule `value` reports: auto = drop-in at N = 1, batched at N > 1")
    ap.add_argument("--no-batched-sh-forward", action="store_true",
                    help="batched schedule: evaluate the SH colours inside every view's rasterizer call instead of "
                         "once per step for all k views (then only the SH backward is batched, see --no-deferred-sh)")
    ap.add_argument("--no-stream-overlap", action="store_true",
                    help="batched schedule: enqueue the backwards on the forwards' stream (default: a second HIP "
                         "stream, so that the HBM-bound stages of one view overlap the ALU-bound compositing of the next)")
    ap.add_argument("--no-deferred-sh", action="store_true",
                    help="batched schedule: per-view SH backward (accumulating) instead of one batched pass per step")
    ap.add_argument("--no-secondary", action="store_true", help="measure only the schedule `value` reports")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `extra` objects (BASELINE configs 2 / 3 / 5 and the heavy 1 M variant)")
    ap.add_argument("--extras", default="config2_300k,heavy_1m,trained_like_10m,trained_cut_10m,config3_train_post,config5_50m_4k_render,config5_budgeted_6gb")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not collect the PMC counters of the roofline object in this run (rocprofv3 child processes, "
                         "~15 s); the committed summary of the same build is used instead.  Implied by --no-extras")
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

    # `--gpus N` means N ranks.  Launched plainly (no torchrun environment) with N > 1, the bench launches itself as the
    # driver would: one process per GPU under torch.distributed.run on 127.0.0.1.  It refuses a node with fewer GPUs than
    # ranks -- unless HGS_DP_BACKEND=gloo says the ranks are meant to share a GPU (the functional test of the N > 1 path).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if torch.cuda.device_count() < args.gpus and os.environ.get("HGS_DP_BACKEND") != "gloo":
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {torch.cuda.device_count()} GPU(s); "
                             "one rank per GPU is the measured configuration (HGS_DP_BACKEND=gloo lets ranks share a GPU "
                             "for a functional run)")
        import socket
        import subprocess
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={os.environ['WORLD_SIZE']}: the two must agree")

    from hgs import _lib, dp, synth
    import diff_gaussian_rasterization as dgr

    rccl_log = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # the first exchanges of the direct (peer-pointer) route check themselves against torch.distributed's all-reduce
        os.environ.setdefault("HGS_P2P_VERIFY", "2")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("HGS_BENCH_RCCL_TRACE", "1") != "0":
        # the first SCALE record should explain itself: RCCL's own account of the algorithm / protocol it picked for the
        # bucket goes to a per-process file (never to stdout: the line below stays the only one) and is quoted in the line
        rccl_log = f"/tmp/hgs_rccl_{os.getpid()}.log"
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING")
        os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
        rccl_log = os.environ["NCCL_DEBUG_FILE"]
    rank, local, world = dp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rasterizer has no CPU path (see oracle/ for the checker)")
    if _lib.lib().hgs_device_count() < 1:
        raise SystemExit("libhgs.so sees no HIP device")
    dev = torch.device("cuda", local if torch.cuda.device_count() > local else 0)
    torch.cuda.set_device(dev)
    W, H, P = args.width, args.height, args.gaussians
    primary = args.schedule if args.schedule != "auto" else ("dropin" if world == 1 else "batched")

    base_cam = synth.make_camera(W, H)
    scene_cpu = synth.make_scene(P, base_cam, seed=0)            # same Gaussians on every rank
    scene = scene_cpu.to(dev)
    gc_cpu, gd_cpu = synth.upstream_grads(H, W, seed=1)
    gc, gd, bg = gc_cpu.to(dev), gd_cpu.to(dev), torch.zeros(3, device=dev)
    e_i = torch.empty(0, dtype=torch.int32, device=dev)
    e_f = torch.empty(0, dtype=torch.float32, device=dev)
    params = dict(means3D=scene.means3D, shs=scene.shs, opacities=scene.opacities, scales=scene.scales,
                  rotations=scene.rotations)
    for t in params.values():
        t.requires_grad_(True)
    info = {"L": 0}

    def settings(cam_c):
        cam = cam_c.to(dev)
        return dgr.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0,
            viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=scene.sh_degree,
            campos=cam.camera_center, prefiltered=False, debug=False, do_depth=True, render_indices=e_i,
            parent_indices=e_i, interpolation_weights=e_f, num_node_kids=e_i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- DROP-IN schedule: the reference's call, one view per step, one stream ------------------------------------
    # (N > 1: every rank has its own cameras and the step ends with the all-reduce of the .grad tensors' bucket copy)
    NCAM = 8
    cams_dropin = [synth.orbit_camera(W, H, rank * NCAM + j, world * NCAM, radius=0.05, tilt=0.004) for j in range(NCAM)]
    dp_bucket = dp.GradBucket({kk: tuple(v.shape) for kk, v in params.items()}, dev) if world > 1 else None
    dropin = DropIn(dgr, params, scene.sh_degree, [settings(c) for c in cams_dropin], gc, gd, dev, dp_bucket)
    step_dropin = dropin.step

    # ---- BATCHED schedule: k views per step through a RasterContext ------------------------------------------------
    k = max(1, args.views_per_step)
    n_views = world * k
    cams_cpu = [base_cam if n_views == 1 else synth.orbit_camera(W, H, rank * k + j, n_views, radius=0.05, tilt=0.004)
                for j in range(k)]
    overlap = not args.no_stream_overlap and k > 1      # one view per step: nothing to run next to
    sh_fwd = bool(k > 1 and not args.no_batched_sh_forward)
    defer_sh = bool(k > 1 and not args.no_deferred_sh and args.no_batched_sh_forward)
    state = {}

    def setup_batched():
        bucket = dp.GradBucket({kk: tuple(v.shape) for kk, v in params.items()}, dev)
        m2_grad = torch.empty(P, 3, device=dev)                 # per-view means2D gradient (densification statistic)
        rc = dgr.RasterContext(grad_buffers=dict(bucket.views, means2D=m2_grad),
                               backward_stream=torch.cuda.Stream(device=dev) if overlap else None,
                               defer_sh_backward=defer_sh)
        rasts = [dgr.GaussianRasterizer(settings(c), context=rc) for c in cams_cpu]
        state.update(bucket=bucket, rc=rc, rasts=rasts, campos=[r.raster_settings.campos for r in rasts],
                     means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
                     d_rgbs=[torch.empty(P, 3, device=dev) for _ in rasts] if sh_fwd else None)

    def step_batched():
        rc, rasts, bucket, means2D = state["rc"], state["rasts"], state["bucket"], state["means2D"]
        if sh_fwd:
            # colours of all k views in ONE pass over the coefficients (the reference's convert_SHs_python route, in
            # HIP), k rasterizations with colors_precomp, then ONE pass for dL/dSH and the view-direction part of
            # dL/dmeans3D
            with torch.no_grad():
                rgbs, clamps = dgr.sh_colors_batched(params["means3D"], params["shs"], scene.sh_degree, state["campos"])
            d_rgbs = state["d_rgbs"]
            for j, rast in enumerate(rasts):
                rc.grad_accumulate = j > 0
                rc.grad_buffers["colors_precomp"] = d_rgbs[j]           # per view, overwritten
                color, radii, invd = rast(means3D=params["means3D"], means2D=means2D,
                                          colors_precomp=rgbs[j].requires_grad_(True), opacities=params["opacities"],
                                          scales=params["scales"], rotations=params["rotations"])
                info["L"], info["radii"] = color.grad_fn.num_rendered, radii
                torch.autograd.backward([color, invd], [gc, gd])        # every gradient lands in a buffer
            # N > 1: the (opacity, scale, rotation) slice of the bucket is final once the last view's per-Gaussian backward
            # has run -- it goes on the wire (ordered after that backward) while the batched SH backward still runs
            early = _reduce_async(rc, bucket, dp.DataParallelStep.EARLY)
            rc.sh_colors_batched_backward(params["means3D"], params["shs"], scene.sh_degree, state["campos"], clamps,
                                          d_rgbs, bucket.views["shs"], bucket.views["means3D"])
            late = _reduce_async(rc, bucket, dp.DataParallelStep.LATE)
        else:
            for j, rast in enumerate(rasts):
                rc.grad_accumulate = j > 0
                color, radii, invd = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"],
                                          opacities=params["opacities"], scales=params["scales"],
                                          rotations=params["rotations"])
                info["L"], info["radii"] = color.grad_fn.num_rendered, radii
                torch.autograd.backward([color, invd], [gc, gd])
            early = None
            if rc.defer_sh_backward:
                early = _reduce_async(rc, bucket, dp.DataParallelStep.EARLY)
                rc.finish_deferred_sh_backward()
                late = _reduce_async(rc, bucket, dp.DataParallelStep.LATE)
            else:
                late = _reduce_async(rc, bucket, bucket.names)
        for work in (early, late):
            if work is not None:
                work.wait()
        rc.wait_backward_stream()

    def _reduce_async(rc, bucket, names):
        """SUM all-reduce of one contiguous group of the bucket, issued on the stream the backwards run on."""
        if world == 1:
            return None
        sb = rc.backward_stream
        if sb is None:
            return bucket.all_reduce_async(names)
        with torch.cuda.stream(sb):
            work = bucket.all_reduce_async(names)

        class _OnStream:                      # wait() must order the BACKWARD stream (wait_backward_stream does the rest)
            def wait(self_inner):
                with torch.cuda.stream(sb):
                    work.wait()
        return _OnStream()

    def measure(step, steps, warmup, dominant_timing, reset=None):
        for _ in range(warmup):
            step()
        barrier()
        if reset is not None:
            reset()                      # per-step statistics (L, visible counts) of the TIMED steps only
        if dominant_timing:
            # inside the timed region only the dominant kernel is bracketed by hipEvents (the roofline figure must
            # come from the timed steps themselves); the other stages are timed in a short extra pass afterwards
            _lib.timing_read(reset=True)
            _lib.timing_enable(True, stages=[DOMINANT])
        barrier()
        host = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            ta = time.perf_counter()
            step()
            host += time.perf_counter() - ta
        barrier()
        elapsed = time.perf_counter() - t0
        info["host_ms_per_step"] = host / steps * 1e3       # time spent INSIDE step(): Python, ctypes, launches, host waits
        measure.last_host_ms = info["host_ms_per_step"]
        stages = {}
        if dominant_timing:
            _lib.timing_enable(False)
            dom_ms = {kk: (ms / max(c, 1)) for kk, (ms, c) in _lib.timing_read(reset=True).items() if c}
            _lib.timing_enable(True)
            for _ in range(max(2, min(steps, 5))):
                step()
            barrier()
            _lib.timing_enable(False)
            stages = {kk: (ms / max(c, 1)) for kk, (ms, c) in _lib.timing_read(reset=True).items() if c}
            stages.update(dom_ms)
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, stages

    timing = not args.no_stage_timing
    res = {}
    # the secondary schedule is measured FIRST: the driver's command times 20 steps (20 ms in the drop-in shape), and a
    # chip that sat idle while the scene was generated on the host is still ramping its clocks then; the primary
    # measurement itself stays "W untimed steps, then exactly K timed steps"
    order = ([] if args.no_secondary else [s_ for s_ in ("dropin", "batched") if s_ != primary]) + [primary]
    for sched in order:
        is_primary = sched == primary
        if sched == "dropin":
            steps = args.steps if is_primary else max(20, min(args.steps, 40))
            from diff_gaussian_rasterization import _C as dgrC
            miss0 = dgrC.stats["capacity_misses"]
            elapsed, stages = measure(step_dropin, steps, args.warmup if is_primary else 20, timing and is_primary,
                                      reset=dropin.reset_stats)
            info["dropin_L"], info["dropin_V"] = dropin.mean_L(), dropin.mean_V()
            info["dropin_misses"] = dgrC.stats["capacity_misses"] - miss0      # warm-up + timed + stage-timing steps
            views = 1
            for t in params.values():
                t.grad = None
        else:
            setup_batched()
            steps = args.steps if is_primary else max(6, min(args.steps, (args.steps * 2 + k - 1) // k))
            elapsed, stages = measure(step_batched, steps, args.warmup if is_primary else 6, timing and is_primary)
            views = k
            state.clear()
        res[sched] = dict(value=world * views * steps / elapsed, ms_per_step=elapsed / steps * 1e3, steps=steps,
                          views_per_step_per_gpu=views, stages=stages, host_ms_per_step=info.get("host_ms_per_step"))

    exchange = None
    if world > 1:
        exchange = measure_exchange(dp, params, dev, rank, world, rccl_log)
    host_floor = None
    if world == 1 and primary == "dropin" and not args.no_extras:      # (--no-extras: the profiling runs -- no tiny frames
        host_floor = measure_host_floor(dgr, synth, dev)                # of the same kernels in their per-kernel averages)

    if rank == 0:
        r = res[primary]
        if primary == "dropin":
            L, V = int(round(info["dropin_L"])), int(round(info["dropin_V"]))     # means over the timed steps' views
        else:
            L = int(info["L"])
            V = int((info["radii"] > 0).sum().item())
        N, T, M = W * H, ((W + 15) // 16) * ((H + 15) // 16), scene.shs.shape[1]
        kk_ = r["views_per_step_per_gpu"]
        batched_primary = primary == "batched"
        ab = algorithmic_bytes(P, V, L, N, T, M, k=kk_, deferred_sh=batched_primary and (defer_sh or sh_fwd),
                               sh_forward=batched_primary and sh_fwd)
        sb = survey_bytes(P, V, L, N, T, M)
        per_step = ("sh_bwd_batched", "sh_colors_batched")
        impl_frame = sum(v for n_, v in ab.items() if n_ not in per_step) + sum(ab.get(n_, 0) for n_ in per_step) / kk_
        survey_frame = sum(sb.values())
        fps_per_gpu = r["value"] / world
        batched_desc = (f"{k} views per rank per step accumulated in place through a RasterContext" +
                        (", SH backward batched over the views" if defer_sh else "") +
                        (", SH colours and their backward batched over the views" if sh_fwd else "") +
                        (", backwards on a second HIP stream" if overlap else "") +
                        (", RCCL all-reduce of the 59P-float grad bucket per step in two parts, the first overlapping the SH backward" if world > 1 else ""))
        dropin_desc = ("one view per step, GaussianRasterizer(raster_settings)(means3D, means2D, shs, opacities, scales, "
                       "rotations) + backward exactly as train_single.py:97,123 / gaussian_renderer/__init__.py:105-113, "
                       "one stream, no opt-in API" + (", all-reduce of the gradients every step" if world > 1 else ""))
        result = {
            "metric": "fwd+bwd frames/s @1080p, 1M Gaussians", "value": r["value"], "unit": "frames/s",
            "n_gpus": world, "steps": r["steps"], "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{P} frustum-filling synthetic Gaussians (SURVEY §8(d) spec, seed 0), "
                                   f"{W}x{H}, SH degree 3, depth channel on, fwd+bwd through GaussianRasterizer",
                       "gaussians": P, "visible": V, "tile_instances": L, "width": W, "height": H,
                       "schedule": primary, "views_per_step_per_gpu": kk_,
                       "cameras": f"{NCAM if primary == 'dropin' else k} orbit cameras per rank, cycled "
                                  "(visible / tile_instances = mean over the timed steps)",
                       "capacity_misses": info.get("dropin_misses") if primary == "dropin" else None,
                       "parallelism": f"per-view dp{world}: " + (batched_desc if batched_primary else dropin_desc),
                       "exchange": (None if world == 1 else
                                    "direct two-shot all-reduce over peer pointers (hgs_p2p_*, HGS_DP_ALLREDUCE=direct)"
                                    if os.environ.get("HGS_DP_ALLREDUCE", "") == "direct" else
                                    f"torch.distributed all-reduce ({dist.get_backend()})"),
                       "scaling_note": "value is the drop-in call shape at N = 1 and the batched data-parallel schedule "
                                       "at N > 1; compare batched.value across N for scaling efficiency",
                       "measurement_order": "the secondary schedule runs before the primary one (chip at its clocks "
                                            "when the W warm-up + K timed steps of `value` start)"},
            "algorithmic_bytes_per_frame": survey_frame,
            "impl_bytes_per_frame": impl_frame,
            "ms_per_frame_per_gpu": r["ms_per_step"] / kk_,
            "frame_hbm_frac": survey_frame * fps_per_gpu / 1e9 / HBM_PEAK_GBS,
            # host side of the call path: time inside step() per timed step (it contains the forward's wait for the
            # instance count, i.e. GPU time of K1 + scan when the host runs ahead), and the floor of the path itself
            "host_ms_per_step": r.get("host_ms_per_step"),
        }
        if host_floor is not None:
            result["host_floor"] = host_floor
        if exchange is not None:
            result["exchange"] = exchange
        if world > 1 and "dropin" in res and "batched" in res:
            # the >= 6x question of north_star answered for BOTH schedules from one run: `batched` (k views per rank per
            # step, one exchange per step) and the partition north_star describes -- one view per rank per optimizer step,
            # global batch N, the 59 P-float gradient bucket all-reduced EVERY step, nothing to hide it under
            pv, ex_ms = res["dropin"], (max(exchange["exchange_ms_per_rank"]) if exchange else None)
            result["per_view_dp"] = {
                "what": "north_star's partition: one view per rank per optimizer step (global batch N = n_gpus), all-reduce "
                        "of the whole gradient bucket every step, one stream; compare value across N",
                "value": pv["value"], "unit": "frames/s", "views_per_step_per_gpu": 1, "global_batch": world,
                "ms_per_step": pv["ms_per_step"], "steps": pv["steps"],
                "exchange_alone_ms": ex_ms,
                "exchange_share_of_step": (ex_ms / pv["ms_per_step"]) if ex_ms else None,
                "measured_on_hardware": True}
        for name, rr in res.items():
            result[name] = {"value": rr["value"], "unit": "frames/s", "views_per_step_per_gpu": rr["views_per_step_per_gpu"],
                            "steps": rr["steps"], "ms_per_step": rr["ms_per_step"],
                            "schedule": batched_desc if name == "batched" else dropin_desc}
        stages = r["stages"]
        src_sha = kernel_source_sha()
        result["kernel_source_sha"] = src_sha
        if stages:
            dom = DOMINANT if DOMINANT in stages else max(stages, key=stages.get)
            sec = stages[dom] * 1e-3
            traffic_db, traffic_path = _profile_json("pmc_traffic.json")
            valu_db, valu_path = _profile_json("pmc_valu.json")
            mix_peak = {k_: v_ for k_, v_ in (valu_db or {}).items() if k_ in ("_peak_ginst_s", "_peak_source")}
            # committed PMC summaries are used only if they were collected on THIS build of the kernels
            if traffic_db is not None and traffic_db.get("_src_sha") != src_sha:
                stale_t, traffic_db = traffic_db.get("_run", "unknown"), None
            else:
                stale_t = None
            if valu_db is not None and valu_db.get("_src_sha") != src_sha:
                valu_db = None
            live_note = None
            if world == 1 and primary == "dropin" and not (args.no_extras or args.no_live_pmc):
                live_t, live_v, live_note = measure_pmc_live(args, N, L)
                if live_t is not None:
                    # (the measured issue rate of K7's instruction mix is a property of the chip, not of the build)
                    traffic_db = dict(live_t, _run="this run", _live=True)
                    valu_db = dict(live_v, _run="this run", _live=True, **mix_peak)
                    traffic_path = valu_path = "rocprofv3 child processes of this run"
            traffic = (traffic_db or {}).get(dom)
            result["roofline"] = roofline_object(
                sb, stages, dom,
                "SURVEY.md §8(d): render_bwd = 24 N + 44 L + 40 V (N pixels, L tile instances, V visible Gaussians: means "
                "over the views of THIS run); avg_ms = hipEvents around every launch inside the timed steps",
                {"traffic": traffic, "traffic_upper": (traffic_db or {}).get(dom + "_upper"),
                 "traffic_source": (((live_note + ".  ") if traffic_db.get("_live") else
                                     (f"{traffic_path}['{dom}'] (run id {traffic_db.get('_run', 'unknown')}, kernel sources "
                                      f"{src_sha} = this build): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      "command -- a committed profile, NOT measured in this run" +
                                      (f" ({live_note})" if live_note else "") + ".  ")) + "Streaming kernels: "
                                    "(2*FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 correction of the guide); the "
                                    "compositing kernels read 64-byte records at random places, for which FETCH_SIZE "
                                    "is exact (profiles/r03_microbench_gather_fetch.txt): FETCH_SIZE KiB + half of "
                                    "the streamed reads of the byte model + WRITE_SIZE KiB; traffic_upper = the "
                                    "uncorrected 2*FETCH_SIZE figure") if traffic else
                                   ("; ".join(x for x in (
                                       live_note, f"dropped: profiles/pmc_traffic.json (run {stale_t}) was collected on "
                                                  "another build of the kernels" if stale_t else None) if x) or None),
                 "impl_bytes": ab[dom], "impl_achieved": ab[dom] / sec / 1e9,
                 "impl_frac": ab[dom] / sec / 1e9 / HBM_PEAK_GBS,
                 "impl_bytes_model": "this implementation's own traffic model (record and per-instance scratch as laid out in HBM)",
                 "note": "compositing kernels are VALU-issue-bound (gather/blend, no MFMA); the HBM fraction is "
                         "reported because the metric mandates it"})
            if traffic_db:      # every stage's HBM bytes per launch beside its §8(d) bytes: where re-reads are
                result["roofline"]["traffic_by_stage"] = {
                    s_: {"traffic": traffic_db[s_], "algorithmic_bytes": sb.get(s_),
                         "ratio": (traffic_db[s_] / sb[s_]) if sb.get(s_) else None}
                    for s_ in stages if s_ in traffic_db}
            # The compositing kernels are VALU-issue-bound: add the vector-ALU view next to the mandated HBM one.
            pv = (valu_db or {}).get(dom)
            if pv:
                mix = (valu_db.get("_peak_ginst_s") or 614.4) * 1e9
                rate = pv["valu_insts_per_launch"] / sec
                result["roofline"]["valu"] = {
                    "insts_per_launch": pv["valu_insts_per_launch"], "achieved_ginst_s": rate / 1e9,
                    # two roofs, so that neither can be misread: the guide's issue rate for plain wave64 VALU
                    # instructions (one per 2 cycles per SIMD: 256 CUs x 4 SIMDs x 2.4 GHz / 2) and the measured rate
                    # of THIS kernel's instruction mix (packed / compare / transcendental instructions issue slower)
                    "peak_guide_ginst_s": VALU_GUIDE_GINST_S, "frac_of_guide_peak": rate / 1e9 / VALU_GUIDE_GINST_S,
                    "peak_mix_ginst_s": mix / 1e9, "frac_of_mix_peak": rate / mix,
                    # SQ_ACTIVE_INST_VALU (quad-cycles the vector ALU was executing, per wave) x waves over the
                    # SIMD-cycles the launch lasted at the nominal 2.4 GHz: how busy the vector ALUs were
                    "alu_busy_frac": (pv["valu_active_quadcycles_per_wave"] * 4.0 * pv["waves"] / 1024.0) / (sec * 2.4e9)
                    if "valu_active_quadcycles_per_wave" in pv else None,
                    "source": ("SQ_INSTS_VALU per launch measured in this run (rocprofv3 --pmc child process of this "
                               "command); time from this run; peak_mix_ginst_s = " if valu_db.get("_live") else
                               f"{valu_path} (run id {valu_db.get('_run', 'unknown')}, kernel sources {src_sha} = this "
                               "build): SQ_INSTS_VALU per launch from a committed rocprofv3 --pmc pass of this command "
                               "(NOT measured in this run); time from this run; peak_mix_ginst_s = ") +
                              str(valu_db.get("_peak_source", "assumed"))}
            result["stages_ms"] = stages
            result["stages_gbs"] = {s_: sb[s_] / (v * 1e-3) / 1e9 for s_, v in stages.items() if s_ in sb}
        if world == 1 and not args.no_extras:
            result["extra"] = run_extras(args, dev, measure)
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
                                          "host_cores": os.cpu_count(), "kind": "port",
                                          "sample": f"not measured: {e!r}"}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
