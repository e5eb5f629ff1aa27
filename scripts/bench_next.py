#!/usr/bin/env python
"""Measurements for the SURVEY.md §8(f) "next" rows (not the headline metric -- that is bench.py):

  f-1  in-op LOD gather-lerp        vs the Python glue of gaussian_renderer/__init__.py:199-235 (torch ops)
  f-3  raw-parameter path            vs torch activations + cat (scene/gaussian_model.py:108-128) around the op
  f-4  fused row-sparse Adam         vs the torch-op chain of scene/OurAdam.py:249-337 (restated with torch ops)
  train  a complete optimiser step: 4 views (batched SH ends) with an L1 + inverse-depth loss, fused Adam

Every leg is a full training-style iteration (forward + backward, or one optimiser step) at 1080p on synthetic
data, timed with the stream drained on both sides.  One JSON object per leg on stdout.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

from hgs import synth, hierarchy


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


def settings(dgr, cam, dev, **over):
    # camera tensors resident on the GPU (as the reference's Camera objects): no host-to-device copies per step
    cache = cam.__dict__.setdefault("_bench_dev", {})
    if dev not in cache:
        cache[dev] = dict(bg=torch.zeros(3, device=dev), viewmatrix=cam.world_view_transform.to(dev),
                          projmatrix=cam.full_proj_transform.to(dev), campos=cam.camera_center.to(dev),
                          e_i=torch.empty(0, dtype=torch.int32, device=dev),
                          e_f=torch.empty(0, dtype=torch.float32, device=dev))
    c = cache[dev]
    kw = dict(image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
              bg=c["bg"], scale_modifier=1.0, viewmatrix=c["viewmatrix"], projmatrix=c["projmatrix"], sh_degree=3,
              campos=c["campos"], prefiltered=False, debug=False, do_depth=True, render_indices=c["e_i"],
              parent_indices=c["e_i"], interpolation_weights=c["e_f"], num_node_kids=c["e_i"])
    kw.update(over)
    return dgr.GaussianRasterizationSettings(**kw)


def leg_raw(args, dev):
    import diff_gaussian_rasterization as dgr
    W, H = 1920, 1080
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(args.gaussians, cam, seed=0)
    op = scene.opacities.clamp(1e-4, 1 - 1e-4)
    raw = dict(xyz=scene.means3D, f_dc=scene.shs[:, :1].contiguous(), f_rest=scene.shs[:, 1:].contiguous(),
               opacity=torch.log(op / (1 - op)), scaling=torch.log(scene.scales), rotation=scene.rotations * 1.7)
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}
    gc, gd = (t.to(dev) for t in synth.upstream_grads(H, W))
    rs = settings(dgr, cam, dev)
    rast = dgr.GaussianRasterizer(rs)
    m2 = torch.zeros(scene.P, 3, device=dev, requires_grad=True)

    def zero():
        for v in leaves.values():
            v.grad = None
        m2.grad = None

    def std():
        zero()
        color, _, invd = rast(means3D=leaves["xyz"], means2D=m2,
                              shs=torch.cat((leaves["f_dc"], leaves["f_rest"]), dim=1),
                              opacities=torch.sigmoid(leaves["opacity"]), scales=torch.exp(leaves["scaling"]),
                              rotations=torch.nn.functional.normalize(leaves["rotation"]))
        ((color * gc).sum() + (invd * gd).sum()).backward()

    def fused():
        zero()
        color, _, invd = rast.forward_raw(leaves["xyz"], m2, leaves["f_dc"], leaves["f_rest"], leaves["opacity"],
                                          leaves["scaling"], leaves["rotation"])
        ((color * gc).sum() + (invd * gd).sum()).backward()

    t_std = timed(std, args.warmup, args.steps)
    t_raw = timed(fused, args.warmup, args.steps)
    return dict(leg="f-3 raw-parameter path", gaussians=scene.P, image=[W, H],
                ms_torch_activations_plus_op=t_std, ms_fused=t_raw, speedup=t_std / t_raw)


def leg_lod(args, dev):
    import diff_gaussian_rasterization as dgr
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
    W, H = 1920, 1080
    cam = synth.make_camera(W, H)
    h = hierarchy.build_hierarchy(synth.make_scene(args.leaves, cam, seed=0))
    nodes, boxes = h.nodes.to(dev), h.boxes.to(dev)
    G = h.xyz.shape[0]
    ri = torch.zeros(G, dtype=torch.int32, device=dev); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
    w = torch.zeros(G, device=dev); ns = torch.zeros(G, dtype=torch.int32, device=dev)
    attrs = dict(xyz=h.xyz, shs=h.shs, op=h.alpha.abs().reshape(-1, 1), sc=torch.exp(h.log_scales),
                 rot=torch.nn.functional.normalize(h.rots))
    leaves = {k: v.to(dev).contiguous().requires_grad_(True) for k, v in attrs.items()}
    tau = (2 * (3.0 + 0.5)) * cam.tanfovx / (0.5 * W)
    t0 = time.perf_counter()
    n = expand_to_size(nodes, boxes, tau, cam.camera_center.to(dev), torch.zeros(3), ri, pi, ni)
    get_interpolation_weights(ni[:n], tau, nodes, boxes, cam.camera_center.cpu(), torch.zeros(3), w, ns)
    torch.cuda.synchronize()
    t_cut_first = (time.perf_counter() - t0) * 1e3

    vp_gpu, vp_cpu, zero3 = cam.camera_center.to(dev), cam.camera_center.cpu(), torch.zeros(3)

    def cut():
        k = expand_to_size(nodes, boxes, tau, vp_gpu, zero3, ri, pi, ni)
        get_interpolation_weights(ni[:k], tau, nodes, boxes, vp_cpu, zero3, w, ns)
    t_cut = timed(cut, 2, 10)
    gc = synth.upstream_grads(H, W)[0].to(dev)
    r, p = ri[:n].long(), pi[:n].long()
    m2 = torch.zeros(n, 3, device=dev, requires_grad=True)

    def zero():
        for v in leaves.values():
            v.grad = None

    def glue():   # the reference's Python interpolation, gaussian_renderer/__init__.py:204-218
        zero()
        t = w[:n, None]
        lerp = lambda a: t.view(-1, *([1] * (a.dim() - 1))) * a[r] + (1 - t).view(-1, *([1] * (a.dim() - 1))) * a[p]
        pr, rr = leaves["rot"][p], leaves["rot"][r]
        pr = torch.where(((rr * pr).sum(1, keepdim=True) < 0), -pr, pr)
        rs = settings(dgr, cam, dev, do_depth=False, interpolation_weights=w, num_node_kids=ns)
        color, _, _ = dgr.GaussianRasterizer(rs)(
            means3D=lerp(leaves["xyz"]), means2D=m2, shs=lerp(leaves["shs"]), opacities=lerp(leaves["op"]),
            scales=lerp(leaves["sc"]), rotations=t * rr + (1 - t) * pr)
        (color * gc).sum().backward()

    mfull = torch.zeros(G, 3, device=dev, requires_grad=True)

    def inop():
        zero()
        rs = settings(dgr, cam, dev, do_depth=False, interpolation_weights=w, num_node_kids=ns,
                      render_indices=ri[:n], parent_indices=pi[:n])
        color, _, _ = dgr.GaussianRasterizer(rs)(
            means3D=leaves["xyz"], means2D=mfull, shs=leaves["shs"], opacities=leaves["op"], scales=leaves["sc"],
            rotations=leaves["rot"])
        (color * gc).sum().backward()

    t_glue = timed(glue, args.warmup, args.steps)
    t_inop = timed(inop, args.warmup, args.steps)
    return dict(leg="f-1 in-op LOD interpolation", hierarchy_nodes=G, cut=n, tau_px=3.0, ms_cut_and_weights=t_cut,
                ms_cut_and_weights_first_call=t_cut_first, ms_python_glue_fwd_bwd=t_glue, ms_in_op_fwd_bwd=t_inop,
                speedup=t_glue / t_inop)


def leg_post(args, dev):
    """A train_post.py-shaped step (train_post.py:66-191) on a merged 2-chunk hierarchy: threshold log-uniform in
    [0.005, 0.1] (:66-74), expand_to_size + get_interpolation_weights (:91-113), render_post with the interpolation
    done IN the op (render_indices / parent_indices non-empty), L1 loss, backward (gather-lerp's scatter lands the
    gradients on node and parent rows), dense Adam over all hierarchy Gaussians (torch.optim.Adam, :37,191)."""
    import diff_gaussian_rasterization as dgr
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
    from hgs.optim import Adam
    W, H = 1920, 1080
    cam = synth.make_camera(W, H)
    full = synth.make_scene(args.leaves, cam, seed=0)
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
    params = {k: torch.nn.Parameter(v.to(dev).contiguous()) for k, v in attrs.items()}
    lrs = dict(xyz=1.6e-5, shs=2.5e-3, op=1e-3, sc=1e-6, rot=1e-5)
    opt = Adam([dict(params=[params[k]], lr=lrs[k], name=k) for k in params], lr=0.0, eps=1e-15)
    target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    mfull = torch.zeros(G, 3, device=dev, requires_grad=True)
    g = torch.Generator().manual_seed(3)
    vp_gpu, vp_cpu, zero3 = cam.camera_center.to(dev), cam.camera_center.cpu(), torch.zeros(3)
    cuts = []

    def step():
        limit = math.pow(2, torch.rand(1, generator=g).item() * (math.log2(0.1) - math.log2(0.005)) + math.log2(0.005))
        n = expand_to_size(nodes, boxes, limit, vp_gpu, zero3, ri, pi, ni)
        get_interpolation_weights(ni[:n], limit, nodes, boxes, vp_cpu, zero3, w, ns)
        cuts.append(n)
        rs = settings(dgr, cam, dev, do_depth=False, interpolation_weights=w, num_node_kids=ns,
                      render_indices=ri[:n], parent_indices=pi)
        color, _, _ = dgr.GaussianRasterizer(rs)(means3D=params["xyz"], means2D=mfull, shs=params["shs"],
                                                 opacities=params["op"], scales=params["sc"], rotations=params["rot"])
        loss = (color - target).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step(None)                       # dense, as torch.optim.Adam in train_post.py:191

    t = timed(step, args.warmup, max(args.steps, 20))
    return dict(leg="train_post-shaped step (log-uniform tau, cut + weights, in-op LOD render, L1, backward, dense fused Adam)",
                hierarchy_nodes=G, chunks=2, image=[W, H], mean_cut=sum(cuts) / len(cuts), min_cut=min(cuts),
                max_cut=max(cuts), ms_per_step=t, steps_per_s=1e3 / t)


def leg_adam(args, dev):
    from hgs.optim import Adam
    P = args.gaussians
    g = torch.Generator().manual_seed(0)
    shapes = dict(xyz=(3,), f_dc=(1, 3), f_rest=(15, 3), opacity=(1,), scaling=(3,), rotation=(4,))
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=5e-2, scaling=5e-3, rotation=1e-3)
    init = {k: torch.randn(P, *s, generator=g) for k, s in shapes.items()}
    grads = {k: (torch.randn(P, *s, generator=g) * 1e-3).to(dev) for k, s in shapes.items()}
    grads["opacity"][torch.rand(P, generator=g).to(dev) < 0.3] = 0
    params = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in init.items()}
    for k in shapes:
        params[k].grad = grads[k]
    opt = Adam([dict(params=[params[k]], lr=lrs[k], name=k) for k in shapes], lr=0.0, eps=1e-15)
    relevant = (grads["opacity"].flatten() != 0).nonzero().flatten().long()
    n = int(relevant.numel())

    # the reference's chain with torch ops (scene/OurAdam.py:249-337), state kept here
    tp = {k: v.clone().to(dev) for k, v in init.items()}
    tm = {k: torch.zeros_like(v) for k, v in tp.items()}
    tv = {k: torch.zeros_like(v) for k, v in tp.items()}
    step = [0]

    def torch_chain():
        step[0] += 1
        rel = (grads["opacity"].flatten() != 0).nonzero().flatten().long()
        for k in shapes:
            gr, m, v, p = grads[k][rel], tm[k][rel], tv[k][rel], tp[k][rel]
            m.mul_(0.9).add_(gr, alpha=0.1)
            v.mul_(0.999).addcmul_(gr, gr, value=0.001)
            bc1, bc2 = 1 - 0.9 ** step[0], 1 - 0.999 ** step[0]
            denom = (v.sqrt() / math.sqrt(bc2)).add_(1e-15)
            p.addcdiv_(m, denom, value=-lrs[k] / bc1)
            tm[k][rel] = m; tv[k][rel] = v; tp[k][rel] = p

    def fused_rows():
        rel = (grads["opacity"].flatten() != 0).nonzero().flatten().long()
        opt.step(rel)

    def fused_masked():
        opt.step_masked(grads["opacity"])

    t_torch = timed(torch_chain, 3, args.steps)
    t_rows = timed(fused_rows, 3, args.steps)
    t_mask = timed(fused_masked, 3, args.steps)
    t_dense = timed(lambda: opt.step(None), 3, args.steps)
    bytes_sel = n * 59 * 28
    return dict(leg="f-4 fused row-sparse Adam", rows=P, relevant=n, ms_torch_op_chain=t_torch,
                ms_fused_row_list_incl_nonzero=t_rows, ms_fused_masked=t_mask, ms_fused_dense=t_dense,
                speedup_masked=t_torch / t_mask,
                hbm={"bound": "hbm", "algorithmic_bytes": bytes_sel, "achieved_GBps": bytes_sel / t_mask / 1e6,
                     "peak_GBps": 8000.0, "frac": bytes_sel / t_mask / 1e6 / 8000.0,
                     "dense_achieved_GBps": P * 59 * 28 / t_dense / 1e6})


def leg_trainstep(args, dev):
    """A complete optimiser step at 1 M Gaussians / 1080p: SH colours of k views in one pass, k x (rasterize, L1 +
    inverse-depth loss, backward accumulating into the flat bucket), SH backward in one pass, fused Adam."""
    import diff_gaussian_rasterization as dgr
    from hgs import dp
    from hgs.optim import Adam
    W, H, k = 1920, 1080, 4
    base = synth.make_camera(W, H)
    scene = synth.make_scene(args.gaussians, base, seed=0).to(dev)
    cams = [synth.orbit_camera(W, H, j, k).to(dev) for j in range(k)]
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    params = {n: torch.nn.Parameter(getattr(scene, n).clone()) for n in names}
    lrs = dict(means3D=1.6e-5, shs=2.5e-3, opacities=1e-3, scales=1e-6, rotations=1e-5)
    bucket = dp.GradBucket({n: tuple(v.shape) for n, v in params.items()}, dev)
    for n in names:
        params[n].grad = bucket.views[n]
    opt = Adam([dict(params=[params[n]], lr=lrs[n], name=n) for n in names], lr=0.0, eps=1e-15)
    rasts, campos = [], []
    m2_grad = torch.empty(scene.P, 3, device=dev)
    d_rgbs = [torch.empty(scene.P, 3, device=dev) for _ in cams]
    rc = dgr.RasterContext(grad_buffers=dict(bucket.views, means2D=m2_grad),
                           backward_stream=torch.cuda.Stream(device=dev))   # backwards next to the following view's forward + loss
    for c in cams:
        rasts.append(dgr.GaussianRasterizer(settings(dgr, c, dev), context=rc))
        campos.append(c.camera_center)
    g = torch.Generator(device=dev).manual_seed(1)
    targets = [(torch.rand(3, H, W, device=dev, generator=g), torch.rand(1, H, W, device=dev, generator=g) * 0.3)
               for _ in range(k)]
    means2D = torch.zeros(scene.P, 3, device=dev, requires_grad=True)

    def step():
        with torch.no_grad():
            rgbs, clamps = dgr.sh_colors_batched(params["means3D"], params["shs"], 3, campos)
        for j, rast in enumerate(rasts):
            rc.grad_accumulate = j > 0
            rc.grad_buffers["colors_precomp"] = d_rgbs[j]
            color, radii, invd = rast(means3D=params["means3D"], means2D=means2D,
                                      colors_precomp=rgbs[j].requires_grad_(True), opacities=params["opacities"],
                                      scales=params["scales"], rotations=params["rotations"])
            loss = (color - targets[j][0]).abs().mean() + 0.1 * (invd - targets[j][1]).abs().mean()
            loss.backward()                                  # every gradient lands in a buffer
        rc.sh_colors_batched_backward(params["means3D"], params["shs"], 3, campos, clamps, d_rgbs,
                                      bucket.views["shs"], bucket.views["means3D"])
        rc.wait_backward_stream()
        opt.step(None)

    t = timed(step, args.warmup, args.steps)
    return dict(leg="training step (4 views fwd+bwd with L1 + inverse-depth loss, batched SH ends, two streams, fused Adam)",
                gaussians=scene.P, image=[W, H], views_per_step=k, ms_per_step=t, steps_per_s=1e3 / t,
                views_per_s=k * 1e3 / t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--leaves", type=int, default=500_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--legs", default="raw,lod,adam,train,post")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench_next.py needs a GPU (no CPU fallback)")
    dev = torch.device("cuda:0")
    for name in args.legs.split(","):
        res = {"raw": leg_raw, "lod": leg_lod, "adam": leg_adam, "train": leg_trainstep, "post": leg_post}[name](args, dev)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
