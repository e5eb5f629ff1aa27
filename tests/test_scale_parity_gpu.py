"""Value-level parity at the BASELINE scales: pixels AND every gradient tensor of the HIP op against the float64
oracle at 300 k / 1080p (config 2), 1 M / 1080p (the metric configuration), the render_post call shape at 1080p
(config 3: interpolation_weights / num_node_kids non-empty, do_depth False) and 8 M / 3840x2160 (towards config 5).

The call being matched: /root/reference/train_single.py:97,123 (forward + ``loss.backward()`` at full resolution)
and train_post.py:119-142.

A dense oracle over a whole 1080p frame is out of reach on a CPU (2e12 pixel x Gaussian pairs), so the comparison is
TILE-SAMPLED, and exact in what it covers:
  * ``n_tiles`` tiles are drawn: half at random (seeded), half the MOST CROWDED tiles of the frame (longest lists:
    the 64-entry batch boundaries, the prezero partition and the emission-slot arithmetic of the backward all see
    their worst case there);
  * the upstream gradients dL/dcolor, dL/dinvdepth are zeroed OUTSIDE the sampled tiles, the HIP op runs forward and
    backward over the WHOLE frame;
  * the oracle runs on the sub-scene of exactly those Gaussians whose instance lists reach a sampled tile (rows in
    their original order, so every sampled tile's sorted list is the same sequence), restricted to the sampled tiles;
  * pixels of the sampled tiles (fragile ones excluded and counted) and ALL gradient tensors incl. means2D are compared
    row by row for the sub-scene; every other row of the HIP gradients must be EXACTLY zero.
The integer side (radii, tile rectangles, depth bits, offsets, sorted (tile | depth) key sequence, point list, tile
ranges) is compared bit-exactly over the whole frame against the float32 geometry / binning specification.
"""
import json
import os

import numpy as np
import pytest
import torch

import parity as pa
from hgs import synth
from oracle import raster_oracle as ro

pytestmark = pytest.mark.gpu

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
F32_ORACLE = os.environ.get("HGS_F32_ORACLE", "1") != "0"     # the float32-oracle reference figures (costs one more oracle pass)


def _log(payload):
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "scale_parity.jsonl"), "a") as f:
            f.write(json.dumps(payload, default=float) + "\n")
    except OSError:
        pass


def _tile_mask(tiles, W, H):
    gx = (W + 15) // 16
    m = torch.zeros(H, W, dtype=torch.bool)
    for t in tiles:
        y0, x0 = (t // gx) * 16, (t % gx) * 16
        m[y0:y0 + 16, x0:x0 + 16] = True
    return m


def _run_case(name, gpu, P, W, H, n_tiles, *, lod=False, do_depth=True, seed=0, bg=(0.05, 0.1, 0.15), prepared=None,
              cam=None):
    cam = synth.make_camera(W, H) if cam is None else cam
    gc, gd = synth.upstream_grads(H, W, seed=seed + 1)
    bg = torch.tensor(bg)
    w = kids = None
    if prepared is not None:     # (scene, interpolation_weights, num_node_kids) built by the caller
        scene, w, kids = prepared
        P = scene.P
    else:
        scene = synth.make_scene(P, cam, seed=seed)
    if lod and prepared is None:  # render_post's call shape: [>= P] weights / sibling counts, opacities may exceed 1
        g = torch.Generator().manual_seed(seed + 2)
        scene.opacities = scene.opacities * 1.3
        w = torch.rand(P + 100, generator=g)
        kids = torch.randint(1, 9, (P + 100,), generator=g, dtype=torch.int32)

    # ---- whole frame, integers: float32 geometry + binning specification --------------------------------------
    geom = ro.geometry_spec(scene.means3D.numpy(), scene.scales.numpy(), scene.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            float(np.float32(cam.tanfovx)), float(np.float32(cam.tanfovy)), 1.0)
    binning = ro.binning_spec(geom)
    T = geom.grid[0] * geom.grid[1]
    per_tile = (binning.ranges[:, 1] - binning.ranges[:, 0]).astype(np.int64)

    # ---- the tile sample ------------------------------------------------------------------------------------------
    rng = np.random.default_rng(1000 + seed)
    heavy = np.argsort(-per_tile, kind="stable")[:n_tiles // 2]
    rest = np.setdiff1d(np.arange(T), heavy)
    tiles = sorted(set(heavy.tolist()) | set(rng.choice(rest, size=n_tiles - len(heavy), replace=False).tolist()))
    mask = _tile_mask(tiles, W, H)
    gc_m, gd_m = gc * mask, gd * mask

    # ---- oracle on the sub-scene that reaches the sampled tiles ---------------------------------------------------
    sub = np.unique(np.concatenate([binning.point_list[binning.ranges[t, 0]:binning.ranges[t, 1]] for t in tiles]))
    sub_t = torch.from_numpy(sub.astype(np.int64))
    sub_scene = synth.Scene(scene.means3D[sub_t], scene.scales[sub_t], scene.rotations[sub_t],
                            scene.opacities[sub_t], scene.shs[sub_t], scene.sh_degree)
    req = lambda t: t.clone().requires_grad_(True)
    m3, sc, rot, op, sh = map(req, (sub_scene.means3D, sub_scene.scales, sub_scene.rotations, sub_scene.opacities,
                                    sub_scene.shs))
    m2 = torch.zeros(sub_scene.P, 3, requires_grad=True)
    oo = ro.rasterize(m3, m2, sh, None, op, sc, rot, None, image_height=H, image_width=W, tanfovx=cam.tanfovx,
                      tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                      projmatrix=cam.full_proj_transform, sh_degree=scene.sh_degree, campos=cam.camera_center,
                      interpolation_weights=None if w is None else w[sub_t],
                      num_node_kids=None if kids is None else kids[sub_t], tiles=tiles)
    # the sub-scene's lists in the sampled tiles are the full frame's lists (same Gaussians, same order)
    for t in tiles:
        a = binning.point_list[binning.ranges[t, 0]:binning.ranges[t, 1]]
        b = sub[oo.binning.point_list[oo.binning.ranges[t, 0]:oo.binning.ranges[t, 1]]]
        assert np.array_equal(a, b), f"{name}: tile {t}: sub-scene list differs from the frame's"
    # pixels with a blend decision on a knife edge leave BOTH losses (tests/parity.py does the same for the small scenes;
    # until round 6 this harness only left them out of the PIXEL comparison: on rows the reference's scripts trained, one
    # such pixel -- 2e-4 off, flagged -- put 2e-5 on the gradients of the four Gaussians around it)
    ok_loss = mask & torch.from_numpy(~oo.fragile)
    gc_m, gd_m = gc * ok_loss, gd * ok_loss
    loss = (oo.color * gc_m.double()).sum()
    if do_depth:
        loss = loss + (oo.invdepth * gd_m.double()).sum()
    loss.backward(retain_graph=F32_ORACLE)
    og = {k: (None if v is None else v.clone()) for k, v in
          dict(means3D=m3.grad, means2D=m2.grad, opacities=op.grad, shs=sh.grad, scales=sc.grad, rotations=rot.grad).items()}

    # ---- HIP: whole frame, forward + backward ---------------------------------------------------------------------
    hip = pa.run_hip(scene, cam, bg, gc_m, gd_m, gpu, interpolation_weights=w, num_node_kids=kids,
                     do_depth=do_depth, debug=False, grad_mask=None)

    class _O:      # what check_indices expects of an oracle output
        pass
    oo_int = _O()
    oo_int.geom, oo_int.binning = geom, binning
    idx = pa.check_indices(hip, oo_int)
    assert all(v == 0 for v in idx.values()), f"{name}: integer mismatch over the whole frame {idx}"

    # ---- whose error is it?  The SAME oracle with the kernels' precision split (per-Gaussian stage in float64, blend and
    # its backward in float32) against the float64 oracle: what float32 arithmetic costs on this scene whatever the
    # implementation.  Fragile pixels leave both losses (their decisions may differ by precision).
    f32 = {}
    if F32_ORACLE:
        okd = torch.from_numpy(~oo.fragile)
        m3f, scf, rotf, opf, shf = map(req, (sub_scene.means3D, sub_scene.scales, sub_scene.rotations,
                                             sub_scene.opacities, sub_scene.shs))
        m2f = torch.zeros(sub_scene.P, 3, requires_grad=True)
        of = ro.rasterize(m3f, m2f, shf, None, opf, scf, rotf, None, image_height=H, image_width=W, tanfovx=cam.tanfovx,
                          tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                          projmatrix=cam.full_proj_transform, sh_degree=scene.sh_degree, campos=cam.camera_center,
                          interpolation_weights=None if w is None else w[sub_t],
                          num_node_kids=None if kids is None else kids[sub_t], tiles=tiles, dtype=torch.float32,
                          geom_dtype=torch.float64)
        okf = okd & torch.from_numpy(~of.fragile)
        for nm, a_, b_ in (("color", of.color, oo.color), ("invdepth", of.invdepth, oo.invdepth)):
            st = pa.err_stats(a_.detach()[:, okf & mask], b_.detach()[:, okf & mask])
            f32[nm] = dict(mixed=st["mixed"], p999_rel=st["p999_rel"], maxrel=st["maxrel"], l2=st["l2"])
        lf = (of.color * (gc_m * okf).to(of.color.dtype)).sum()
        if do_depth:
            lf = lf + (of.invdepth * (gd_m * okf).to(of.invdepth.dtype)).sum()
        lf.backward()
        # the float64 gradients of the same (fragile-free) loss
        for t in (m3, m2, op, sh, sc, rot):
            t.grad = None
        l64 = (oo.color * (gc_m * okf).double()).sum()
        if do_depth:
            l64 = l64 + (oo.invdepth * (gd_m * okf).double()).sum()
        l64.backward()
        ref64 = dict(means3D=m3.grad, means2D=m2.grad, opacities=op.grad, shs=sh.grad, scales=sc.grad, rotations=rot.grad)
        got32 = dict(means3D=m3f.grad, means2D=m2f.grad, opacities=opf.grad, shs=shf.grad, scales=scf.grad,
                     rotations=rotf.grad)
        for k in ref64:
            st = pa.err_stats(got32[k], ref64[k])
            f32["d_" + k] = dict(mixed=st["mixed"], p999_rel=st["p999_rel"], maxrel=st["maxrel"], l2=st["l2"])

    # ---- compare ---------------------------------------------------------------------------------------------------
    ok = mask & torch.from_numpy(~oo.fragile)
    stats = {"fragile_frac": float(oo.fragile[mask.numpy()].mean()), "rows_touching_fragile": pa.rows_touching_fragile(oo)}
    stats["color"] = pa.err_stats(hip["color"][:, ok], oo.color.detach()[:, ok])
    if do_depth:
        stats["invdepth"] = pa.err_stats(hip["invdepth"][:, ok], oo.invdepth.detach()[:, ok])
    nc_h = hip["views"]["n_contrib"][ok].numpy()
    stats["n_contrib_mismatch"] = int((nc_h != oo.n_contrib[ok.numpy()]).sum())
    outside = torch.ones(P, dtype=torch.bool)
    outside[sub_t] = False
    nonzero_outside = {}
    for k, g in og.items():
        hg = hip["grads"][k]
        stats["d_" + k] = pa.err_stats(hg[sub_t], g)
        nonzero_outside[k] = int((hg[outside] != 0).reshape(int(outside.sum()), -1).any(dim=1).sum())
    payload = dict(case=name, P=P, W=W, H=H, L=int(binning.num_rendered), tiles=len(tiles),
                   tile_instances_sampled=int(per_tile[tiles].sum()), longest_list=int(per_tile.max()),
                   sub_scene=int(sub.shape[0]), indices=idx, stats=stats, nonzero_rows_outside=nonzero_outside,
                   float32_oracle_vs_float64=f32)
    _log(payload)
    print(json.dumps(payload, default=float))

    def dump_case(dump):        # everything needed to look at a case offline: the sub-scene, the sample, both sides' results
        os.makedirs(dump, exist_ok=True)
        np.savez_compressed(
            os.path.join(dump, "".join(c if c.isalnum() else "_" for c in name)[:80] + ".npz"),
            sub=sub, tiles=np.asarray(tiles), W=W, H=H, viewmatrix=cam.world_view_transform.numpy(),
            projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(), tanfovx=cam.tanfovx,
            tanfovy=cam.tanfovy, bg=bg.numpy(), seed=seed, do_depth=do_depth,
            **{"in_" + k: getattr(sub_scene, k).numpy() for k in ("means3D", "scales", "rotations", "opacities", "shs")},
            **({} if w is None else {"w": w[sub_t].numpy(), "kids": kids[sub_t].numpy()}),
            **{"hip_d_" + k: hip["grads"][k][sub_t].numpy() for k in og}, **{"ora_d_" + k: v.numpy() for k, v in og.items()},
            hip_color=hip["color"][:, mask].numpy(), ora_color=oo.color.detach()[:, mask].numpy(),
            fragile=oo.fragile[mask.numpy()], mask_idx=np.flatnonzero(mask.numpy().reshape(-1)))
    dump = os.environ.get("HGS_PARITY_DUMP")
    if dump and os.environ.get("HGS_PARITY_DUMP_ON_FAIL", "0") != "1":
        dump_case(dump)
    try:
        # every candidate of a pixel has its own small chance of sitting inside its band (which grows with the footprint,
        # oracle FRAGILE_FP32_K): 1e-3 of the pixels for the short lists of the benchmark scenes, 4e-6 per list entry beyond
        # (measured 1.8e-3 on the 64 sampled tiles of the synthetic trained-scale frame, 2.6e-3 .. 4.8e-3 on rows the
        # reference's scripts trained: 1 200 .. 1 700 entries on average, footprints of hundreds of pixels)
        assert stats["fragile_frac"] <= max(pa.FRAGILE_FRAC, 4e-6 * payload["tile_instances_sampled"] / len(tiles))
        assert stats["n_contrib_mismatch"] == 0
        assert all(v == 0 for v in nonzero_outside.values()), nonzero_outside
        # Norm-wise 1e-5 and the element-wise bound on everything -- or, per tensor, as good as the float32 ORACLE is against
        # float64 on the same tiles (a float32 sum over a tile's pixels and a list's thousands of entries has that error
        # whoever evaluates it: on rows the reference's scripts TRAINED the float32 oracle itself is at 2e-5 norm-wise on
        # d_scales / d_rotations, profiles/r06_trained_rows_parity.md).  "As good as": the relative L2 error -- a mean over
        # 1e5 .. 1e6 entries -- within 1.5 x the float32 oracle's; the two MAXIMA over those entries (max-rel, the
        # element-wise figure) within 3 x: the largest of a million rounding errors differs by that much between two
        # float32 evaluation orders (28 tensors of four trained-row cases, profiles/r06_trained_rows_parity.md: HIP's
        # rel-L2 is the smaller one in 26, ratio 0.35 .. 1.05; the ratios of the maxima spread over 0.19 .. 1.65).
        for k, v in stats.items():
            if not isinstance(v, dict):
                continue
            f = f32.get(k, {})
            assert v["l2"] <= max(pa.REL_TOL, 1.5 * f.get("l2", 0.0)), f"{name}: {k} rel-L2 error {v['l2']:.3e} (float32 oracle {f.get('l2', 0.0):.3e})"
            assert v["maxrel"] <= max(pa.REL_TOL, 3.0 * f.get("maxrel", 0.0), 1.5 * f.get("l2", 0.0)), \
                f"{name}: {k} max error {v['maxrel']:.3e} rel. to max (float32 oracle {f.get('maxrel', 0.0):.3e})"
            assert v["mixed"] <= max(1.0, 3.0 * f.get("mixed", 0.0)), \
                f"{name}: {k} element-wise error {v['mixed']:.3f} x the bound (float32 oracle {f.get('mixed', 0.0):.3f})"
    except AssertionError:
        if dump and os.environ.get("HGS_PARITY_DUMP_ON_FAIL", "0") == "1":
            dump_case(dump)
        raise


def test_config2_300k_1080p(gpu):
    """BASELINE.json configs[1]: ~300 k Gaussians at 1080p, train_single.py fwd+bwd."""
    _run_case("config2_300k_1080p", gpu, 300_000, 1920, 1080, 96)


def test_metric_config_1m_1080p(gpu):
    """The configuration the metric is quoted on: 1 M Gaussians at 1080p."""
    _run_case("metric_1m_1080p", gpu, 1_000_000, 1920, 1080, 128)


def test_config3_shape_lod_tensors_1080p(gpu):
    """render_post's call shape (gaussian_renderer/__init__.py:247-277): interpolation_weights / num_node_kids
    non-empty and longer than P, do_depth False, abs-activated opacities above 1."""
    _run_case("config3_shape_lod_500k_1080p", gpu, 500_000, 1920, 1080, 96, lod=True, do_depth=False, seed=5,
              bg=(0.0, 0.0, 0.0))


def test_4k_8m_backward(gpu):
    """8 M Gaussians at 3840x2160 (L ~ 21 M): the backward at 4K, which round 1 never compared."""
    _run_case("4k_8m", gpu, 8_000_000, 3840, 2160, 96, seed=0)


def test_config3_merged_two_chunk_hierarchy_1080p(gpu):
    """BASELINE.json configs[2] for real: a merged 2-chunk hierarchy (hgs.hierarchy.merge_hierarchies -- the shape
    GaussianHierarchyMerger produces, scripts/full_train.py:240-250), cut by expand_to_size at a train_post-range
    threshold (train_post.py:66-74), weights from get_interpolation_weights, attribute lerp as render_post does it
    (gaussian_renderer/__init__.py:204-218), then the op at 1080p with do_depth False.  Cut and weights are compared
    bit-exactly with the LOD oracle; pixels and every gradient w.r.t. the interpolated rows tile-sampled against the
    raster oracle."""
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
    from hgs import hierarchy
    from oracle import lod_oracle as lo
    W, H = 1920, 1080
    cam = synth.make_camera(W, H)
    full = synth.make_scene(300_000, cam, seed=21)
    left = full.means3D[:, 0] < 0
    chunks = [hierarchy.build_hierarchy(synth.Scene(full.means3D[m], full.scales[m], full.rotations[m],
                                                    full.opacities[m], full.shs[m], 3)) for m in (left, ~left)]
    h = hierarchy.merge_hierarchies(chunks)
    G = h.num_nodes
    assert G == 1 + sum(c.num_nodes for c in chunks)
    nodes, boxes = h.nodes.to(gpu), h.boxes.to(gpu)
    ri = torch.zeros(G, dtype=torch.int32, device=gpu); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
    wt = torch.zeros(G, device=gpu); ns = torch.zeros(G, dtype=torch.int32, device=gpu)
    tau = 0.006                                   # inside train_post's log-uniform [0.005, 0.1] range: a fine cut
    vp = cam.camera_center
    n = expand_to_size(nodes, boxes, tau, vp.to(gpu), torch.zeros(3), ri, pi, ni)
    get_interpolation_weights(ni[:n], tau, nodes, boxes, vp.cpu(), torch.zeros(3), wt, ns)
    r_o, p_o, n_o = lo.expand_to_size(h.nodes.numpy(), h.boxes.numpy(), tau, vp.numpy())
    w_o, k_o = lo.get_interpolation_weights(n_o, tau, h.nodes.numpy(), h.boxes.numpy(), vp.numpy())
    assert n == len(r_o) and 50_000 < n < G
    assert np.array_equal(ri[:n].cpu().numpy(), r_o) and np.array_equal(pi[:n].cpu().numpy(), p_o)
    assert np.array_equal(ni[:n].cpu().numpy(), n_o)
    assert np.array_equal(wt[:n].cpu().numpy().view(np.uint32), w_o.view(np.uint32))
    assert np.array_equal(ns[:n].cpu().numpy(), k_o)
    frac = float(((w_o > 0) & (w_o < 1)).mean())
    assert frac > 0.02, f"only {frac:.3f} of the cut is in transition: the case would not exercise the LOD opacity"
    # the rows render_post hands to the op (float32, torch's rounding)
    r, p = torch.from_numpy(r_o).long(), torch.from_numpy(p_o).long()
    t = torch.from_numpy(w_o).unsqueeze(1); ti = 1 - t
    rots_n, rots_p = torch.nn.functional.normalize(h.rots)[r], torch.nn.functional.normalize(h.rots)[p]
    flip = (rots_n * rots_p).sum(1) < 0
    rots_p = torch.where(flip[:, None], -rots_p, rots_p)
    sc = torch.exp(h.log_scales)
    rows = synth.Scene((t * h.xyz[r] + ti * h.xyz[p]).contiguous(), (t * sc[r] + ti * sc[p]).contiguous(),
                       (t * rots_n + ti * rots_p).contiguous(), (t * h.alpha.abs()[r] + ti * h.alpha.abs()[p]).contiguous(),
                       (t.unsqueeze(2) * h.shs[r] + ti.unsqueeze(2) * h.shs[p]).contiguous(), 3)
    weights = torch.zeros(G); weights[:n] = torch.from_numpy(w_o)            # [P_total] arrays, first n valid
    kids = torch.ones(G, dtype=torch.int32); kids[:n] = torch.from_numpy(k_o)
    _run_case("config3_merged_2chunk_hierarchy_1080p", gpu, n, W, H, 96, do_depth=False, seed=21, bg=(0.0, 0.0, 0.0),
              prepared=(rows, weights, kids))


def test_trained_scale_10m_1080p(gpu):
    """The workload the reference's own scripts produce (train_single.py:97-176 at 1080p: 375 k Gaussians, 28 instances
    per Gaussian, L ~ 10 M, lists beyond 3 000): pixels and EVERY gradient on the frame's most crowded tiles -- the
    long-run route of K8, the multi-wave classes of the per-tile depth sort and 3 000-entry lists in K6 / K7."""
    cam = synth.make_camera(1920, 1080)
    scene = synth.make_scene_trained_scale(375_000, cam, seed=0)
    _run_case("trained_scale_10m_1080p", gpu, scene.P, 1920, 1080, 64, seed=11, prepared=(scene, None, None))


def test_trained_scale_cut_order_1080p(gpu):
    """The same scene in the row order of a hierarchy cut (train_post.py:91-142, render_hierarchy.py:58-92: the cut's big
    nodes side by side -- one K1 workgroup's rows emit 840 000 instances while the mean is 7 000), render_post's call
    shape (do_depth False, LOD tensors): K3's shared emission and K8's dealt-out long runs against the oracle."""
    cam = synth.make_camera(1920, 1080)
    scene = synth.make_scene_trained_scale(375_000, cam, seed=0, order="clustered")
    g = torch.Generator().manual_seed(5)
    w = torch.rand(scene.P, generator=g)
    kids = torch.randint(1, 5, (scene.P,), generator=g, dtype=torch.int32)
    _run_case("trained_scale_cut_order_1080p", gpu, scene.P, 1920, 1080, 64, do_depth=False, seed=12, bg=(0.0, 0.0, 0.0),
              prepared=(scene, w, kids))
