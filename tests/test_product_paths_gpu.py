"""Product paths the round-2 suite did not drive ("untested product paths"):

  * the speculative single-call forward's CAPACITY MISS (``hgs_raster_fwd`` -> ``HGS_ERR_CAPACITY`` -> exact two-stage
    retry, diff_gaussian_rasterization/_C.py) -- plain call shape and the ``lod=`` shape key (coarse -> fine cut);
  * ``render_coarse``'s call shape (/root/reference/gaussian_renderer/__init__.py:296-389: M = 4 coefficients,
    ``debug=True`` always, ``do_depth`` False, skybox rows at the head) and the other short SH blocks, M in {1, 4, 9}
    (M = 9: the ``3M % 4 != 0`` scalar SH path of preprocess.hip);
  * 30 seeded cases of the soak tool tests/tools/fuzz_parity.py;
  * heavy footprints at 1080p, tile-sampled: ``s_px`` in [1, 8] (SURVEY App. C "heavy 1 M") and a trained-scene-like
    log-normal footprint distribution with needles (anisotropy 0.05, up to ~100 px).
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

import parity as pa
from hgs import synth
from test_raster_gpu import _assert_case
from test_scale_parity_gpu import _run_case

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


def _fwd_bwd(gpu, scene, cam, gc, gd, scale_modifier, bg, debug=False):
    import diff_gaussian_rasterization as dgr
    req = lambda t: t.clone().to(gpu).requires_grad_(True)
    m3, sc, rot, op, sh = map(req, (scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs))
    m2 = torch.zeros(scene.P, 3, device=gpu, requires_grad=True)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, bg, scene.sh_degree, debug=debug,
                                                                scale_modifier=scale_modifier, device=gpu))
    color, radii, invd = dgr.GaussianRasterizer(rs)(means3D=m3, means2D=m2, shs=sh, opacities=op, scales=sc, rotations=rot)
    call = color.grad_fn.call
    views = {k: v.clone() for k, v in dgr._C.raster_views(call).items()}
    ((color * gc.to(gpu)).sum() + (invd * gd.to(gpu)).sum()).backward()
    torch.cuda.synchronize()
    grads = dict(means3D=m3.grad, means2D=m2.grad, shs=sh.grad, opacities=op.grad, scales=sc.grad, rotations=rot.grad)
    return dict(color=color.detach(), radii=radii, invdepth=invd.detach(), views=views, grads=grads, L=call.L,
                L_ws=call.L_ws)


def test_capacity_miss_retry_is_exact_at_default_slack(gpu):
    """Same (W, H, P), the second view has > 1.25 x + 64 Ki the instances of the first (scale_modifier 0.2 -> 1.6):
    the speculative forward overflows its capacity, returns HGS_ERR_CAPACITY after having run every kernel on the
    clamped list, and the op finishes on the exact two-stage path.  The retried call must be bit-identical -- integers,
    pixels, every gradient -- to the same view rendered with speculation switched off."""
    import diff_gaussian_rasterization as dgr
    C = dgr._C
    W, H, P = 640, 368, 120_000
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=11, s_px=(1.0, 6.0))
    gc, gd = synth.upstream_grads(H, W, seed=12)
    bg = torch.tensor([0.2, 0.1, 0.3])
    C._last_L.clear()
    small = _fwd_bwd(gpu, scene, cam, gc, gd, 0.2, bg)            # first call of the shape: two-stage, records L
    assert small["L_ws"] == small["L"]
    cap = int(small["L"] * C.SPEC_GROWTH) + C.SPEC_SLACK
    misses0, spec0 = C.stats["capacity_misses"], C.stats["speculative_calls"]
    big = _fwd_bwd(gpu, scene, cam, gc, gd, 1.6, bg)              # speculative, overflows, retried
    assert C.stats["speculative_calls"] == spec0 + 1 and C.stats["capacity_misses"] == misses0 + 1
    assert big["L"] > cap, (big["L"], cap)
    assert big["L_ws"] == big["L"]                                # finished on the exact path
    again = _fwd_bwd(gpu, scene, cam, gc, gd, 1.6, bg)            # now the speculative path fits
    assert C.stats["capacity_misses"] == misses0 + 1 and again["L_ws"] > again["L"]
    C.SPECULATIVE = False
    try:
        ref = _fwd_bwd(gpu, scene, cam, gc, gd, 1.6, bg)
    finally:
        C.SPECULATIVE = True
    for name, got in (("retried", big), ("speculative", again)):
        assert got["L"] == ref["L"]
        assert torch.equal(got["radii"], ref["radii"]), name
        for k in ("point_list", "tile_ids_sorted", "ranges", "offsets", "n_contrib"):
            assert torch.equal(got["views"][k], ref["views"][k]), (name, k)
        assert torch.equal(got["color"], ref["color"]) and torch.equal(got["invdepth"], ref["invdepth"]), name
        for k, g in ref["grads"].items():
            assert torch.equal(got["grads"][k], g), (name, k)


def test_capacity_miss_retry_matches_oracle(gpu, monkeypatch):
    """The same miss at a size the dense oracle renders whole (slack lowered so that a 4 k-Gaussian frame can overflow):
    indices exact, pixels and gradients within 1e-5 of the oracle on the RETRIED call."""
    import diff_gaussian_rasterization as dgr
    C = dgr._C
    monkeypatch.setattr(C, "SPEC_SLACK", 64)
    cam, scene, gc, gd = pa.default_case(4000, 208, 144, seed=31)
    bg = torch.tensor([0.1, 0.2, 0.3])
    C._last_L.clear()
    pa.run_hip(scene, cam, bg, gc, gd, gpu, scale_modifier=0.3, debug=False, grad_mask=None)
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd, scale_modifier=1.5)     # (before its run_hip: fragile-pixel pairing)
    misses0 = C.stats["capacity_misses"]
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu, scale_modifier=1.5, debug=False)
    assert C.stats["capacity_misses"] == misses0 + 1
    _assert_case("capacity_retry", hip, oo, og)


def test_capacity_miss_on_the_lod_shape_key(gpu, monkeypatch):
    """In-op LOD path (shape key = the hierarchy, not the cut): a fine cut (many small Gaussians, few tiles each)
    followed by a coarse one of the same hierarchy (few nodes, each covering many tiles: 3.6 x the instances here)
    overflows the speculative capacity; the retried render must equal the non-speculative one bit for bit, forward
    and backward."""
    import diff_gaussian_rasterization as dgr
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
    from hgs import hierarchy
    C = dgr._C
    monkeypatch.setattr(C, "SPEC_SLACK", 64)
    W, H = 320, 208
    cam = synth.make_camera(W, H)
    h = hierarchy.build_hierarchy(synth.make_scene(6000, cam, seed=5, s_px=(0.7, 3.0)))
    nodes, boxes = h.nodes.to(gpu), h.boxes.to(gpu)
    G = h.xyz.shape[0]
    ri = torch.zeros(G, dtype=torch.int32, device=gpu); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
    w = torch.zeros(G, device=gpu); ns = torch.zeros(G, dtype=torch.int32, device=gpu)
    gc = synth.upstream_grads(H, W)[0].to(gpu)
    attrs = dict(xyz=h.xyz, shs=h.shs, op=h.alpha.abs().reshape(-1, 1), sc=torch.exp(h.log_scales),
                 rot=torch.nn.functional.normalize(h.rots))

    def render(tau_px):
        tau = (2 * tau_px + 1) * cam.tanfovx / (0.5 * W)
        n = expand_to_size(nodes, boxes, tau, cam.camera_center.to(gpu), torch.zeros(3), ri, pi, ni)
        get_interpolation_weights(ni[:n], tau, nodes, boxes, cam.camera_center.cpu(), torch.zeros(3), w, ns)
        L = {k: v.to(gpu).contiguous().requires_grad_(True) for k, v in attrs.items()}
        kw = pa.settings_kwargs(cam, torch.zeros(3), 3, do_depth=False, device=gpu, interpolation_weights=w, num_node_kids=ns)
        kw["render_indices"], kw["parent_indices"] = ri[:n].contiguous(), pi
        m2 = torch.zeros(G, 3, device=gpu, requires_grad=True)
        c, radii, _ = dgr.GaussianRasterizer(dgr.GaussianRasterizationSettings(**kw))(
            means3D=L["xyz"], means2D=m2, shs=L["shs"], opacities=L["op"], scales=L["sc"], rotations=L["rot"])
        num = c.grad_fn.num_rendered
        (c * gc).sum().backward()
        torch.cuda.synchronize()
        return n, num, c.detach(), radii, {k: v.grad for k, v in L.items()}, m2.grad

    C._last_L.clear()
    n_fine, L_fine, *_ = render(0.0)
    misses0 = C.stats["capacity_misses"]
    n_coarse, L_coarse, c1, r1, g1, m1 = render(40.0)
    assert n_fine > 2 * n_coarse and L_coarse > int(L_fine * C.SPEC_GROWTH) + 64, (n_coarse, n_fine, L_coarse, L_fine)
    assert C.stats["capacity_misses"] == misses0 + 1
    C.SPECULATIVE = False
    try:
        _, L_ref, c2, r2, g2, m2 = render(40.0)
    finally:
        C.SPECULATIVE = True
    assert L_ref == L_coarse and torch.equal(c1, c2) and torch.equal(r1, r2) and torch.equal(m1, m2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


@pytest.mark.parametrize("deg_stored,deg_active", [(0, 0), (1, 1), (2, 2), (2, 1), (1, 0)])
def test_short_sh_blocks_with_debug(gpu, deg_stored, deg_active):
    """M = (deg_stored + 1)^2 in {1, 4, 9} coefficients per Gaussian, active degree <= stored, ``debug=True``
    (synchronous checking after every kernel), depth channel on: value parity with the oracle."""
    cam, scene, gc, gd = pa.default_case(3000, 272, 176, seed=40 + deg_stored, sh_degree=deg_stored)
    assert scene.shs.shape[1] == (deg_stored + 1) ** 2
    scene.sh_degree = deg_active
    bg = torch.tensor([0.3, 0.0, 0.6])
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu, debug=True)
    _assert_case(f"sh_M{scene.shs.shape[1]}_deg{deg_active}_debug", hip, oo, og)


def test_render_coarse_call_shape(gpu):
    """render_coarse (/root/reference/gaussian_renderer/__init__.py:296-407) as train_coarse.py:94 drives it:
    GaussianModel(1) -> 4 SH coefficients, sh_degree 1, debug=True unconditionally (:331), do_depth left at its
    default, random background, the skybox rows at the HEAD of the arrays (far, large, README.md:490); the third
    output is ignored (:381) and the visibility filter is the bool mask ``radii > 0`` (:393-407)."""
    import diff_gaussian_rasterization as dgr
    W, H = 400, 240
    cam = synth.make_camera(W, H)
    body = synth.make_scene(5000, cam, seed=50, sh_degree=1)
    sky = synth.make_scene(600, cam, seed=51, sh_degree=1, s_px=(8.0, 30.0), z_range=(40.0, 60.0))
    cat = lambda a, b: torch.cat((a, b)).contiguous()
    scene = synth.Scene(cat(sky.means3D, body.means3D), cat(sky.scales, body.scales), cat(sky.rotations, body.rotations),
                        cat(sky.opacities, body.opacities), cat(sky.shs, body.shs), 1)
    gc, gd = synth.upstream_grads(H, W, seed=52)
    bg = torch.rand(3, generator=torch.Generator().manual_seed(53))
    oo, og = pa.run_oracle(scene, cam, bg, gc, gd, do_depth=False)
    hip = pa.run_hip(scene, cam, bg, gc, gd, gpu, do_depth=False, debug=True)
    _assert_case("render_coarse_shape", hip, oo, og, do_depth=False)
    # the reference builds the settings WITHOUT do_depth (:319-337): the field must default to False and the third
    # output must still be a [1,H,W] tensor
    kw = pa.settings_kwargs(cam, bg, 1, debug=True, device=gpu)
    kw.pop("do_depth")
    rs = dgr.GaussianRasterizationSettings(**kw)
    assert rs.do_depth is False
    s = scene.to(gpu)
    m2 = torch.zeros(scene.P, 3, device=gpu, requires_grad=True)
    color, radii, third = dgr.GaussianRasterizer(raster_settings=rs)(
        means3D=s.means3D, means2D=m2, shs=s.shs, colors_precomp=None, opacities=s.opacities, scales=s.scales,
        rotations=s.rotations, cov3D_precomp=None)
    assert tuple(third.shape) == (1, H, W) and float(third.abs().sum()) == 0.0
    assert torch.equal(color.detach().cpu(), hip["color"])
    vis = radii > 0
    assert vis.dtype == torch.bool and int(vis[:600].sum()) > 0 and int(vis.sum()) > 3000


def test_fuzz_subset(gpu):
    """30 seeded cases of tests/tools/fuzz_parity.py (random sizes that are not tile multiples, fields of view, SH
    degrees / block sizes, footprints from 0.05 to 150 px, opaque stacks, backgrounds, scale modifiers, depth on / off,
    SH or precomputed colours): no index mismatch, every quantity within 1e-5."""
    import fuzz_parity
    rep = fuzz_parity.run_cases(30, 1000, gpu)
    print(json.dumps(rep, default=str))
    assert not rep["index_mismatches"], rep["index_mismatches"]
    assert not rep["above_tolerance"], rep["above_tolerance"]


def test_heavy_footprints_1m_1080p(gpu):
    """SURVEY App. C "heavy 1 M": s_px in [1, 8] -> L ~ 4.9 M, ~600 instances per tile on average, the most crowded
    tiles beyond the 512-entry in-register sort."""
    cam = synth.make_camera(1920, 1080)
    scene = synth.make_scene(1_000_000, cam, seed=0, s_px=(1.0, 8.0))
    _run_case("heavy_1m_spx_1_8_1080p", gpu, scene.P, 1920, 1080, 64, prepared=(scene, None, None))


def test_trained_like_footprints_1080p(gpu):
    """Log-normal footprints up to ~100 px with needles (anisotropy 0.05) and bimodal opacities: a few Gaussians cover
    hundreds of tiles, tile lists run into the thousands."""
    cam = synth.make_camera(1920, 1080)
    scene = synth.make_scene_trained_like(400_000, cam, seed=3)
    _run_case("trained_like_400k_1080p", gpu, scene.P, 1920, 1080, 48, seed=3, prepared=(scene, None, None))


_FRAME_DIGEST = r"""
import hashlib, os, sys
root = sys.argv[1]
for p in (root, os.path.join(root, "hierarchical-3d-gaussians_amd"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import torch
import parity as pa
from hgs import synth
cam = synth.make_camera(200, 120)
scene = synth.make_scene(3000, cam, seed=5)
gc, gd = synth.upstream_grads(120, 200, seed=1)
r = pa.run_hip(scene, cam, torch.zeros(3), gc, gd, "cuda:0", debug=False, grad_mask=None)
h = hashlib.sha256()
for t in [r["color"], r["radii"], r["invdepth"]] + [r["grads"][k] for k in sorted(r["grads"])]:
    h.update(t.contiguous().numpy().tobytes())
print("digest", h.hexdigest(), r["L"])
"""


def test_diagnostic_wait_and_count_routes_give_the_same_frame(gpu):
    """HGS_BLOCKING_WAIT (sleeping host waits instead of polling), HGS_COUNT_BY_COPY (the instance count by a copy
    command instead of a kernel's store into mapped host memory), HGS_SCAN_LAUNCH (the workgroup sums scanned by a launch
    between K1 and K3 -- the route of rounds 1-4, kept for the radix path and very large P -- instead of K1's superblock
    totals finished by K3), HGS_SCAN_SPLIT (that launch once per array) and HGS_K8_PRESUM (long runs of instance records
    summed by kernels of their own whatever the frame's mean run) are read once per process: one small
    fwd+bwd per setting in a process of its own, bit-identical outputs and gradients."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for name, env in (("default", {}), ("blocking", {"HGS_BLOCKING_WAIT": "1"}), ("copy", {"HGS_COUNT_BY_COPY": "1"}),
                      ("split", {"HGS_SCAN_SPLIT": "1", "HGS_SCAN_LAUNCH": "1"}), ("scanlaunch", {"HGS_SCAN_LAUNCH": "1"}),
                      ("k8_presum", {"HGS_K8_PRESUM": "1"})):
        r = subprocess.run([sys.executable, "-c", _FRAME_DIGEST, root], env={**os.environ, **env}, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        digests[name] = [l for l in r.stdout.splitlines() if l.startswith("digest")][-1]
    assert len(set(digests.values())) == 1, digests


def test_culled_workgroups_and_active_degrees_on_the_half_row_route(gpu):
    """K1 at M = 16 stages the SH block in two halves straight into LDS (round 5).  Three things the usual scenes do not
    reach: workgroups whose Gaussians are mostly off screen (per-lane loads of the visible rows instead of the DMA), a
    last workgroup that is not full, and active degrees 0 / 1 (the second half is never fetched) and 2 -- each against the
    oracle, indices bit-exact."""
    cam = synth.make_camera(160, 96)
    for deg, P, spread in ((3, 2000 + 37, 4.0), (0, 1500, 1.0), (1, 1500 + 255, 2.5), (2, 1024 + 1, 1.0)):
        scene = synth.make_scene(P, cam, seed=40 + deg, sh_degree=3)
        scene.sh_degree = deg                                   # stored M = 16, active degree deg
        # spread > 1: most Gaussians leave the frustum sideways, in random order -> every workgroup is "mostly culled"
        scene.means3D[:, :2] *= spread
        gc, gd = synth.upstream_grads(96, 160, seed=3)
        bg = torch.tensor([0.2, 0.1, 0.3])
        oo, og = pa.run_oracle(scene, cam, bg, gc, gd)
        hip = pa.run_hip(scene, cam, bg, gc, gd, gpu)
        idx = pa.check_indices(hip, oo)
        assert all(v == 0 for v in idx.values()), (deg, idx)
        vis = float((oo.geom.radii > 0).mean())
        if spread > 2:
            assert vis < 0.45, vis                              # the fallback is what ran
        pa.assert_stats(f"half-row route deg={deg} P={P} visible={vis:.2f}", pa.compare(hip, oo, og))


def test_binning_hand_overs_hold_under_load(gpu):
    """The counting kernel of the tile binning scans a chunk group's rows in the workgroup that completes the group
    (release / acquire inside one launch, tile_bin.hip), K3 finishes the scans of K1's workgroup sums from superblock
    totals that K1 adds up with atomics, and both lean on words that an earlier kernel of the SAME frame zeroed.  A stale
    row or a dirty counter is a wrong tile list, so: 1 500 frames of one scene back to back -- while a second stream keeps
    the memory system busy with uneven bursts -- every frame's tile ranges and sorted list equal to the oracle's."""
    import diff_gaussian_rasterization as dgr
    from oracle import raster_oracle as ro
    W, H, P = 1280, 720, 150_000
    cam = synth.make_camera(W, H)
    scene = synth.make_scene(P, cam, seed=21)
    geom = ro.geometry_spec(scene.means3D.numpy(), scene.scales.numpy(), scene.rotations.numpy(), None,
                            cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), W, H,
                            cam.tanfovx, cam.tanfovy, 1.0)
    binning = ro.binning_spec(geom)
    rs = dgr.GaussianRasterizationSettings(**pa.settings_kwargs(cam, torch.zeros(3), 3, device=gpu))
    sc = scene.to(gpu)
    ref_ranges = torch.from_numpy(binning.ranges.astype(np.int64)).to(gpu)
    ref_list = torch.from_numpy(binning.point_list.astype(np.int64)).to(gpu)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=gpu)
    bad = torch.zeros((), dtype=torch.int64, device=gpu)
    n_frames = 1500
    for it in range(n_frames):
        if it % 3 == 0:
            with torch.cuda.stream(side):                        # bursts of different lengths next to the frames
                for _ in range(1 + it % 4):
                    junk[: (8 + 8 * (it % 7)) << 20].add_(1)
        L, color, radii, geomb, binb, img, invd, call = dgr._C.rasterize_gaussians(
            rs.bg, sc.means3D, None, sc.opacities, sc.scales, sc.rotations, 1.0, None, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, H, W, sc.shs, 3, rs.campos, False, False, rs.render_indices, rs.parent_indices,
            rs.interpolation_weights, rs.num_node_kids, False)
        assert L == binning.num_rendered, (it, L)
        v = dgr._C.raster_views(call)
        bad += (v["ranges"].to(torch.int64) != ref_ranges).sum() + (v["point_list"].to(torch.int64) != ref_list).sum()
    torch.cuda.synchronize()
    assert int(bad.item()) == 0, f"{int(bad.item())} mismatching entries over {n_frames} frames"
