"""VRAM-budgeted residency of a hierarchy's attribute rows (hgs/residency.py, csrc/residency.hip; BASELINE configs[4]
"VRAM-budgeted streaming LOD", the reference viewer's --budget, README.md:233-235).  The property: rendering through the
slot arrays gives the SAME BITS as rendering the fully resident hierarchy through the same in-op LOD path -- whatever
was fetched, evicted or already there -- and a view that does not fit is rendered at a coarser granularity."""
import numpy as np
import pytest
import torch

import parity as pa
from hgs import hierarchy, synth

pytestmark = pytest.mark.gpu
W, H = 320, 200


def _scene(gpu, leaves=20_000, seed=4):
    cam = synth.make_camera(W, H)
    h = hierarchy.build_hierarchy(synth.make_scene(leaves, cam, seed=seed))
    attrs = dict(means3D=h.xyz, shs=h.shs, opacities=h.alpha.abs().reshape(-1, 1), scales=torch.exp(h.log_scales),
                 rotations=torch.nn.functional.normalize(h.rots))
    return h, attrs, h.nodes.to(gpu), h.boxes.to(gpu)


def _render(gpu, cam, arrays, ri, pi, w, ns):
    import diff_gaussian_rasterization as dgr
    kw = pa.settings_kwargs(cam, torch.zeros(3), 3, do_depth=False, device=gpu, interpolation_weights=w, num_node_kids=ns)
    kw.update(render_indices=ri, parent_indices=pi)
    rs = dgr.GaussianRasterizationSettings(**kw)
    G = arrays["means3D"].shape[0]
    with torch.no_grad():
        color, radii, _ = dgr.GaussianRasterizer(rs)(means3D=arrays["means3D"], means2D=torch.zeros(G, 3, device=gpu),
                                                     shs=arrays["shs"], opacities=arrays["opacities"],
                                                     scales=arrays["scales"], rotations=arrays["rotations"])
    return color, radii


def _reference(gpu, cam, full, nodes, boxes, tau):
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
    G = full["means3D"].shape[0]
    ri = torch.zeros(G, dtype=torch.int32, device=gpu); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
    w = torch.zeros(G, device=gpu); ns = torch.zeros(G, dtype=torch.int32, device=gpu)
    n = expand_to_size(nodes, boxes, tau, cam.camera_center.to(gpu), torch.zeros(3), ri, pi, ni)
    get_interpolation_weights(ni[:n], tau, nodes, boxes, cam.camera_center.cpu(), torch.zeros(3), w, ns)
    color, radii = _render(gpu, cam, full, ri[:n], pi, w, ns)
    # rows the view needs: every node row, and the parent row of the entries whose weight is not exactly 1 (the in-op
    # LOD gather does not read the parent of a weight-1 entry, gaussian_math.h: lod_row_gather)
    rows = int(torch.unique(torch.cat([ri[:n], pi[:n][w[:n] != 1.0]])).numel())
    return color, radii, n, rows


def _views():
    return [(synth.orbit_camera(W, H, j, 6, radius=0.4, tilt=0.05), tau) for j, tau in
            enumerate((0.02, 0.004, 0.05, 0.01, 0.1, 0.004))]


def test_budgeted_rendering_is_bit_identical_and_recycles_slots(gpu):
    from hgs.residency import BudgetedHierarchy
    h, attrs, nodes, boxes = _scene(gpu)
    full = {k: v.to(gpu).contiguous() for k, v in attrs.items()}
    G = full["means3D"].shape[0]
    refs = [_reference(gpu, cam, full, nodes, boxes, tau) for cam, tau in _views()]
    need = max(r[3] for r in refs)
    assert need < 0.9 * G                                     # the views leave room below the full hierarchy
    bh = BudgetedHierarchy(attrs["means3D"], attrs["shs"], attrs["opacities"], attrs["scales"], attrs["rotations"], gpu,
                           budget_rows=int(need * 1.05))     # every view fits, the union of all views does not
    assert bh.B < G
    for rnd in range(2):
        for (cam, tau), (color_ref, radii_ref, n_ref, rows_ref) in zip(_views(), refs):
            sel = bh.select(nodes, boxes, tau, cam.camera_center.to(gpu), cam.camera_center.cpu())
            assert sel.attempts == 1 and sel.tau == tau and sel.n == n_ref
            assert int(sel.render_indices.min()) >= 0 and int(sel.parent_indices.min()) >= 0
            assert int(sel.render_indices.max()) < bh.B
            arrays = dict(means3D=bh.means3D, shs=bh.shs, opacities=bh.opacities, scales=bh.scales, rotations=bh.rotations)
            color, radii = _render(gpu, cam, arrays, sel.render_indices, sel.parent_indices, sel.weights, sel.kids)
            assert torch.equal(color, color_ref) and torch.equal(radii, radii_ref)
            assert bh.resident_rows <= bh.B
    st = bh.stats
    print(st, "budget rows", bh.B, "of", G)
    assert st["evictions"] > 0 and st["retries"] == 0
    assert st["rows_fetched"] > bh.B                           # more rows went through the slots than there are slots
    # bookkeeping is consistent: every occupied slot is the slot of its row, the free list holds the rest
    ids = bh.id_of_slot.long()
    occ = ids >= 0
    assert int(occ.sum()) == bh.resident_rows
    assert torch.equal(bh.slot_of[ids[occ]].long(), torch.nonzero(occ).reshape(-1))
    assert int((bh.slot_of >= 0).sum()) == bh.resident_rows and int((bh.slot_of == -2).sum()) == 0
    free = bh.free_list[:bh.free_top].long()
    assert free.unique().numel() == bh.free_top and not bool(occ[free].any())


def test_view_that_does_not_fit_is_rendered_coarser(gpu):
    """The reference viewer "auto-regulates and raises the granularity until the scene can fit inside the defined VRAM
    budget" (README.md:235): tau = 0 asks for every leaf; with a quarter of the rows as budget the cut is repeated at
    1.2 x tau until it fits, and what is rendered equals the fully resident render at THAT tau."""
    from hgs.residency import BudgetedHierarchy
    h, attrs, nodes, boxes = _scene(gpu, leaves=8_000, seed=6)
    full = {k: v.to(gpu).contiguous() for k, v in attrs.items()}
    G = full["means3D"].shape[0]
    cam = synth.make_camera(W, H)
    bh = BudgetedHierarchy(attrs["means3D"], attrs["shs"], attrs["opacities"], attrs["scales"], attrs["rotations"], gpu,
                           budget_mb=(G // 4) * 4 * (3 * 16 + 11) / 1e6)
    assert bh.B == G // 4
    sel = bh.select(nodes, boxes, 0.0, cam.camera_center.to(gpu), cam.camera_center.cpu())
    assert sel.attempts > 1 and sel.tau > 0.0 and bh.stats["retries"] == sel.attempts - 1
    color_ref, radii_ref, n_ref, rows_ref = _reference(gpu, cam, full, nodes, boxes, sel.tau)
    assert sel.n == n_ref and rows_ref <= bh.B
    arrays = dict(means3D=bh.means3D, shs=bh.shs, opacities=bh.opacities, scales=bh.scales, rotations=bh.rotations)
    color, radii = _render(gpu, cam, arrays, sel.render_indices, sel.parent_indices, sel.weights, sel.kids)
    assert torch.equal(color, color_ref) and torch.equal(radii, radii_ref)
    assert int((bh.slot_of == -2).sum()) == 0                  # nothing is left queued by the attempts that failed
    # a second, finer request after the coarse one: the rows of the failed attempts did not leak slots
    sel2 = bh.select(nodes, boxes, sel.tau, cam.camera_center.to(gpu), cam.camera_center.cpu())
    assert sel2.attempts == 1 and sel2.misses == 0


def test_index_outside_the_hierarchy_is_refused(gpu):
    from hgs.residency import BudgetedHierarchy
    h, attrs, nodes, boxes = _scene(gpu, leaves=500, seed=1)
    bh = BudgetedHierarchy(attrs["means3D"], attrs["shs"], attrs["opacities"], attrs["scales"], attrs["rotations"], gpu,
                           budget_rows=100)
    bad = torch.tensor([0, 1, bh.G], dtype=torch.int32, device=gpu)
    with pytest.raises(RuntimeError):
        bh.make_resident(bad, bad)


def test_hier_file_to_budgeted_render(gpu, tmp_path):
    """The viewer's chain: a .hier file on disk -> pinned host arrays -> a budget of rows on the GPU -> the same bits as
    the resident render of the loaded hierarchy (activations as scene/gaussian_model.py applies them to a hierarchy)."""
    from gaussian_hierarchy._C import write_hierarchy
    from hgs.residency import BudgetedHierarchy
    h, attrs, nodes, boxes = _scene(gpu, leaves=3_000, seed=9)
    path = str(tmp_path / "scene.hier")
    write_hierarchy(path, h.xyz, h.shs, h.alpha, h.log_scales, h.rots, h.nodes, h.boxes)
    bh, nodes_f, boxes_f = BudgetedHierarchy.from_hier_file(path, gpu, budget_rows=int(0.8 * h.xyz.shape[0]))
    assert torch.equal(nodes_f, nodes) and torch.equal(boxes_f, boxes)
    full = {k: v.to(gpu).contiguous() for k, v in attrs.items()}
    cam, tau = synth.make_camera(W, H), 0.02
    sel = bh.select(nodes_f, boxes_f, tau, cam.camera_center.to(gpu), cam.camera_center.cpu())
    color_ref, radii_ref, n_ref, rows_ref = _reference(gpu, cam, full, nodes, boxes, sel.tau)   # (tau may have been raised)
    assert sel.n == n_ref and sel.misses == rows_ref <= bh.B
    arrays = dict(means3D=bh.means3D, shs=bh.shs, opacities=bh.opacities, scales=bh.scales, rotations=bh.rotations)
    color, radii = _render(gpu, cam, arrays, sel.render_indices, sel.parent_indices, sel.weights, sel.kids)
    assert torch.equal(color, color_ref) and torch.equal(radii, radii_ref)


def test_fly_through_of_a_million_node_hierarchy_streams_bit_identically(gpu):
    """streaming while something streams.  A 1.2 M-node hierarchy, a camera flying forward through
    it (every frame's cut differs from the last; one jump back to the start), a budget of 1.15 x the largest single
    view: rows are fetched on every frame and slots are recycled continuously -- and every frame equals the fully
    resident render of the same cut bit for bit."""
    from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
    from hgs.residency import BudgetedHierarchy
    cam0 = synth.make_camera(W, H)
    h = hierarchy.build_hierarchy_on_device(600_000, cam0, gpu, seed=11)
    G = h.xyz.shape[0]
    assert G >= 1_000_000
    full = dict(means3D=h.xyz, shs=h.shs, opacities=h.alpha.abs().reshape(-1, 1).contiguous(),
                scales=torch.exp(h.log_scales), rotations=torch.nn.functional.normalize(h.rots))
    nodes, boxes = h.nodes, h.boxes
    tau = (2 * 8.0 + 1) * cam0.tanfovx / (0.5 * W)                     # 8 px: a cut in the middle of the tree (~270-400 k rows)
    # forward 0.25 units per frame (~7 % of the cut changes per frame), after 16 frames a jump 3 units sideways and back to
    # the start depth, then forward again
    cams = [synth.make_camera(W, H, T=np.array([0.0 if k < 16 else -3.0, 0.0, -0.25 * (k % 16)])) for k in range(24)]
    refs = [_reference(gpu, c, full, nodes, boxes, tau) for c in cams]
    need = max(r[3] for r in refs)
    bh = BudgetedHierarchy(full["means3D"].cpu(), full["shs"].cpu(), full["opacities"].cpu(), full["scales"].cpu(),
                           full["rotations"].cpu(), gpu, budget_rows=int(1.15 * need))
    assert bh.B < 0.8 * G
    fetched = []
    for c, (color_ref, radii_ref, n_ref, rows_ref) in zip(cams, refs):
        sel = bh.select(nodes, boxes, tau, c.camera_center.to(gpu), c.camera_center.cpu())
        assert sel.attempts == 1 and sel.n == n_ref
        arrays = dict(means3D=bh.means3D, shs=bh.shs, opacities=bh.opacities, scales=bh.scales, rotations=bh.rotations)
        color, radii = _render(gpu, c, arrays, sel.render_indices, sel.parent_indices, sel.weights, sel.kids)
        assert torch.equal(color, color_ref) and torch.equal(radii, radii_ref)
        fetched.append(sel.misses)
    print("rows fetched per frame:", fetched, "cut sizes", [r[2] for r in refs], "budget", bh.B, "evictions",
          bh.stats["evictions"])
    assert all(m >= 0.01 * r[2] for m, r in zip(fetched, refs))    # >= 1 % of the cut streams in on EVERY frame
    assert fetched[16] > 1.5 * fetched[15]                         # the jump is a burst
    assert sum(fetched) > 1.2 * bh.B and bh.stats["evictions"] > 0.4 * bh.B    # slots are recycled continuously
    assert bh.stats["retries"] == 0
    assert int((bh.slot_of == -2).sum()) == 0


def test_fly_through_with_prefetch_of_the_next_view(gpu):
    """The same kind of fly-through with ``prefetch`` of the NEXT view on a second stream after every render was enqueued
    (hgs/residency.py, round 6): every frame -- the one after the jump included -- still equals the fully resident render
    bit for bit, the rows now cross PCIe under the previous frame (``select`` finds the prefetched cut resident: it fetches
    nothing), rows of the frame being rendered are never evicted by the prefetch, and a view that was NOT the prefetched
    one (the pose prediction missed) is rendered correctly all the same."""
    from hgs.residency import BudgetedHierarchy
    cam0 = synth.make_camera(W, H)
    h = hierarchy.build_hierarchy_on_device(300_000, cam0, gpu, seed=13)
    G = h.xyz.shape[0]
    full = dict(means3D=h.xyz, shs=h.shs, opacities=h.alpha.abs().reshape(-1, 1).contiguous(),
                scales=torch.exp(h.log_scales), rotations=torch.nn.functional.normalize(h.rots))
    nodes, boxes = h.nodes, h.boxes
    tau = (2 * 8.0 + 1) * cam0.tanfovx / (0.5 * W)
    cams = [synth.make_camera(W, H, T=np.array([0.0 if k < 10 else -3.0, 0.0, -0.25 * (k % 10)])) for k in range(18)]
    refs = [_reference(gpu, c, full, nodes, boxes, tau) for c in cams]
    need = max(r[3] for r in refs)
    bh = BudgetedHierarchy(full["means3D"].cpu(), full["shs"].cpu(), full["opacities"].cpu(), full["scales"].cpu(),
                           full["rotations"].cpu(), gpu, budget_rows=int(1.1 * need))
    vps = [(c.camera_center.to(gpu), c.camera_center.cpu()) for c in cams]
    fetched_in_select, mispredicted = [], 13
    for k, (c, (color_ref, radii_ref, n_ref, rows_ref)) in enumerate(zip(cams, refs)):
        sel = bh.select(nodes, boxes, tau, *vps[k])
        assert sel.attempts == 1 and sel.n == n_ref
        arrays = dict(means3D=bh.means3D, shs=bh.shs, opacities=bh.opacities, scales=bh.scales, rotations=bh.rotations)
        color, radii = _render(gpu, c, arrays, sel.render_indices, sel.parent_indices, sel.weights, sel.kids)
        if k + 1 < len(cams):          # (frame `mispredicted` is prefetched for the wrong pose)
            nxt = vps[k + 1] if k + 1 != mispredicted else vps[0]
            bh.prefetch(nodes, boxes, tau, *nxt)
        assert torch.equal(color, color_ref) and torch.equal(radii, radii_ref), k
        fetched_in_select.append(sel.misses)
    torch.cuda.synchronize()
    print("rows fetched inside select per frame:", fetched_in_select, "prefetched", bh.stats.get("prefetched_rows"))
    # the budget is 1.1 x the largest view: the prefetch may only take free slots and slots of OLDER frames, so select
    # still fetches what did not fit -- but most rows cross the bus behind a render
    later = sum(m for i, m in enumerate(fetched_in_select[1:], 1) if i != mispredicted)
    assert fetched_in_select[0] > 0 and fetched_in_select[mispredicted] > 0
    assert bh.stats["prefetched_rows"] > 2 * later, (bh.stats["prefetched_rows"], later)
    assert bh.stats["evictions"] > 0 and bh.stats["retries"] == 0 and int((bh.slot_of == -2).sum()) == 0
