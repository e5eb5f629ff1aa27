"""LOD-cut oracle + synthetic hierarchy generator: structural invariants (CPU)."""
import numpy as np

from hgs import hierarchy, synth
from oracle import lod_oracle as lo


def _hier(P=500, seed=2):
    cam = synth.make_camera(128, 96)
    return hierarchy.build_hierarchy(synth.make_scene(P, cam, seed=seed)), cam


def test_hierarchy_topology():
    h, _ = _hier(333)
    nodes, boxes = h.nodes.numpy(), h.boxes.numpy()
    N = nodes.shape[0]
    assert N == 2 * 333 - 1 and nodes[0, 1] == -1 and nodes[0, 0] == 0
    for n in range(N):
        c0, cc = nodes[n, 5], nodes[n, 6]
        for c in range(c0, c0 + cc):
            assert nodes[c, 1] == n and nodes[c, 0] == nodes[n, 0] + 1
            assert np.all(boxes[c, 0, :3] >= boxes[n, 0, :3] - 1e-6) and np.all(boxes[c, 1, :3] <= boxes[n, 1, :3] + 1e-6)
        assert nodes[n, 3] + nodes[n, 4] == 1
    assert (nodes[:, 6] == 0).sum() == 333


def test_cut_is_a_frontier_and_monotone_in_tau():
    h, cam = _hier(700)
    nodes, boxes = h.nodes.numpy(), h.boxes.numpy()
    vp = np.array([0.1, -0.2, -1.0], dtype=np.float32)
    prev = None
    for tau in (1e-4, 0.003, 0.01, 0.05, 0.3, 1e3):
        r, p, nn = lo.expand_to_size(nodes, boxes, tau, vp)
        assert np.all(np.diff(nn) > 0)                               # canonical order: ascending node index
        # every leaf has exactly one ancestor-or-self in the cut
        sel = np.zeros(nodes.shape[0], dtype=bool); sel[nn] = True
        for leaf in np.nonzero(nodes[:, 6] == 0)[0][::7]:
            cnt, n = 0, leaf
            while n >= 0:
                cnt += sel[n]; n = nodes[n, 1]
            assert cnt == 1
        assert np.array_equal(r, nodes[nn, 2])
        par = nodes[nn, 1]
        assert np.array_equal(p, np.where(par >= 0, nodes[np.maximum(par, 0), 2], r))
        if prev is not None:
            assert len(nn) <= prev                                     # coarser threshold -> fewer nodes
        prev = len(nn)
        w, kids = lo.get_interpolation_weights(nn, tau, nodes, boxes, vp)
        assert np.all((w >= 0) & (w <= 1)) and np.all(kids >= 1)
        s_n = lo.node_size(boxes, nn, vp)
        has_par = par >= 0
        s_p = lo.node_size(boxes, np.maximum(par, 0), vp)
        assert np.all(s_p[has_par] > np.float32(tau))                  # parents of cut nodes were expanded
        inner = nodes[nn, 6] > 0
        assert np.all(s_n[inner] <= np.float32(tau))                   # interior cut nodes are small enough
    assert prev == 1                                                   # huge tau -> root only
    r, _, nn = lo.expand_to_size(nodes, boxes, 0.0, vp)
    assert len(nn) == (nodes[:, 6] == 0).sum()                         # tau = 0 -> all leaves


def test_weight_continuity_at_switch():
    """w -> 1 when tau reaches the node's own size, w -> 0 when tau reaches the parent's size."""
    h, _ = _hier(64)
    nodes, boxes = h.nodes.numpy(), h.boxes.numpy()
    vp = np.array([0.0, 0.0, -3.0], dtype=np.float32)
    n = int(np.nonzero((nodes[:, 6] > 0) & (nodes[:, 1] >= 0))[0][3])
    s_n = float(lo.node_size(boxes, np.array([n]), vp)[0])
    s_p = float(lo.node_size(boxes, np.array([nodes[n, 1]]), vp)[0])
    if s_p > s_n:
        w_lo, _ = lo.get_interpolation_weights([n], s_n, nodes, boxes, vp)
        w_hi, _ = lo.get_interpolation_weights([n], s_p, nodes, boxes, vp)
        assert abs(w_lo[0] - 1) < 1e-6 and abs(w_hi[0]) < 1e-6
