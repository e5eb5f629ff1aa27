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
        assert np.all(s_p[has_par] >= np.float32(tau))                 # parents of cut nodes were expanded
        inner = nodes[nn, 6] > 0
        assert np.all(s_n[inner] < np.float32(tau))                    # interior cut nodes are small enough
    assert prev == 1                                                   # huge tau -> root only
    r, _, nn = lo.expand_to_size(nodes, boxes, 0.0, vp)
    assert len(nn) == (nodes[:, 6] == 0).sum()                         # tau = 0 -> all leaves


def test_weight_continuity_at_switch():
    """The transition of a cut node runs while its parent's size falls from 2 tau to tau: t = 1 as long as
    size(parent) >= 2 tau, t -> 0 when tau reaches the parent's size (the node is about to collapse into it)."""
    h, _ = _hier(64)
    nodes, boxes = h.nodes.numpy(), h.boxes.numpy()
    vp = np.array([0.0, 0.0, -3.0], dtype=np.float32)
    n = int(np.nonzero((nodes[:, 6] > 0) & (nodes[:, 1] >= 0))[0][3])
    s_n = float(lo.node_size(boxes, np.array([n]), vp)[0])
    s_p = float(lo.node_size(boxes, np.array([nodes[n, 1]]), vp)[0])
    assert s_p > s_n
    w_hi, _ = lo.get_interpolation_weights([n], s_p, nodes, boxes, vp)           # tau = size(parent)
    w_lo, _ = lo.get_interpolation_weights([n], 0.5 * s_p, nodes, boxes, vp)     # size(parent) = 2 tau
    w_far, _ = lo.get_interpolation_weights([n], 0.1 * s_p, nodes, boxes, vp)
    assert abs(w_hi[0]) < 1e-6 and abs(w_lo[0] - 1) < 1e-6 and w_far[0] == 1.0
    taus = np.linspace(0.5 * s_p, s_p, 9)
    ws = [float(lo.get_interpolation_weights([n], t, nodes, boxes, vp)[0][0]) for t in taus]
    assert all(a >= b for a, b in zip(ws, ws[1:]))                               # monotone in between


def test_weight_known_answers():
    """Hand-computed values of t = max(1 - max(0, tau - s0) / (p - s0), 0), p = min(size(parent), 2 tau),
    s0 = max(p / 2, size(n)), on a two-level toy tree seen from far away (sizes = extent / distance)."""
    nodes = np.array([[0, -1, 0, 0, 1, 1, 3],
                      [1, 0, 1, 1, 0, 0, 0], [1, 0, 2, 1, 0, 0, 0], [1, 0, 3, 1, 0, 0, 0]], dtype=np.int32)
    boxes = np.zeros((4, 2, 4), dtype=np.float32)
    boxes[0, 0] = (-1, -1, -1, 12.0); boxes[0, 1, :3] = (1, 1, 1)              # root: extent 12
    for i, ext in enumerate((2.0, 7.0, 11.0)):                                   # children: extents 2, 7, 11
        boxes[1 + i, 0] = (-1, -1, -1, ext); boxes[1 + i, 1, :3] = (1, 1, 1)
    vp = np.array([0.0, 0.0, 11.0], dtype=np.float32)                            # distance to every box = 10
    # sizes: root 1.2, children 0.2, 0.7, 1.1
    w, kids = lo.get_interpolation_weights([0, 1, 2, 3], 1.0, nodes, boxes, vp)
    # tau = 1: p = min(1.2, 2) = 1.2; child 1: s0 = max(0.6, 0.2) = 0.6 -> 1 - 0.4 / 0.6; child 2: s0 = 0.7 -> 1 - 0.3 / 0.5
    # child 3: s0 = 1.1 -> tau - s0 < 0 -> 1
    assert kids.tolist() == [1, 3, 3, 3]
    assert np.allclose(w, [1.0, 1 - 0.4 / 0.6, 1 - 0.3 / 0.5, 1.0], atol=1e-6)
    w, _ = lo.get_interpolation_weights([1, 2], 0.5, nodes, boxes, vp)          # size(parent) > 2 tau -> p = 1.0
    # child 1: s0 = 0.5 -> tau - s0 = 0 -> 1; child 2: s0 = 0.7 -> 1
    assert np.allclose(w, [1.0, 1.0])
    w, _ = lo.get_interpolation_weights([1], 1.2, nodes, boxes, vp)             # tau = size(parent) -> 0
    assert abs(w[0]) < 1e-6
    # the cut at tau = 1: the root (1.2 >= 1) is too coarse, holds no leaves of its own; all three children are drawn
    r, p, nn = lo.expand_to_size(nodes, boxes, 1.0, vp)
    assert nn.tolist() == [1, 2, 3] and r.tolist() == [1, 2, 3] and p.tolist() == [0, 0, 0]
    r, p, nn = lo.expand_to_size(nodes, boxes, 1.3, vp)                          # root fine enough
    assert nn.tolist() == [0] and r.tolist() == [0] and p.tolist() == [0]


def test_nodes_with_own_leaves_and_several_merged():
    """The general node record: an interior node that holds leaf Gaussians itself keeps drawing them when it is
    expanded; a cut node draws leafs + merged."""
    nodes = np.array([[0, -1, 0, 2, 1, 1, 2],      # root: 2 own leaves (Gaussians 0, 1), 1 merged (2), children 1, 2
                      [1, 0, 3, 1, 0, 0, 0],       # leaf node: Gaussian 3
                      [1, 0, 4, 0, 2, 0, 0]],      # childless node with 2 merged Gaussians (4, 5)
                     dtype=np.int32)
    boxes = np.zeros((3, 2, 4), dtype=np.float32)
    for i, ext in enumerate((10.0, 1.0, 1.0)):
        boxes[i, 0] = (-1, -1, -1, ext); boxes[i, 1, :3] = (1, 1, 1)
    vp = np.array([0.0, 0.0, 11.0], dtype=np.float32)
    r, p, nn = lo.expand_to_size(nodes, boxes, 0.5, vp)      # root 1.0 >= 0.5: expanded; children 0.1 < 0.5: whole
    assert r.tolist() == [0, 1, 3, 4, 5] and nn.tolist() == [0, 0, 1, 2, 2] and p.tolist() == [0, 1, 0, 0, 0]
    r, p, nn = lo.expand_to_size(nodes, boxes, 2.0, vp)      # root fine enough: leaves + merged
    assert r.tolist() == [0, 1, 2] and nn.tolist() == [0, 0, 0]
    r, p, nn = lo.expand_to_size(nodes, boxes, 0.05, vp)     # everything too coarse: only leaf Gaussians remain
    assert r.tolist() == [0, 1, 3] and nn.tolist() == [0, 0, 1]


def test_two_chunk_merge_topology():
    """BASELINE config 3's 'merged 2-chunk toy hierarchy' (scripts/full_train.py:240-250): two chunk hierarchies under
    a common root."""
    cam = synth.make_camera(128, 96)
    a = hierarchy.build_hierarchy(synth.make_scene(150, cam, seed=1))
    sb = synth.make_scene(90, cam, seed=2)
    sb.means3D[:, 0] += 40.0                                   # the neighbouring chunk
    b = hierarchy.build_hierarchy(sb)
    m = hierarchy.merge_hierarchies([a, b])
    nodes, boxes = m.nodes.numpy(), m.boxes.numpy()
    N = nodes.shape[0]
    assert N == 1 + a.num_nodes + b.num_nodes and m.xyz.shape[0] == N
    assert nodes[0].tolist() == [0, -1, 0, 0, 1, 1, 2]
    assert nodes[1, 1] == 0 and nodes[2, 1] == 0
    for n in range(N):
        c0, cc = nodes[n, 5], nodes[n, 6]
        for c in range(c0, c0 + cc):
            assert nodes[c, 1] == n and nodes[c, 0] == nodes[n, 0] + 1
            assert np.all(boxes[c, 0, :3] >= boxes[n, 0, :3] - 1e-6) and np.all(boxes[c, 1, :3] <= boxes[n, 1, :3] + 1e-6)
        assert nodes[n, 2] == n and nodes[n, 3] + nodes[n, 4] == 1
    assert (nodes[:, 6] == 0).sum() == 240
    # the chunks' own Gaussians arrive unchanged (root of chunk b -> node 2, its first child -> first row after a's)
    assert np.array_equal(m.xyz[2].numpy(), b.xyz[0].numpy())
    assert np.array_equal(m.xyz[3:3 + a.num_nodes - 1].numpy(), a.xyz[1:].numpy())
    assert np.array_equal(m.shs[3 + a.num_nodes - 1:].numpy(), b.shs[1:].numpy())
    # a camera inside chunk a at a moderate threshold: chunk a is cut finer than the far chunk b
    vp = np.array([0.0, 0.0, 5.0], dtype=np.float32)
    r, p, nn = lo.expand_to_size(nodes, boxes, 0.05, vp)
    in_a = ((nn == 1) | ((nn >= 3) & (nn < 3 + a.num_nodes - 1))).sum()
    assert in_a > (len(nn) - in_a) and len(nn) - in_a >= 1
