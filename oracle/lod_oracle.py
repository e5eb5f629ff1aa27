"""CPU oracle for the hierarchy LOD cut (``expand_to_size`` / ``get_interpolation_weights``).

TEST INFRASTRUCTURE ONLY (see oracle/raster_oracle.py header).

PARITY UNPINNED: the reference implementation lives in the un-vendored
submodule ``graphdeco-inria/gaussian-hierarchy`` (/root/reference/.gitmodules:10-12,
empty directory, SHA unknown); the reference has no tests for it.  What is
restated here is fixed by the call sites and by the consumer's semantics:

* train_post.py:91-113, render_hierarchy.py:58-80 -- argument order, which
  arrays are filled, that the return value counts the filled entries;
* gaussian_renderer/__init__.py:204-218 -- ``w = 1`` renders the node's own
  attributes, ``w = 0`` renders the parent's; ``parent_indices`` index the
  same Gaussian arrays as ``render_indices``;
* render_hierarchy.py:55-56 -- the threshold is a tangent-space angular size
  (``(2 tau + 1)`` pixels).

Data model (documented in DESIGN.md, section 4; the node record and the box layout are those of the public
gaussian-hierarchy repository, which the reference checkout does not vendor):
  nodes  int32 [N,7]  = (depth, parent, start, count_leafs, count_merged,
                         start_children, count_children); root is node 0 with
                         parent -1; children of a node are contiguous.  A node's Gaussians are
                         [start, start + count_leafs) leaves it holds itself, then count_merged merged ones.
  boxes  f32 [N,2,4]  = [n,0,:3] AABB min, [n,1,:3] AABB max,
                        [n,0,3] node extent (world units), [n,1,3] unused.
  size(n, v) = extent(n) / dist(v, AABB(n)); FLT_MAX when v is inside.
Cut: top-down from the root.  A reached node with size >= tau is too coarse: its count_leafs own Gaussians are
drawn and its children are reached; a reached node with size < tau is drawn as a whole (leafs + merged).  With
sizes that shrink from parent to child this equals the per-node test "size < tau <= size(parent)" of the upstream
tools.  Output is ordered by ascending node index (then by Gaussian offset inside the node).
Weight t of a cut node (1 = the node itself, 0 = looks like its parent): 1 at the root; else with
p = min(size(parent), 2 tau) and s0 = max(p / 2, size(n)): t = 1 if p <= s0 else max(1 - max(0, tau - s0)/(p - s0), 0)
-- the transition runs while the parent's size falls from 2 tau to tau.
"""
from __future__ import annotations

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)
F = np.float32


def node_size(boxes, n, viewpoint):
    """float32, fixed operation order (the HIP kernel follows it, contraction off)."""
    v = np.asarray(viewpoint, dtype=np.float32)
    mn = boxes[n, 0, :3]
    mx = boxes[n, 1, :3]
    d = np.maximum(np.maximum(mn - v, v - mx), F(0.0))       # per-axis distance outside the box
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    dist = np.sqrt(d2)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = boxes[n, 0, 3] / dist
    return np.where(d2 > F(0.0), s, FLT_MAX).astype(np.float32)


def expand_to_size(nodes, boxes, size, viewpoint, viewdir=None):
    """Returns (render_indices, parent_indices, nodes_for_render_indices) int32 arrays."""
    nodes = np.asarray(nodes)
    boxes = np.asarray(boxes, dtype=np.float32)
    tau = F(size)
    N = nodes.shape[0]
    count = np.zeros(N, dtype=np.int64)
    frontier = np.array([0], dtype=np.int64) if N else np.zeros(0, dtype=np.int64)
    while frontier.size:
        s = node_size(boxes, frontier, viewpoint)
        nchild = nodes[frontier, 6]
        coarse = s >= tau
        count[frontier[coarse]] = nodes[frontier[coarse], 3]
        count[frontier[~coarse]] = nodes[frontier[~coarse], 3] + nodes[frontier[~coarse], 4]
        ex = frontier[coarse & (nchild > 0)]
        if ex.size == 0:
            break
        starts = nodes[ex, 5].astype(np.int64)
        cnts = nodes[ex, 6].astype(np.int64)
        frontier = np.concatenate([np.arange(a, a + c) for a, c in zip(starts, cnts)])
    sel = np.nonzero(count)[0]
    r, p, nn = [], [], []
    for n in sel:
        start = int(nodes[n, 2])
        cnt = int(count[n])
        par = int(nodes[n, 1])
        pg = int(nodes[par, 2]) if par >= 0 else -1
        for k in range(cnt):
            r.append(start + k)
            p.append(pg if pg >= 0 else start + k)
            nn.append(n)
    return (np.asarray(r, dtype=np.int32), np.asarray(p, dtype=np.int32), np.asarray(nn, dtype=np.int32))


def get_interpolation_weights(node_indices, size, nodes, boxes, viewpoint, viewdir=None):
    """Returns (weights f32 [n], num_siblings int32 [n]) for the given node indices."""
    nodes = np.asarray(nodes)
    boxes = np.asarray(boxes, dtype=np.float32)
    tau = F(size)
    ni = np.asarray(node_indices, dtype=np.int64)
    par = nodes[ni, 1].astype(np.int64)
    has_par = par >= 0
    ps = np.where(has_par, par, 0)
    two_tau = F(2.0) * tau
    sp = np.minimum(node_size(boxes, ps, viewpoint), two_tau)
    sn = node_size(boxes, ni, viewpoint)
    start = np.maximum(F(0.5) * sp, sn)
    diff = sp - start
    tdiff = np.maximum(F(0.0), tau - start)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.maximum(F(1.0) - tdiff / diff, F(0.0))
    w = np.where(has_par & (diff > F(0.0)), t, F(1.0)).astype(np.float32)
    kids = np.where(has_par, nodes[ps, 6], 1).astype(np.int32)
    return w, kids
