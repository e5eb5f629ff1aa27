"""Synthetic Gaussian hierarchies for the LOD-cut path (BASELINE.json configs 3 and 5).

The reference builds hierarchies offline with its GaussianHierarchyCreator / Merger tools
(scripts/full_train.py:138-139,188-196,242-250 -- C++ sources absent, out of scope).  This
module only produces *inputs* for ``expand_to_size`` / ``get_interpolation_weights`` /
``render_post``-style rendering: a balanced binary BVH over Morton-sorted leaves, interior
nodes holding a moment-matched merge of their children.  Layout = DESIGN.md '.hier layout':

  one Gaussian per node, Gaussian index == node index (``start`` = node id)
  nodes int32 [N,7] = depth, parent, start, count_leafs, count_merged, start_children, count_children
  boxes f32 [N,2,4]  = AABB min + extent (max edge), AABB max + 0
Children of a node are contiguous (BFS numbering).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class Hierarchy:
    xyz: torch.Tensor         # [G,3]
    shs: torch.Tensor         # [G,16,3]
    alpha: torch.Tensor       # [G,1] activated opacity
    log_scales: torch.Tensor  # [G,3]
    rots: torch.Tensor        # [G,4]
    nodes: torch.Tensor       # [N,7] int32
    boxes: torch.Tensor       # [N,2,4] float32

    @property
    def num_nodes(self):
        return self.nodes.shape[0]


def _morton(xyz: np.ndarray) -> np.ndarray:
    lo, hi = xyz.min(0), xyz.max(0)
    q = np.clip(((xyz - lo) / np.maximum(hi - lo, 1e-12) * 1023.0), 0, 1023).astype(np.uint64)

    def spread(v):
        v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))


def _rot_from_quat(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)


def _quat_from_rot(R):
    m00, m01, m02 = R[:, 0, 0], R[:, 0, 1], R[:, 0, 2]
    m10, m11, m12 = R[:, 1, 0], R[:, 1, 1], R[:, 1, 2]
    m20, m21, m22 = R[:, 2, 0], R[:, 2, 1], R[:, 2, 2]
    q = np.empty((R.shape[0], 4))
    tr = m00 + m11 + m22
    c0 = tr > 0
    c1 = (~c0) & (m00 >= m11) & (m00 >= m22)
    c2 = (~c0) & (~c1) & (m11 >= m22)
    c3 = ~(c0 | c1 | c2)
    with np.errstate(invalid="ignore"):
        s = np.sqrt(np.maximum(tr + 1.0, 1e-20)) * 2
        q[c0] = np.stack([0.25 * s, (m21 - m12) / s, (m02 - m20) / s, (m10 - m01) / s], 1)[c0]
        s = np.sqrt(np.maximum(1.0 + m00 - m11 - m22, 1e-20)) * 2
        q[c1] = np.stack([(m21 - m12) / s, 0.25 * s, (m01 + m10) / s, (m02 + m20) / s], 1)[c1]
        s = np.sqrt(np.maximum(1.0 + m11 - m00 - m22, 1e-20)) * 2
        q[c2] = np.stack([(m02 - m20) / s, (m01 + m10) / s, 0.25 * s, (m12 + m21) / s], 1)[c2]
        s = np.sqrt(np.maximum(1.0 + m22 - m00 - m11, 1e-20)) * 2
        q[c3] = np.stack([(m10 - m01) / s, (m02 + m20) / s, (m12 + m21) / s, 0.25 * s], 1)[c3]
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def build_hierarchy(scene) -> Hierarchy:
    """scene: hgs.synth.Scene (activated scales / opacities, SH degree-3 storage)."""
    xyz = scene.means3D.double().numpy()
    P = xyz.shape[0]
    assert P >= 1
    order = np.argsort(_morton(xyz), kind="stable")
    # ---- topology: BFS over index ranges of the Morton-sorted leaves ------------------
    lo, hi, depth, parent = [np.array([0])], [np.array([P])], [np.array([0])], [np.array([-1])]
    start_children, count_children = [], []
    first_id = [0]
    next_id = 1
    while True:
        l, h = lo[-1], hi[-1]
        interior = (h - l) > 1
        n_int = int(interior.sum())
        sc = np.zeros(l.shape[0], dtype=np.int64)
        cc = np.where(interior, 2, 0)
        sc[interior] = next_id + 2 * np.arange(n_int)
        start_children.append(sc)
        count_children.append(cc)
        if n_int == 0:
            break
        mid = (l[interior] + h[interior]) // 2
        ids = first_id[-1] + np.nonzero(interior)[0]
        lo.append(np.stack([l[interior], mid], 1).reshape(-1))
        hi.append(np.stack([mid, h[interior]], 1).reshape(-1))
        depth.append(np.full(2 * n_int, len(lo) - 1))
        parent.append(np.repeat(ids, 2))
        first_id.append(next_id)
        next_id += 2 * n_int
    N = next_id
    lo_a, hi_a = np.concatenate(lo), np.concatenate(hi)
    depth_a, parent_a = np.concatenate(depth), np.concatenate(parent)
    sc_a, cc_a = np.concatenate(start_children), np.concatenate(count_children)
    is_leaf = cc_a == 0

    # ---- attributes ---------------------------------------------------------------------
    mu = np.zeros((N, 3)); cov = np.zeros((N, 3, 3)); w = np.zeros(N)
    sh = np.zeros((N, 16, 3)); op = np.zeros(N)
    bmin = np.zeros((N, 3)); bmax = np.zeros((N, 3))
    src = order[lo_a[is_leaf]]
    s_leaf = scene.scales.double().numpy()[src]
    R_leaf = _rot_from_quat(scene.rotations.double().numpy()[src])
    Lm = R_leaf * s_leaf[:, None, :]
    mu[is_leaf] = xyz[src]
    cov[is_leaf] = Lm @ Lm.transpose(0, 2, 1)
    op[is_leaf] = scene.opacities.double().numpy().reshape(-1)[src]
    w[is_leaf] = op[is_leaf] * np.prod(s_leaf, axis=1)
    M = scene.shs.shape[1]
    sh[is_leaf, :M] = scene.shs.double().numpy()[src]
    ext = 3.0 * s_leaf.max(axis=1, keepdims=True)
    bmin[is_leaf] = xyz[src] - ext
    bmax[is_leaf] = xyz[src] + ext
    for lvl in range(len(lo) - 1, -1, -1):                     # bottom-up merge
        a, b = first_id[lvl], first_id[lvl] + lo[lvl].shape[0]
        ids = np.arange(a, b)[~is_leaf[a:b]]
        if ids.size == 0:
            continue
        c0, c1 = sc_a[ids], sc_a[ids] + 1
        ws = np.maximum(w[c0] + w[c1], 1e-30)
        f0, f1 = (w[c0] / ws)[:, None], (w[c1] / ws)[:, None]
        m = f0 * mu[c0] + f1 * mu[c1]
        d0, d1 = mu[c0] - m, mu[c1] - m
        cov[ids] = f0[:, :, None] * (cov[c0] + d0[:, :, None] * d0[:, None, :]) + \
            f1[:, :, None] * (cov[c1] + d1[:, :, None] * d1[:, None, :])
        mu[ids] = m
        sh[ids] = f0[:, :, None] * sh[c0] + f1[:, :, None] * sh[c1]
        op[ids] = np.clip(f0[:, 0] * op[c0] + f1[:, 0] * op[c1], 0.0, 1.0)
        w[ids] = ws
        bmin[ids] = np.minimum(bmin[c0], bmin[c1])
        bmax[ids] = np.maximum(bmax[c0], bmax[c1])
    evals, evecs = np.linalg.eigh(cov)
    evals = np.maximum(evals, 1e-12)
    flip = np.linalg.det(evecs) < 0
    evecs[flip, :, 0] *= -1
    quat = _quat_from_rot(evecs)
    scales = np.sqrt(evals)
    # leaves keep their exact input parametrisation
    quat[is_leaf] = scene.rotations.double().numpy()[src]
    scales[is_leaf] = s_leaf

    nodes = np.stack([depth_a, parent_a, np.arange(N), is_leaf.astype(np.int64), (~is_leaf).astype(np.int64),
                      np.where(is_leaf, 0, sc_a), cc_a], 1).astype(np.int32)
    boxes = np.zeros((N, 2, 4), dtype=np.float32)
    boxes[:, 0, :3] = bmin
    boxes[:, 1, :3] = bmax
    boxes[:, 0, 3] = (boxes[:, 1, :3] - boxes[:, 0, :3]).max(axis=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return Hierarchy(xyz=t(mu), shs=t(sh), alpha=t(op[:, None]), log_scales=t(np.log(scales)), rots=t(quat),
                     nodes=torch.from_numpy(nodes), boxes=torch.from_numpy(boxes))


def build_hierarchy_on_device(P, cam, device, seed=0, sh_degree=3, s_px=(0.5, 4.0), z_range=(2.0, 20.0)) -> Hierarchy:
    """Scale-test generator (BASELINE config 5: tens of millions of nodes): same topology and layout as
    ``build_hierarchy`` over the same leaf distribution as ``hgs.synth.make_scene``, but built with torch ops on
    ``device`` in float32, and interior nodes are AXIS-ALIGNED moment matches (scales = sqrt of the merged covariance's
    diagonal, identity rotation) instead of eigen-decomposed ones.  2 P - 1 nodes; everything stays on ``device``."""
    import math
    g = torch.Generator(device=device).manual_seed(seed)
    U = lambda *s: torch.rand(*s, generator=g, device=device)
    N_ = lambda *s: torch.randn(*s, generator=g, device=device)
    z = z_range[0] + (z_range[1] - z_range[0]) * U(P)
    xyz = torch.stack([z * cam.tanfovx * (2 * U(P) - 1), z * cam.tanfovy * (2 * U(P) - 1), z], 1)
    fx = cam.image_width / (2.0 * cam.tanfovx)
    spx = torch.exp(math.log(s_px[0]) + (math.log(s_px[1]) - math.log(s_px[0])) * U(P))
    s_leaf = (z * spx / fx)[:, None] * (0.3 + 0.7 * U(P, 3))
    q_leaf = torch.nn.functional.normalize(N_(P, 4), dim=1)
    o_leaf = 0.05 + 0.9 * U(P)
    M = (sh_degree + 1) ** 2
    # Morton order of the leaves
    lo_, hi_ = xyz.min(0).values, xyz.max(0).values
    qi = ((xyz - lo_) / (hi_ - lo_).clamp_min(1e-12) * 1023.0).clamp(0, 1023).long()

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    order = torch.argsort(spread(qi[:, 0]) | (spread(qi[:, 1]) << 1) | (spread(qi[:, 2]) << 2), stable=True)
    del qi
    # ---- topology (BFS over index ranges; children contiguous) ------------------------------------------
    lo, hi = [torch.zeros(1, dtype=torch.int64, device=device)], [torch.full((1,), P, dtype=torch.int64, device=device)]
    parent = [torch.full((1,), -1, dtype=torch.int64, device=device)]
    start_children, first_id, next_id = [], [0], 1
    while True:
        l, h = lo[-1], hi[-1]
        interior = (h - l) > 1
        n_int = int(interior.sum())
        sc = torch.zeros_like(l)
        sc[interior] = next_id + 2 * torch.arange(n_int, device=device)
        start_children.append(sc)
        if n_int == 0:
            break
        mid = (l[interior] + h[interior]) // 2
        ids = first_id[-1] + interior.nonzero().flatten()
        lo.append(torch.stack([l[interior], mid], 1).reshape(-1))
        hi.append(torch.stack([mid, h[interior]], 1).reshape(-1))
        parent.append(ids.repeat_interleave(2))
        first_id.append(next_id)
        next_id += 2 * n_int
    N = next_id
    lo_a = torch.cat(lo)
    sc_a = torch.cat(start_children)
    depth_a = torch.cat([torch.full((t.shape[0],), d, dtype=torch.int64, device=device) for d, t in enumerate(lo)])
    parent_a = torch.cat(parent)
    is_leaf = sc_a == 0
    is_leaf[0] = P == 1
    # ---- attributes -----------------------------------------------------------------------------------------
    f32 = dict(dtype=torch.float32, device=device)
    mu = torch.zeros(N, 3, **f32); var = torch.zeros(N, 3, **f32); w = torch.zeros(N, **f32)
    op = torch.zeros(N, **f32); sh = torch.zeros(N, 16, 3, **f32)
    bmin = torch.zeros(N, 3, **f32); bmax = torch.zeros(N, 3, **f32)
    rots = torch.zeros(N, 4, **f32); rots[:, 0] = 1.0
    leaf_ids = is_leaf.nonzero().flatten()
    src = order[lo_a[leaf_ids]]
    mu[leaf_ids] = xyz[src]
    # axis-aligned second moments of an oriented leaf: diag(R diag(s^2) R^T)
    r, x, y, zq = q_leaf[src].unbind(1)
    R = torch.stack([1 - 2 * (y * y + zq * zq), 2 * (x * y - r * zq), 2 * (x * zq + r * y),
                     2 * (x * y + r * zq), 1 - 2 * (x * x + zq * zq), 2 * (y * zq - r * x),
                     2 * (x * zq - r * y), 2 * (y * zq + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    var[leaf_ids] = ((R * R) * (s_leaf[src] ** 2)[:, None, :]).sum(2)
    del R
    op[leaf_ids] = o_leaf[src]
    w[leaf_ids] = o_leaf[src] * s_leaf[src].prod(1)
    sh[leaf_ids, 0] = 0.5 * N_(P, 3)
    if M > 1:
        sh[leaf_ids, 1:M] = 0.05 * N_(P, M - 1, 3)
    ext = 3.0 * s_leaf[src].max(1, keepdim=True).values
    bmin[leaf_ids] = xyz[src] - ext
    bmax[leaf_ids] = xyz[src] + ext
    for lvl in range(len(lo) - 1, -1, -1):
        a, b = first_id[lvl], first_id[lvl] + lo[lvl].shape[0]
        ids = a + (~is_leaf[a:b]).nonzero().flatten()
        if ids.numel() == 0:
            continue
        c0 = sc_a[ids]; c1 = c0 + 1
        ws = (w[c0] + w[c1]).clamp_min(1e-30)
        f0, f1 = (w[c0] / ws)[:, None], (w[c1] / ws)[:, None]
        m = f0 * mu[c0] + f1 * mu[c1]
        var[ids] = f0 * (var[c0] + (mu[c0] - m) ** 2) + f1 * (var[c1] + (mu[c1] - m) ** 2)
        mu[ids] = m
        sh[ids] = f0[:, :, None] * sh[c0] + f1[:, :, None] * sh[c1]
        op[ids] = (f0[:, 0] * op[c0] + f1[:, 0] * op[c1]).clamp(0.0, 1.0)
        w[ids] = ws
        bmin[ids] = torch.minimum(bmin[c0], bmin[c1])
        bmax[ids] = torch.maximum(bmax[c0], bmax[c1])
    scales = var.clamp_min(1e-12).sqrt()
    scales[leaf_ids] = s_leaf[src]
    rots[leaf_ids] = q_leaf[src]
    cc = torch.where(is_leaf, 0, 2)
    nodes = torch.stack([depth_a, parent_a, torch.arange(N, device=device), is_leaf.long(), (~is_leaf).long(),
                         torch.where(is_leaf, 0, sc_a), cc], 1).to(torch.int32).contiguous()
    boxes = torch.zeros(N, 2, 4, **f32)
    boxes[:, 0, :3] = bmin
    boxes[:, 1, :3] = bmax
    boxes[:, 0, 3] = (bmax - bmin).max(1).values
    return Hierarchy(xyz=mu, shs=sh, alpha=op[:, None].contiguous(), log_scales=scales.log(), rots=rots, nodes=nodes,
                     boxes=boxes)


def merge_hierarchies(chunks) -> Hierarchy:
    """Several per-chunk hierarchies under one common root -- the shape the reference's GaussianHierarchyMerger
    produces from its chunks (scripts/full_train.py:240-250; BASELINE config 3 'merged 2-chunk toy hierarchy').
    New numbering: node 0 = the new root, nodes 1..k = the chunks' roots (contiguous children of the new root), then
    every chunk's remaining nodes in their own order, so children stay contiguous and 'Gaussian index == node index'
    still holds.  The new root's Gaussian is the moment-matched merge of the chunk roots (weights alpha * volume) with
    an axis-aligned covariance; its box is the union.  Works on whatever device the chunks live on."""
    k = len(chunks)
    assert k >= 1
    dev = chunks[0].nodes.device
    sizes = [int(c.num_nodes) for c in chunks]
    bases, b = [], 1 + k
    for n in sizes:
        bases.append(b)
        b += n - 1
    N = b

    def remap(c, ids):           # old node id of chunk c -> new id (negative ids stay negative)
        out = torch.where(ids == 0, torch.full_like(ids, 1 + c), ids + (bases[c] - 1))
        return torch.where(ids < 0, ids, out)

    new_of = [remap(c, torch.arange(sizes[c], device=dev, dtype=torch.int64)) for c in range(k)]
    perm = torch.empty(N, dtype=torch.int64, device=dev)         # new id -> row of the concatenated chunk arrays
    offs = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    for c in range(k):
        perm[new_of[c]] = torch.arange(sizes[c], device=dev, dtype=torch.int64) + int(offs[c])
    perm[0] = 0                                                   # placeholder row, overwritten below
    cat = lambda name: torch.cat([getattr(c, name) for c in chunks])[perm].clone()
    xyz, shs, alpha, log_scales, rots = (cat(n) for n in ("xyz", "shs", "alpha", "log_scales", "rots"))
    boxes = cat("boxes")
    nodes = torch.zeros(N, 7, dtype=torch.int32, device=dev)
    for c, ch in enumerate(chunks):
        nd = ch.nodes.to(torch.int64)
        ids = new_of[c]
        row = torch.stack([nd[:, 0] + 1,                                              # one level deeper
                           torch.where(nd[:, 1] < 0, torch.zeros_like(nd[:, 1]), remap(c, nd[:, 1])),
                           ids, nd[:, 3], nd[:, 4],
                           torch.where(nd[:, 6] > 0, remap(c, nd[:, 5]), torch.zeros_like(nd[:, 5])), nd[:, 6]], 1)
        nodes[ids] = row.to(torch.int32)
    nodes[0] = torch.tensor([0, -1, 0, 0, 1, 1, k], dtype=torch.int32, device=dev)
    # the new root's Gaussian and box
    r = torch.arange(1, 1 + k, device=dev)
    sc = log_scales[r].double().exp()
    w = (alpha[r, 0].double() * sc.prod(1)).clamp_min(1e-30)
    f = (w / w.sum())[:, None]
    m = (f * xyz[r].double()).sum(0)
    # axis-aligned second moments of the children: diag(R diag(s^2) R^T) + spread of the means
    q = rots[r].double()
    q = q / q.norm(dim=1, keepdim=True)
    qr, qx, qy, qz = q.unbind(1)
    R = torch.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qr * qz), 2 * (qx * qz + qr * qy),
                     2 * (qx * qy + qr * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qr * qx),
                     2 * (qx * qz - qr * qy), 2 * (qy * qz + qr * qx), 1 - 2 * (qx * qx + qy * qy)], 1).reshape(-1, 3, 3)
    var = ((R * R) * (sc ** 2)[:, None, :]).sum(2)
    var = (f * (var + (xyz[r].double() - m) ** 2)).sum(0)
    xyz[0] = m.float()
    log_scales[0] = var.clamp_min(1e-12).sqrt().log().float()
    rots[0] = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev)
    shs[0] = (f[:, :, None] * shs[r].double()).sum(0).float()
    alpha[0, 0] = float((f[:, 0] * alpha[r, 0].double()).sum().clamp(0.0, 1.0))
    boxes[0, 0, :3] = boxes[r, 0, :3].min(0).values
    boxes[0, 1, :3] = boxes[r, 1, :3].max(0).values
    boxes[0, 0, 3] = (boxes[0, 1, :3] - boxes[0, 0, :3]).max()
    boxes[0, 1, 3] = 0.0
    return Hierarchy(xyz=xyz, shs=shs, alpha=alpha, log_scales=log_scales, rots=rots, nodes=nodes, boxes=boxes)
