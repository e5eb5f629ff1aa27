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
