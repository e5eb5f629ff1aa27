"""point_cloud.ply of a trained chunk (scene/gaussian_model.py:491-508 layout) -> merged 2-chunk .hier -- TEST FIXTURE.

Stands where scripts/full_train.py:212-250 runs the reference's C++ GaussianHierarchyCreator / Merger (hierarchy
CONSTRUCTION is out of scope): the trained Gaussians are split at the median x into two chunks, each
chunk gets hgs.hierarchy.build_hierarchy, the two are merged under one root and written in the upstream .hier layout
by gaussian_hierarchy._C.write_hierarchy."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd"), os.path.join(ROOT, "tests", "shims")):
    if p not in sys.path:
        sys.path.insert(0, p)


def scene_from_ply(ply):
    """Activated hgs.synth.Scene (sigmoid opacity, exp scales, normalised rotations, [P,16,3] SH) of a saved model."""
    from plyfile import PlyData
    from hgs import synth
    el = PlyData.read(ply).elements[0]
    col = lambda *names: torch.from_numpy(np.stack([np.asarray(el[n], np.float32) for n in names], 1))
    xyz = col("x", "y", "z")
    dc = col("f_dc_0", "f_dc_1", "f_dc_2")[:, None, :]                                  # [P,1,3]
    n_rest = sum(1 for p in el.properties if p.name.startswith("f_rest_"))
    rest = col(*[f"f_rest_{i}" for i in range(n_rest)]).reshape(-1, 3, n_rest // 3).transpose(1, 2)   # [P,15,3]
    shs = torch.zeros(xyz.shape[0], 16, 3)
    shs[:, :1], shs[:, 1:1 + rest.shape[1]] = dc, rest
    op = torch.sigmoid(col("opacity"))
    sc = torch.exp(col("scale_0", "scale_1", "scale_2"))
    rot = torch.nn.functional.normalize(col("rot_0", "rot_1", "rot_2", "rot_3"), dim=1)
    return synth.Scene(xyz.contiguous(), sc.contiguous(), rot.contiguous(), op.contiguous(), shs.contiguous(), 3)


def hier_from_ply(ply, out_path):
    from gaussian_hierarchy._C import write_hierarchy
    from hgs import hierarchy, synth
    sc = scene_from_ply(ply)
    left = sc.means3D[:, 0] < sc.means3D[:, 0].median()
    chunks = []
    for sel in (left, ~left):
        sub = synth.Scene(sc.means3D[sel], sc.scales[sel], sc.rotations[sel], sc.opacities[sel], sc.shs[sel], 3)
        chunks.append(hierarchy.build_hierarchy(sub))
    h = hierarchy.merge_hierarchies(chunks)
    write_hierarchy(out_path, h.xyz, h.shs, h.alpha, h.log_scales, h.rots, h.nodes, h.boxes)
    return sc.P, int(h.nodes.shape[0])


if __name__ == "__main__":
    print(hier_from_ply(sys.argv[1], sys.argv[2]))
