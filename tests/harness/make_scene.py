"""Synthetic COLMAP scene + hierarchy on disk for the acceptance harness -- TEST FIXTURE.

    <dir>/sparse/0/cameras.txt   one PINHOLE camera
    <dir>/sparse/0/images.txt    n views on a small circle, COLMAP world->camera poses (qw qx qy qz tx ty tz)
    <dir>/sparse/0/points3D.ply  the Gaussians' centres as an SfM point cloud (x y z nx ny nz red green blue)
    <dir>/images/view_XX.png     the 'ground truth': oracle renders of the same Gaussians
    <dir>/chunks.hier            two chunk hierarchies (left / right half of the points) merged under one root, in the
                                 upstream .hier layout, written by gaussian_hierarchy._C.write_hierarchy
Everything is seeded; the images are rendered with the CPU oracle (test infrastructure)."""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd"), os.path.join(ROOT, "tests", "shims")):
    if p not in sys.path:
        sys.path.insert(0, p)


def rotmat2qvec(R):
    """COLMAP's convention (w, x, y, z); same formula as scene/colmap_loader.py:277-291."""
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = R.flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0],
                  [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0], [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def view_pose(k, n_views, radius=None, look_at_depth=None):
    """(R camera->world, T of the world->camera transform) of view k, as written to images.txt by ``make``."""
    ang = 2 * math.pi * k / n_views
    if radius is None:
        c = np.array([0.4 * math.cos(ang), 0.3 * math.sin(ang), 0.0])
        yaw, pitch = 0.04 * math.cos(ang), 0.04 * math.sin(ang)
    else:
        c = np.array([radius * math.cos(ang), 0.75 * radius * math.sin(ang), 0.0])
        yaw = math.atan2(-c[0], look_at_depth)
        pitch = math.atan2(c[1], math.hypot(c[0], look_at_depth))
    Ry = np.array([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(pitch), -math.sin(pitch)], [0, math.sin(pitch), math.cos(pitch)]])
    Rc2w = Ry @ Rx
    return Rc2w, -Rc2w.T @ c


def make(path, n_points=240, n_views=6, W=64, H=48, seed=0, hier=True, radius=None, look_at_depth=None,
         renderer="oracle", n_sfm=None, s_px=(1.0, 4.0)):
    """radius / look_at_depth: cameras on an ellipse (radius, 0.75 radius) around the origin, turned towards the point
    (0, 0, look_at_depth) -- a scene extent of a sane size for the densification rules of train_single.py (default: the
    small 0.4 x 0.3 circle with a 0.04 rad wobble of the short chain test).
    renderer "hip" (GPU box, BASELINE configs[1] / [2] at their stated scale: a 1080p frame of 300 k Gaussians is out of
    the dense CPU oracle's reach): the ground-truth images are rendered by the HIP op itself -- they are training
    TARGETS, parity at that scale is checked separately (scripts/run_config2_config3.py).  n_sfm: only that many of the
    Gaussians' centres go into the SfM cloud (the optimisation has to densify towards the rest)."""
    from PIL import Image
    from plyfile import PlyData, PlyElement
    from hgs import hierarchy, synth
    from oracle import raster_oracle as ro
    os.makedirs(os.path.join(path, "sparse", "0"), exist_ok=True)
    os.makedirs(os.path.join(path, "images"), exist_ok=True)
    base = synth.make_camera(W, H)
    scene = synth.make_scene(n_points, base, seed=seed, s_px=s_px, z_range=(3.0, 8.0))
    dev_scene = None
    if renderer == "hip":
        import diff_gaussian_rasterization as dgr
        dev = torch.device("cuda:0")
        dev_scene = scene.to(dev)
        e_i, e_f = torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, device=dev)
    fx = W / (2 * base.tanfovx)
    fy = H / (2 * base.tanfovy)
    with open(os.path.join(path, "sparse", "0", "cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n")
        f.write(f"1 PINHOLE {W} {H} {fx:.9f} {fy:.9f} {W / 2:.4f} {H / 2:.4f}\n")
    lines = ["# Image list with two lines of data per image:\n"]
    bg = torch.zeros(3)
    for k in range(n_views):
        Rc2w, T = view_pose(k, n_views, radius, look_at_depth)
        q = rotmat2qvec(Rc2w.T)                       # COLMAP stores world -> camera
        name = f"view_{k:02d}.png"
        lines.append(f"{k + 1} {q[0]:.12f} {q[1]:.12f} {q[2]:.12f} {q[3]:.12f} {T[0]:.12f} {T[1]:.12f} {T[2]:.12f} 1 {name}\n\n")
        cam = synth.make_camera(W, H, R=Rc2w, T=T)
        with torch.no_grad():
            if dev_scene is not None:
                rs = dgr.GaussianRasterizationSettings(
                    image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg.to(dev), scale_modifier=1.0,
                    viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev), sh_degree=3,
                    campos=cam.camera_center.to(dev), prefiltered=False, debug=False, do_depth=False, render_indices=e_i,
                    parent_indices=e_i, interpolation_weights=e_f, num_node_kids=e_i)
                color = dgr.GaussianRasterizer(rs)(means3D=dev_scene.means3D, means2D=None, shs=dev_scene.shs,
                                                   opacities=dev_scene.opacities, scales=dev_scene.scales,
                                                   rotations=dev_scene.rotations)[0].cpu()
            else:
                color = ro.rasterize(scene.means3D, None, scene.shs, None, scene.opacities, scene.scales, scene.rotations,
                                     None, image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg,
                                     scale_modifier=1.0, viewmatrix=cam.world_view_transform,
                                     projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center).color
        img = (color.clamp(0, 1).permute(1, 2, 0).numpy() * 255 + 0.5).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(path, "images", name))
    with open(os.path.join(path, "sparse", "0", "images.txt"), "w") as f:
        f.writelines(lines)
    xyz = scene.means3D.numpy()
    rgb = np.clip((0.5 + ro.SH_C0 * scene.shs[:, 0].numpy()) * 255, 0, 255).astype(np.uint8)
    if n_sfm is not None and n_sfm < n_points:
        keep = np.sort(np.random.default_rng(seed + 5).choice(n_points, size=n_sfm, replace=False))
        xyz, rgb, n_points = xyz[keep], rgb[keep], n_sfm
    dt = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
          ("red", "u1"), ("green", "u1"), ("blue", "u1")]
    el = np.empty(n_points, dtype=dt)
    for i, n in enumerate("xyz"):
        el[n] = xyz[:, i]
        el["n" + n] = 0
    el["red"], el["green"], el["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    PlyData([PlyElement.describe(el, "vertex")]).write(os.path.join(path, "sparse", "0", "points3D.ply"))
    hier_path = None
    if hier:
        from gaussian_hierarchy._C import write_hierarchy
        left = scene.means3D[:, 0] < scene.means3D[:, 0].median()
        chunks = []
        for sel in (left, ~left):
            sub = synth.Scene(scene.means3D[sel], scene.scales[sel], scene.rotations[sel], scene.opacities[sel],
                              scene.shs[sel], 3)
            chunks.append(hierarchy.build_hierarchy(sub))
        h = hierarchy.merge_hierarchies(chunks)
        hier_path = os.path.join(path, "chunks.hier")
        write_hierarchy(hier_path, h.xyz, h.shs, h.alpha, h.log_scales, h.rots, h.nodes, h.boxes)
    return hier_path


if __name__ == "__main__":
    print(make(sys.argv[1]))
