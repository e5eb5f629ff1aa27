import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
import torch, numpy as np
import bench
import diff_gaussian_rasterization as dgr
from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
from hgs import hierarchy, synth
dev = torch.device("cuda:0")
W, H = 3840, 2160
cam = synth.make_camera(W, H)
h = hierarchy.build_hierarchy_on_device(25_000_000, cam, dev, seed=0)
G = h.nodes.shape[0]
ri = torch.zeros(G, dtype=torch.int32, device=dev); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
w = torch.zeros(G, device=dev); ns = torch.zeros(G, dtype=torch.int32, device=dev)
sc = torch.exp(h.log_scales)
for tau_px in (3.0, 12.04):
    tau = (2 * tau_px + 1) * cam.tanfovx / (0.5 * W)
    n = expand_to_size(h.nodes, h.boxes, tau, cam.camera_center.to(dev), torch.zeros(3), ri, pi, ni)
    get_interpolation_weights(ni[:n], tau, h.nodes, h.boxes, cam.camera_center.cpu(), torch.zeros(3), w, ns)
    rs = bench._settings(dgr, cam, dev, do_depth=False, interpolation_weights=w, num_node_kids=ns, render_indices=ri[:n], parent_indices=pi)
    from diff_gaussian_rasterization import _C as C_
    with torch.no_grad():
        color, radii, _ = dgr.GaussianRasterizer(rs)(means3D=h.xyz, means2D=torch.zeros(G, 3, device=dev), shs=h.shs, opacities=h.alpha, scales=sc, rotations=h.rots)
    torch.cuda.synchronize()
    print("tau", tau_px, "cut", n, "L", C_.stats["last_L"], "radii max", int(radii.max()), "mean", float(radii[radii>0].float().mean()))
    r = radii[radii > 0].float()
    print("  radii quantiles", [float(torch.quantile(r[:5_000_000], q)) for q in (0.5, 0.9, 0.99, 0.999)])
