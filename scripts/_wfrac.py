import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/hierarchical-3d-gaussians_amd") else os.environ["GRAFT_REPO_ROOT"]
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
import torch, math
from gaussian_hierarchy._C import expand_to_size, get_interpolation_weights
from hgs import hierarchy, synth
dev = torch.device("cuda:0")
for (W, H, leaves, taus) in ((3840, 2160, 25_000_000, (3.0,)), (1920, 1080, 500_000, (0.5, 3.0, 6.0, 15.0))):
    cam = synth.make_camera(W, H)
    h = hierarchy.build_hierarchy_on_device(leaves, cam, dev, seed=0)
    G = h.nodes.shape[0]
    ri = torch.zeros(G, dtype=torch.int32, device=dev); pi = torch.zeros_like(ri); ni = torch.zeros_like(ri)
    w = torch.zeros(G, device=dev); ns = torch.zeros(G, dtype=torch.int32, device=dev)
    for tau_px in taus:
        tau = (2 * tau_px + 1) * cam.tanfovx / (0.5 * W)
        n = expand_to_size(h.nodes, h.boxes, tau, cam.camera_center.to(dev), torch.zeros(3), ri, pi, ni)
        get_interpolation_weights(ni[:n], tau, h.nodes, h.boxes, cam.camera_center.cpu(), torch.zeros(3), w, ns)
        print(leaves, tau_px, "cut", n, "w==1 frac", float((w[:n] == 1).float().mean()), "w==0 frac", float((w[:n] == 0).float().mean()))
    del h, ri, pi, ni, w, ns
    torch.cuda.empty_cache()
