"""Oracle-backed stand-ins for the three native extension layers, for running the REFERENCE'S OWN Python (render glue
and the three entry scripts, unmodified, from /root/reference) on a machine without a GPU.

TEST INFRASTRUCTURE ONLY.  The product packages have no CPU path; these classes exist because the build container has
no GPU and /root/reference does not exist on the GPU box, so the container is the only place where the reference's
callers can meet this repository's call signatures.  What they prove is the SURFACE: argument order, which tensors are
filled, return shapes / dtypes, autograd contract, files written -- the arithmetic behind them is the CPU oracle, not
the HIP kernels (those are checked against the same oracle by the ``-m gpu`` tests)."""
import types

import numpy as np
import torch


class OracleRasterC:
    """Stand-in for ``diff_gaussian_rasterization._C`` with the same two entry points as the HIP-backed module."""

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos,
                            prefiltered, debug, render_indices, parent_indices, interpolation_weights,
                            num_node_kids, do_depth, sh_rest=None, activations=0, prepare_backward=False):
        from oracle import raster_oracle as ro
        assert render_indices.numel() == 0 and parent_indices.numel() == 0
        ins = dict(means3D=means3D, shs=sh, colors_precomp=colors, opacities=opacity, scales=scales,
                   rotations=rotations, cov3D_precomp=cov3D_precomp)
        leaves = {k: (v.detach().clone().requires_grad_(True) if v is not None and v.numel() else None)
                  for k, v in ins.items()}
        m2 = torch.zeros(means3D.shape[0], 3, requires_grad=True)
        with torch.enable_grad():
            out = ro.rasterize(leaves["means3D"], m2, leaves["shs"], leaves["colors_precomp"], leaves["opacities"],
                               leaves["scales"], leaves["rotations"], leaves["cov3D_precomp"],
                               image_height=image_height, image_width=image_width, tanfovx=tanfovx, tanfovy=tanfovy,
                               bg=background, scale_modifier=scale_modifier, viewmatrix=viewmatrix,
                               projmatrix=projmatrix, sh_degree=degree, campos=campos,
                               interpolation_weights=interpolation_weights, num_node_kids=num_node_kids)
        call = types.SimpleNamespace(out=out, leaves=leaves, m2=m2, do_depth=do_depth)
        invd = out.invdepth.detach().float() if do_depth else torch.zeros(1, image_height, image_width)
        return out.binning.num_rendered, out.color.detach().float(), out.radii.clone(), None, None, None, invd, call

    @staticmethod
    def rasterize_gaussians_backward(call, color, invdepth, dL_dcolor, dL_dinvdepth, out=None, accumulate=False,
                                     defer_sh=False):
        with torch.enable_grad():
            loss = (call.out.color * dL_dcolor.double()).sum()
            if call.do_depth and dL_dinvdepth is not None:
                loss = loss + (call.out.invdepth * dL_dinvdepth.double()).sum()
        names = [k for k, v in call.leaves.items() if v is not None]
        grads = torch.autograd.grad(loss, [call.leaves[k] for k in names] + [call.m2], allow_unused=True)
        g = {k: (None if t is None else t.float()) for k, t in zip(names + ["m2"], grads)}
        return (g["m2"], g.get("colors_precomp"), g.get("opacities"), g.get("means3D"), g.get("cov3D_precomp"),
                g.get("shs"), g.get("scales"), g.get("rotations"))

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        vm = viewmatrix.to(torch.float32)
        return (means3D @ vm[:3, 2] + vm[3, 2]) > 0.2


def _np3(t):
    return t.detach().to("cpu", torch.float32).reshape(3).numpy() if torch.is_tensor(t) else np.asarray(t, np.float32)


def expand_to_size(nodes, boxes, size, viewpoint, viewdir, render_indices, parent_indices, nodes_for_render_indices):
    """``gaussian_hierarchy._C.expand_to_size`` (train_post.py:91-99, render_hierarchy.py:58-66) over the LOD oracle."""
    from oracle import lod_oracle as lo
    r, p, n = lo.expand_to_size(nodes.cpu().numpy(), boxes.cpu().numpy(), float(size), _np3(viewpoint))
    k = len(r)
    if k > min(render_indices.numel(), parent_indices.numel(), nodes_for_render_indices.numel()):
        raise RuntimeError("expand_to_size: output arrays too short")
    render_indices[:k] = torch.from_numpy(r)
    parent_indices[:k] = torch.from_numpy(p)
    nodes_for_render_indices[:k] = torch.from_numpy(n)
    return k


def get_interpolation_weights(node_indices, size, nodes, boxes, viewpoint, viewdir, interpolation_weights, num_siblings):
    from oracle import lod_oracle as lo
    n = int(node_indices.numel())
    if n == 0:
        return
    w, kids = lo.get_interpolation_weights(node_indices.cpu().numpy(), float(size), nodes.cpu().numpy(),
                                           boxes.cpu().numpy(), _np3(viewpoint))
    interpolation_weights[:n] = torch.from_numpy(w)
    num_siblings[:n] = torch.from_numpy(kids)


def distCUDA2(xyz):
    """``simple_knn._C.distCUDA2`` (scene/gaussian_model.py:190): mean squared distance to the 3 nearest neighbours."""
    x = xyz.detach().to("cpu", torch.float64)
    d2 = torch.cdist(x, x).pow(2)
    d2.fill_diagonal_(float("inf"))
    k = min(3, x.shape[0] - 1)
    return d2.topk(k, dim=1, largest=False).values.mean(dim=1).float() if k > 0 else torch.zeros(x.shape[0])


def install():
    """Swap the extension-module layers of the three drop-in packages for the stand-ins above.  load_hierarchy /
    write_hierarchy stay the real host code of libhgs.so (they need no GPU)."""
    import diff_gaussian_rasterization as dgr
    import gaussian_hierarchy._C as gh
    import simple_knn._C as knn
    dgr._C = OracleRasterC
    gh.expand_to_size = expand_to_size
    gh.get_interpolation_weights = get_interpolation_weights
    knn.distCUDA2 = distCUDA2
