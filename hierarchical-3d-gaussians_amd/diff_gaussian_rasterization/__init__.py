"""Drop-in ``diff_gaussian_rasterization`` for AMD MI355X (gfx950).

Same import names and call surface as the reference's hierarchy-rasterizer submodule
(/root/reference/.gitmodules:4-6), as used by gaussian_renderer/__init__.py:14,17,44-64,
105-113,247-277,319-389:

    GaussianRasterizationSettings(image_height=..., ..., num_node_kids=...)   # 17 keyword fields
    GaussianRasterizer(raster_settings=...)(means3D=..., means2D=..., shs=..., colors_precomp=...,
        opacities=..., scales=..., rotations=..., cov3D_precomp=...) -> (color, radii, invdepth)
    _C                                                                        # extension-module surface

The op is a ``torch.autograd.Function`` over the C ABI of libhgs.so (hand-written HIP).  There
is no CPU or PyTorch fallback: without the built library and a GPU every call raises.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    render_indices: torch.Tensor
    parent_indices: torch.Tensor
    interpolation_weights: torch.Tensor
    num_node_kids: torch.Tensor
    do_depth: bool = False


class RasterContext:
    """Options of ONE training loop's rasterizer calls that the reference's API has no place for (none of the
    reference's scripts uses them; ``GaussianRasterizer(raster_settings)`` without a context is the drop-in call).
    State lives here -- per object, never on the autograd class -- so two models, or a viewer render at another
    resolution (train_single.py:76-78), never share buffers, pending views or streams.

    grad_buffers       {input name: preallocated float32 GPU tensor} (``hgs.dp.GradBucket.views``; keys means3D, shs,
                       colors_precomp, opacities, scales, rotations, cov3D_precomp, shs_rest, means2D).  The backward
                       writes those gradients straight into the buffers and hands autograd ``None`` for them, so
                       ``.grad`` is never touched (no double counting under ``loss.backward()``) and no autograd node
                       consumes them.  Meant for LEAF inputs: nothing upstream of a buffered input receives a gradient.
    grad_accumulate    add to the buffers (later views of one optimizer step) instead of overwriting; ``means2D`` and
                       ``colors_precomp`` gradients are per-view quantities and always overwritten.
    defer_sh_backward  leave the SH part of every backward (dL_dshs, 81 % of the gradient bytes, and the
                       view-direction term of dL_dmeans3D) pending; ``finish_deferred_sh_backward()`` then does it for
                       all pending views in ONE pass over the coefficients.  Needs ``shs`` and ``means3D`` buffers.
    backward_stream    torch.cuda.Stream the backwards are enqueued on, after waiting for what the forward's stream
                       holds at that moment: in the usual loop (forward j, backward j, forward j+1, ...) the HBM-bound
                       stages of one view overlap with the ALU-bound compositing kernels of the next.  Gradients that
                       went into ``grad_buffers`` are valid ON THAT STREAM until ``wait_backward_stream()``; gradients
                       returned to autograd are made safe by letting the forward's stream wait for the backward
                       (correct, but it serialises the two streams -- buffer every input to get the overlap).
    skybox_points      in-op LOD interpolation only (non-empty ``render_indices``): the last ``skybox_points`` rows of
                       the attribute tensors are the skybox (scene/gaussian_model.py:371-383); they are appended to
                       the interpolated rows with weight 1 / 1 sibling, as gaussian_renderer/__init__.py:220-234 does.
    """

    def __init__(self, grad_buffers=None, backward_stream=None, defer_sh_backward=False, skybox_points=0):
        self.skybox_points = int(skybox_points)
        self.grad_buffers = grad_buffers
        self.grad_accumulate = False
        self.defer_sh_backward = bool(defer_sh_backward)
        self.backward_stream = backward_stream
        self.pending_sh = []

    def wait_backward_stream(self):
        """Make the current stream wait for the backwards enqueued on ``backward_stream``."""
        sb = self.backward_stream
        if sb is not None:
            torch.cuda.current_stream(sb.device).wait_stream(sb)

    def finish_deferred_sh_backward(self, accumulate=False):
        """Complete the backward of every view rendered since the last call with ``defer_sh_backward`` set: one pass
        over the SH coefficients for all of them (hgs_raster_sh_bwd_batched).  ``accumulate``: add to what the shs
        gradient buffer already holds instead of overwriting it."""
        pending, self.pending_sh = self.pending_sh, []
        sb = self.backward_stream
        if sb is None:
            _C.sh_backward_batched(pending, accumulate=accumulate)
        else:                      # the pending views' backwards were enqueued there
            with torch.cuda.stream(sb):
                _C.sh_backward_batched(pending, accumulate=accumulate)

    def sh_colors_batched_backward(self, means3D, shs, sh_degree, campos_list, clamps, d_rgbs, d_shs, d_means3D,
                                   accumulate=False):
        """``sh_colors_batched_backward`` on ``backward_stream`` (the d_rgbs were produced there)."""
        sb = self.backward_stream
        if sb is None:
            return _C.sh_colors_batched_backward(means3D, shs, sh_degree, campos_list, clamps, d_rgbs, d_shs,
                                                 d_means3D, accumulate)
        sb.wait_stream(torch.cuda.current_stream(sb.device))
        for t in tuple(clamps) + tuple(campos_list):
            t.record_stream(sb)
        with torch.cuda.stream(sb):
            return _C.sh_colors_batched_backward(means3D, shs, sh_degree, campos_list, clamps, d_rgbs, d_shs,
                                                 d_means3D, accumulate)


class _on_backward_stream:
    """Context manager: run the enclosed backward on ``stream`` (None: no-op) -- it first waits for everything the
    current stream holds, and every tensor of the forward that the backward kernels read is registered with the
    caching allocator as in use on that stream.  ``returned``: gradient tensors handed back to autograd (they were
    allocated and written on ``stream``): the forward's stream is made to wait for them on exit."""

    def __init__(self, stream, call, tensors):
        self.stream, self.call, self.tensors = stream, call, tensors
        self.ctx = None
        self.returned = []

    def __enter__(self):
        sb = self.stream
        if sb is None:
            return self
        self.main = torch.cuda.current_stream(sb.device)
        sb.wait_stream(self.main)
        c = self.call
        for t in tuple(self.tensors) + tuple(c.bufs) + tuple(c.keep):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(sb)
        self.ctx = torch.cuda.stream(sb)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            live = [t for t in self.returned if t is not None]
            if live:
                # autograd's next nodes (AccumulateGrad, the backward of the caller's activations, ...) run on the
                # forward's stream and read these tensors there
                self.main.wait_stream(self.stream)
                for t in live:
                    t.record_stream(self.main)
        return False


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, context):
        rs = raster_settings
        num_rendered, color, radii, geom, binb, img, invdepth, call = _C.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
            rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, rs.render_indices, rs.parent_indices,
            rs.interpolation_weights, rs.num_node_kids, rs.do_depth, prepare_backward=any(ctx.needs_input_grad))
        ctx.call = call
        ctx.context = context
        ctx.num_rendered = num_rendered
        # the inputs go through save_for_backward as well (the kernels read call.keep, the same storage): an in-place
        # update of a parameter between forward and backward then trips autograd's version check instead of silently
        # producing gradients of a mixed state
        ctx.save_for_backward(color, invdepth, *[t for t in (means3D, sh, colors_precomp, opacities, scales, rotations,
                                                             cov3Ds_precomp) if t is not None])
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)      # no zero-filled "gradient" for radii / an unused invdepth
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_invdepth):
        color, invdepth = ctx.saved_tensors[:2]
        call = ctx.call
        if call is None:
            raise RuntimeError("the rasterizer's backward ran twice (its workspaces are released after the first pass)")
        if grad_color is None:
            grad_color = torch.zeros_like(color)
        rc = ctx.context
        bufs = rc.grad_buffers if rc is not None else None
        defer = bool(rc is not None and rc.defer_sh_backward and bufs is not None)
        if defer and not ("shs" in bufs and "means3D" in bufs):
            raise RuntimeError("defer_sh_backward needs grad_buffers for shs and means3D")
        with _on_backward_stream(rc.backward_stream if rc is not None else None, call,
                                 (color, invdepth, grad_color, grad_invdepth)) as side:
            grads = _C.rasterize_gaussians_backward(
                call, color, invdepth, grad_color, grad_invdepth, out=bufs,
                accumulate=bool(rc is not None and rc.grad_accumulate), defer_sh=defer)
            d_m2, d_col, d_op, d_m3, d_cov, d_sh, d_sc, d_rot = grads
            named = dict(means3D=d_m3, means2D=d_m2, shs=d_sh, colors_precomp=d_col, opacities=d_op, scales=d_sc,
                         rotations=d_rot, cov3D_precomp=d_cov)
            if bufs is not None:      # buffered gradients are the caller's business, not autograd's
                named = {k: (None if k in bufs else v) for k, v in named.items()}
            side.returned = list(named.values())
        if getattr(call, "deferred", None) is not None:
            if len(rc.pending_sh) >= 64:      # every pending view pins its workspaces
                raise RuntimeError("64 views are waiting for finish_deferred_sh_backward(); call it once per step")
            rc.pending_sh.append(call)
        ctx.call = None
        # order: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings, context
        return (named["means3D"], named["means2D"], named["shs"], named["colors_precomp"], named["opacities"],
                named["scales"], named["rotations"], named["cov3D_precomp"], None, None)


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Raw-parameter fast path: takes the optimiser's tensors of scene/gaussian_model.py
    (_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation) and applies the activations of
    scene/gaussian_model.py:108-128 inside the HIP kernels; gradients come back w.r.t. the raw tensors."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                raster_settings, activations, context):
        rs = raster_settings
        num_rendered, color, radii, geom, binb, img, invdepth, call = _C.rasterize_gaussians(
            rs.bg, xyz, None, opacity_raw, scaling_raw, rotation_raw, rs.scale_modifier, None,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, features_dc,
            rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, rs.render_indices, rs.parent_indices,
            rs.interpolation_weights, rs.num_node_kids, rs.do_depth, sh_rest=features_rest,
            activations=activations, prepare_backward=any(ctx.needs_input_grad))
        ctx.call = call
        ctx.context = context
        ctx.num_rendered = num_rendered
        ctx.split = features_rest is not None and features_rest.numel() > 0
        ctx.save_for_backward(color, invdepth, *[t for t in (xyz, features_dc, features_rest, opacity_raw, scaling_raw,
                                                             rotation_raw) if t is not None])
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_invdepth):
        color, invdepth = ctx.saved_tensors[:2]
        call = ctx.call
        if call is None:
            raise RuntimeError("the rasterizer's backward ran twice (its workspaces are released after the first pass)")
        if grad_color is None:
            grad_color = torch.zeros_like(color)
        rc = ctx.context
        bufs = rc.grad_buffers if rc is not None else None
        with _on_backward_stream(rc.backward_stream if rc is not None else None, call,
                                 (color, invdepth, grad_color, grad_invdepth)) as side:
            res = _C.rasterize_gaussians_backward(call, color, invdepth, grad_color, grad_invdepth, out=bufs,
                                                  accumulate=bool(rc is not None and rc.grad_accumulate))
            d_m2, _, d_op, d_m3, _, d_sh, d_sc, d_rot = res[:8]
            named = dict(means3D=d_m3, means2D=d_m2, shs=d_sh, shs_rest=res[8] if ctx.split else None, opacities=d_op,
                         scales=d_sc, rotations=d_rot)
            if bufs is not None:
                named = {k: (None if k in bufs else v) for k, v in named.items()}
            side.returned = list(named.values())
        ctx.call = None
        return (named["means3D"], named["means2D"], named["shs"], named["shs_rest"], named["opacities"],
                named["scales"], named["rotations"], None, None, None)


def sh_colors_batched(means3D, shs, sh_degree, campos_list):
    """See _C.sh_colors_batched: view-dependent colours for several cameras in one pass (no autograd: pass every
    returned colour tensor with requires_grad_() as colors_precomp and hand its gradient to
    sh_colors_batched_backward / RasterContext.sh_colors_batched_backward)."""
    return _C.sh_colors_batched(means3D, shs, sh_degree, campos_list)


def sh_colors_batched_backward(means3D, shs, sh_degree, campos_list, clamps, d_rgbs, d_shs, d_means3D, accumulate=False):
    """On the current stream; with a backward stream use RasterContext.sh_colors_batched_backward."""
    return _C.sh_colors_batched_backward(means3D, shs, sh_degree, campos_list, clamps, d_rgbs, d_shs, d_means3D,
                                         accumulate)


def rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                            raster_settings, opacity_activation="sigmoid", context=None):
    """opacity_activation: 'sigmoid' (scene/gaussian_model.py:126-127), 'abs' (hierarchy mode, :393) or 'none'."""
    from hgs import _lib
    act = _lib.ACT_SCALE_EXP | _lib.ACT_ROT_NORMALIZE
    try:
        act |= {"sigmoid": _lib.ACT_OPACITY_SIGMOID, "abs": _lib.ACT_OPACITY_ABS, "none": 0}[opacity_activation]
    except KeyError:
        raise RuntimeError(f"unknown opacity_activation {opacity_activation!r}") from None
    rs = raster_settings
    if rs.render_indices is not None and rs.render_indices.numel() > 0:
        raise RuntimeError("the raw-parameter path takes already selected rows (empty render_indices)")
    return _RasterizeGaussiansRaw.apply(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw,
                                        rotation_raw, raster_settings, act, context)


class _RasterizeGaussiansLod(torch.autograd.Function):
    """In-op LOD interpolation: the op takes the FULL hierarchy attribute tensors plus the cut
    (render / parent indices, weights) and interpolates node and parent rows in registers inside its per-Gaussian
    kernels, forward and backward (hgs_raster_args.lod_*) -- what gaussian_renderer/__init__.py:199-234 does with ~25
    torch kernels and three materialised copies of the rows.  The backward's per-Gaussian kernels scatter the
    gradients to node and parent rows themselves (hgs_raster_args.lod_scatter; with 3M % 4 != 0 the row gradients go
    through memory and hgs_lod_gather_bwd)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, raster_settings, render_indices,
                parent_indices, weights, kids, skybox_points):
        rs = raster_settings
        empty = render_indices.new_empty(0)
        num_rendered, color, radii, geom, binb, img, invdepth, call = _C.rasterize_gaussians(
            rs.bg, means3D, None, opacities, scales, rotations, rs.scale_modifier, None,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
            rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, empty, empty, weights, kids, rs.do_depth,
            prepare_backward=any(ctx.needs_input_grad), lod=(render_indices, parent_indices, skybox_points))
        ctx.call = call
        ctx.num_rendered = num_rendered
        ctx.skybox_points = skybox_points
        ctx.shapes = tuple(tuple(t.shape) for t in (means3D, scales, rotations, sh, opacities))
        ctx.save_for_backward(color, invdepth, render_indices, parent_indices, weights, means3D, sh, opacities, scales,
                              rotations)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_invdepth):
        color, invdepth, ri, pi, w, means3D, sh, opacities, scales, rotations = ctx.saved_tensors
        call = ctx.call
        if call is None:
            raise RuntimeError("the rasterizer's backward ran twice (its workspaces are released after the first pass)")
        if grad_color is None:
            grad_color = torch.zeros_like(color)
        d_m2, _, d_op, d_m3, _, d_sh, d_sc, d_rot = _C.rasterize_gaussians_backward(call, color, invdepth, grad_color,
                                                                                   grad_invdepth)
        ctx.call = None
        if call.args.lod_scatter:        # scattered to node / parent rows inside the op's backward kernels
            return d_m3, d_m2, d_sh, d_op, d_sc, d_rot, None, None, None, None, None, None
        n, K = int(ri.numel()), ctx.skybox_points
        pi, w = pi[:n], w[:n]
        if K > 0:      # the skybox rows are their own parents with weight 1
            sky = torch.arange(means3D.shape[0] - K, means3D.shape[0], dtype=ri.dtype, device=ri.device)
            ri, pi = torch.cat((ri, sky)), torch.cat((pi, sky))
            w = torch.cat((w, torch.ones(K, dtype=w.dtype, device=w.device)))
        g_means, g_scales, g_rot, g_shs, g_op = _C.lod_gather_backward(ri, pi, w, rotations,
                                                                       (d_m3, d_sc, d_rot, d_sh, d_op), ctx.shapes)
        return g_means, d_m2, g_shs, g_op, g_scales, g_rot, None, None, None, None, None, None


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, context=None):
    rs = raster_settings
    if rs.render_indices is not None and rs.render_indices.numel() > 0:
        # hierarchy mode with the interpolation done in-op: the attribute tensors hold ALL hierarchy Gaussians,
        # render_indices / parent_indices / interpolation_weights select and blend the rows to draw
        if colors_precomp is not None or cov3Ds_precomp is not None:
            raise RuntimeError("in-op LOD interpolation needs shs and scales/rotations (no precomputed colours/covariances)")
        if context is not None and (context.grad_buffers is not None or context.backward_stream is not None):
            raise RuntimeError("grad_buffers / backward_stream are not available with in-op LOD interpolation: its "
                               "gradients are scattered to the hierarchy's rows after the op's own backward")
        n = rs.render_indices.numel()
        K = context.skybox_points if context is not None else 0
        w, kids = rs.interpolation_weights, rs.num_node_kids
        if K > 0:
            # the skybox rows (the last K rows of the arrays) are drawn as they are, with weight 1 and one sibling
            # (gaussian_renderer/__init__.py:220-234); the caller's tensors are left untouched
            w = torch.cat((w[:n], torch.ones(K, dtype=w.dtype, device=w.device)))
            kids = torch.cat((kids[:n], torch.ones(K, dtype=kids.dtype, device=kids.device)))
        return _RasterizeGaussiansLod.apply(means3D, means2D[:n + K], sh, opacities, scales, rotations, rs,
                                            rs.render_indices, rs.parent_indices, w, kids, K)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, context)


class GaussianRasterizer(nn.Module):
    """``GaussianRasterizer(raster_settings)`` is the reference's constructor (gaussian_renderer/__init__.py:64);
    ``context`` (a RasterContext, optional, keyword) carries this implementation's training-loop extras."""

    def __init__(self, raster_settings, context=None):
        super().__init__()
        self.raster_settings = raster_settings
        self.context = context

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings, self.context)

    def forward_raw(self, xyz, means2D, features_dc, features_rest, opacity, scaling, rotation,
                    opacity_activation="sigmoid"):
        """Extension (not in the reference's API): render from the optimiser's raw tensors, activations fused."""
        return rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation,
                                       self.raster_settings, opacity_activation, self.context)


__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "RasterContext", "rasterize_gaussians",
           "rasterize_gaussians_raw", "sh_colors_batched", "sh_colors_batched_backward", "_C"]
