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


class _on_backward_stream:
    """Context: run the enclosed backward on ``stream`` (None: no-op) -- it first waits for everything the current
    stream holds, and every tensor of the forward that the backward kernels read is registered with the caching
    allocator as in use on that stream."""

    def __init__(self, stream, call, tensors):
        self.stream, self.call, self.tensors = stream, call, tensors
        self.ctx = None

    def __enter__(self):
        sb = self.stream
        if sb is None:
            return self
        sb.wait_stream(torch.cuda.current_stream(sb.device))
        c = self.call
        for t in tuple(self.tensors) + (c.geom, c.binb, c.img, getattr(c, "scratch", None)) + tuple(c.keep):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(sb)
        self.ctx = torch.cuda.stream(sb)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def wait_backward_stream():
    """Make the current stream wait for the backwards enqueued on ``_RasterizeGaussians.backward_stream``."""
    sb = _RasterizeGaussians.backward_stream
    if sb is not None:
        torch.cuda.current_stream(sb.device).wait_stream(sb)


class _RasterizeGaussians(torch.autograd.Function):
    # Optional {input name: preallocated float32 GPU tensor}: when set, the backward writes the gradients of
    # the Gaussian parameters straight into these buffers (hgs.dp.GradBucket views) instead of fresh tensors.
    grad_buffers = None
    grad_accumulate = False   # with grad_buffers: add to the buffers (accumulation over several views)
    # With grad_buffers: leave the SH part of every backward (dL_dshs, 81 % of the gradient bytes, and the view-direction
    # term of dL_dmeans3D) pending; finish_deferred_sh_backward() then does it for all pending views in ONE pass over
    # the coefficients.  Until then the shs / means3D gradient buffers are incomplete.
    defer_sh_backward = False
    pending_sh = []
    # Optional torch.cuda.Stream: every backward is enqueued there instead of on the forward's stream, after waiting for
    # what the forward's stream holds at that moment.  In the usual loop (forward j, backward j, forward j+1, ...) the
    # HBM-bound stages of one view then overlap with the ALU-bound compositing kernels of the next.  The returned
    # gradients (and grad_buffers) are valid ON THAT STREAM: call wait_backward_stream() before using them elsewhere.
    backward_stream = None

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        num_rendered, color, radii, geom, binb, img, invdepth, call = _C.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
            rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, rs.render_indices, rs.parent_indices,
            rs.interpolation_weights, rs.num_node_kids, rs.do_depth, getattr(_RasterizeGaussians, "variant", 0),
            prepare_backward=any(ctx.needs_input_grad))
        ctx.call = call
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(color, invdepth)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)      # no zero-filled "gradient" for radii / an unused invdepth
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_invdepth):
        color, invdepth = ctx.saved_tensors
        call = ctx.call
        if grad_color is None:
            grad_color = torch.zeros_like(color)
        cls = _RasterizeGaussians
        defer = bool(cls.defer_sh_backward and cls.grad_buffers is not None)
        with _on_backward_stream(cls.backward_stream, call, (color, invdepth, grad_color, grad_invdepth)):
            d_m2, d_col, d_op, d_m3, d_cov, d_sh, d_sc, d_rot = _C.rasterize_gaussians_backward(
                call, color, invdepth, grad_color, grad_invdepth, out=cls.grad_buffers,
                accumulate=cls.grad_accumulate, defer_sh=defer)
        if getattr(call, "deferred", None) is not None:
            if len(cls.pending_sh) >= 64:      # every pending view pins its workspaces
                raise RuntimeError("64 views are waiting for finish_deferred_sh_backward(); call it once per step")
            cls.pending_sh.append(call)
        ctx.call = None
        # order: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings
        return d_m3, d_m2, d_sh, d_col, d_op, d_sc, d_rot, d_cov, None


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Raw-parameter fast path: takes the optimiser's tensors of scene/gaussian_model.py
    (_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation) and applies the activations of
    scene/gaussian_model.py:108-128 inside the HIP kernels; gradients come back w.r.t. the raw tensors."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                raster_settings, activations):
        rs = raster_settings
        num_rendered, color, radii, geom, binb, img, invdepth, call = _C.rasterize_gaussians(
            rs.bg, xyz, None, opacity_raw, scaling_raw, rotation_raw, rs.scale_modifier, None,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, features_dc,
            rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, rs.render_indices, rs.parent_indices,
            rs.interpolation_weights, rs.num_node_kids, rs.do_depth, 0, sh_rest=features_rest,
            activations=activations, prepare_backward=any(ctx.needs_input_grad))
        ctx.call = call
        ctx.split = features_rest is not None and features_rest.numel() > 0
        ctx.save_for_backward(color, invdepth)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii, invdepth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_invdepth):
        color, invdepth = ctx.saved_tensors
        if grad_color is None:
            grad_color = torch.zeros_like(color)
        res = _C.rasterize_gaussians_backward(ctx.call, color, invdepth, grad_color, grad_invdepth,
                                              out=_RasterizeGaussians.grad_buffers,
                                              accumulate=_RasterizeGaussians.grad_accumulate)
        ctx.call = None
        d_m2, _, d_op, d_m3, _, d_sh, d_sc, d_rot = res[:8]
        d_rest = res[8] if ctx.split else None
        return d_m3, d_m2, d_sh, d_rest, d_op, d_sc, d_rot, None, None


def finish_deferred_sh_backward(accumulate=False):
    """Complete the backward of every view rendered since the last call with ``_RasterizeGaussians.defer_sh_backward``
    set: one pass over the SH coefficients for all of them (hgs_raster_sh_bwd_batched).  ``accumulate``: add to what
    the shs gradient buffer already holds instead of overwriting it."""
    pending, _RasterizeGaussians.pending_sh = _RasterizeGaussians.pending_sh, []
    sb = _RasterizeGaussians.backward_stream
    if sb is None:
        _C.sh_backward_batched(pending, accumulate=accumulate)
    else:                      # the pending views' backwards were enqueued there
        with torch.cuda.stream(sb):
            _C.sh_backward_batched(pending, accumulate=accumulate)


def sh_colors_batched(means3D, shs, sh_degree, campos_list):
    """See _C.sh_colors_batched: view-dependent colours for several cameras in one pass (no autograd: pass every
    returned colour tensor with requires_grad_() as colors_precomp and hand its gradient to
    sh_colors_batched_backward)."""
    return _C.sh_colors_batched(means3D, shs, sh_degree, campos_list)


def sh_colors_batched_backward(means3D, shs, sh_degree, campos_list, clamps, d_rgbs, d_shs, d_means3D, accumulate=False):
    """Runs on ``_RasterizeGaussians.backward_stream`` when that is set (the d_rgbs were produced there)."""
    sb = _RasterizeGaussians.backward_stream
    if sb is None:
        return _C.sh_colors_batched_backward(means3D, shs, sh_degree, campos_list, clamps, d_rgbs, d_shs, d_means3D,
                                             accumulate)
    sb.wait_stream(torch.cuda.current_stream(sb.device))
    for t in tuple(clamps) + tuple(campos_list):
        t.record_stream(sb)
    with torch.cuda.stream(sb):
        return _C.sh_colors_batched_backward(means3D, shs, sh_degree, campos_list, clamps, d_rgbs, d_shs, d_means3D,
                                             accumulate)


def rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                            raster_settings, opacity_activation="sigmoid"):
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
                                        rotation_raw, raster_settings, act)


class _LodGather(torch.autograd.Function):
    """In-op LOD interpolation: gather + lerp of node and parent attributes, and the matching
    scatter in the backward -- what gaussian_renderer/__init__.py:199-218 does with ~25 torch kernels."""

    @staticmethod
    def forward(ctx, render_indices, parent_indices, weights, means3D, scales, rotations, shs, opacities):
        n = render_indices.numel()
        ctx.save_for_backward(render_indices, parent_indices[:n], weights[:n], rotations)
        ctx.shapes = tuple(None if t is None else tuple(t.shape) for t in (means3D, scales, rotations, shs, opacities))
        return _C.lod_gather(render_indices, parent_indices[:n], weights[:n], means3D, scales, rotations, shs, opacities)

    @staticmethod
    def backward(ctx, g_means, g_scales, g_rot, g_shs, g_op):
        ri, pi, w, rotations = ctx.saved_tensors
        ds = _C.lod_gather_backward(ri, pi, w, rotations, (g_means, g_scales, g_rot, g_shs, g_op), ctx.shapes)
        return (None, None, None) + ds


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    rs = raster_settings
    if rs.render_indices is not None and rs.render_indices.numel() > 0:
        # hierarchy mode with the interpolation done in-op: the attribute tensors hold ALL hierarchy Gaussians,
        # render_indices / parent_indices / interpolation_weights select and blend the rows to draw
        if colors_precomp is not None or cov3Ds_precomp is not None:
            raise RuntimeError("in-op LOD interpolation needs shs and scales/rotations (no precomputed colours/covariances)")
        n = rs.render_indices.numel()
        means3D, scales, rotations, sh, opacities = _LodGather.apply(
            rs.render_indices, rs.parent_indices, rs.interpolation_weights, means3D, scales, rotations, sh, opacities)
        means2D = means2D[:n]
        empty = rs.render_indices.new_empty(0)
        raster_settings = rs._replace(render_indices=empty, parent_indices=empty)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

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
                                   cov3D_precomp, self.raster_settings)

    def forward_raw(self, xyz, means2D, features_dc, features_rest, opacity, scaling, rotation,
                    opacity_activation="sigmoid"):
        """Extension (not in the reference's API): render from the optimiser's raw tensors, activations fused."""
        return rasterize_gaussians_raw(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation,
                                       self.raster_settings, opacity_activation)


__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_gaussians_raw",
           "finish_deferred_sh_backward", "sh_colors_batched", "sh_colors_batched_backward", "wait_backward_stream",
           "_C"]
