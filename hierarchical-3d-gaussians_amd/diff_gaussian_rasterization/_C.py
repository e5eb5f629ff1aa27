"""``diff_gaussian_rasterization._C`` -- the extension-module surface of the reference's
hierarchy-rasterizer submodule (imported at gaussian_renderer/__init__.py:17), implemented
over the C ABI of libhgs.so.  Thin glue only: argument checks, torch-owned workspaces,
pointer passing.  All arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C

import torch

from hgs import _lib


def _require_gpu(t: torch.Tensor, name: str):
    """Device / dtype checks; returns the tensor made contiguous (a sliced or expanded input, e.g. an override_color
    broadcast, is copied once -- the upstream extension calls .contiguous() on its inputs in the same way)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor (got {t.device}); this op has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def _opt(t, name, P, inner):
    """None for an absent/empty optional input, else the validated (contiguous) tensor."""
    if t is None or t.numel() == 0:
        return None
    t = _require_gpu(t, name)
    if t.shape[0] != P or t.numel() != P * inner:
        raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected [{P}, ...] with {inner} values per Gaussian")
    return t


def _small(t, name, n):
    if not torch.is_tensor(t):
        raise RuntimeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be on the GPU (got {t.device})")
    if t.dtype is not torch.float32 or not t.is_contiguous():      # (the usual case costs two attribute reads)
        t = t.to(torch.float32).contiguous()
    if t.numel() != n:
        raise RuntimeError(f"{name} must have {n} elements, got {t.numel()}")
    return t


def _lod(weights, kids, P):
    """The hierarchy tensors are accepted empty on any device (gaussian_renderer/__init__.py:39-42,244-245)."""
    if weights is None or kids is None or weights.numel() == 0 or kids.numel() == 0:
        return None, None
    if not weights.is_cuda or not kids.is_cuda:
        raise RuntimeError("non-empty interpolation_weights / num_node_kids must be GPU tensors")
    if weights.numel() < P or kids.numel() < P:
        raise RuntimeError(f"interpolation_weights/num_node_kids hold {weights.numel()}/{kids.numel()} entries, need >= {P}")
    w = weights.to(torch.float32).contiguous()
    k = kids.to(torch.int32).contiguous()
    return w, k


class _Call:
    """Everything one forward needs to hand back to the backward.  ``L`` = instances rendered; ``L_ws`` = the
    instance capacity the binning workspace was carved with (== L on the two-stage path).  The workspaces are byte
    ranges of at most three arena tensors (``bufs``): raw addresses for the C ABI, ``geom`` / ``binb`` / ``img`` /
    ``scratch`` as uint8 views on demand (tests, introspection)."""
    __slots__ = ("args", "keep", "bufs", "p_geom", "p_img", "p_bin", "p_bwd", "n_geom", "n_img", "n_bin", "n_bwd",
                 "L", "L_ws", "P", "W", "H", "device", "deferred")

    def _view(self, ptr, n):
        if not ptr:
            return None
        for t in self.bufs:
            off = ptr - t.data_ptr()
            if 0 <= off and off + n <= t.numel():
                return t[off:off + n]
        raise RuntimeError("workspace address outside the call's arenas")

    geom = property(lambda s: s._view(s.p_geom, s.n_geom))
    img = property(lambda s: s._view(s.p_img, s.n_img))
    binb = property(lambda s: s._view(s.p_bin, s.n_bin))
    scratch = property(lambda s: s._view(s.p_bwd, s.n_bwd))


_ALIGN = 256
_plan_cache = {}


def _plan(lib, P, W, H, L_ws):
    """{"geom", "bin", "img", "bwd"} workspace bytes of a frame, each rounded up to 256 -- one hgs_raster_ws_sizes round
    trip per distinct (P, W, H, L_ws), then a dictionary hit (the speculative capacity is quantised so that it repeats)."""
    key = (P, W, H, L_ws)
    pl = _plan_cache.get(key)
    if pl is None:
        geom, binb, img, bwd = (C.c_size_t() for _ in range(4))          # (the order of hgs_raster_ws_sizes)
        _lib.check(lib.hgs_raster_ws_sizes(P, W, H, L_ws, C.byref(geom), C.byref(binb), C.byref(img), C.byref(bwd)),
                   "hgs_raster_ws_sizes")
        up = lambda x: (x.value + _ALIGN - 1) // _ALIGN * _ALIGN
        pl = {"geom": up(geom), "bin": up(binb), "img": up(img), "bwd": up(bwd)}
        if len(_plan_cache) > 512:
            _plan_cache.clear()
        _plan_cache[key] = pl
    return pl


def size_class(n, steps_per_octave=4, floor=1 << 20):
    """``n`` rounded up to the next size class: 2^(k / steps_per_octave) above ``floor`` (19 % steps: 9 % over-allocation on
    average).  The workspace arena of a frame and the row capacity of its gradient tensors change with every view and
    every cut (train_post.py renders another cut per iteration); asked for byte-exact sizes, torch's caching allocator
    kept splitting and re-requesting 100 MB-blocks -- 11.0 GB reserved for a 3.1 GB peak in the round-5 run of
    train_post.py at 1080p, and a hipMalloc inside one step in three of the trained-scale bench.  A handful of classes per
    octave makes every request an exact fit of a cached block."""
    n = int(n)
    if n <= floor:
        return n
    e = (n - 1).bit_length() - 1                      # 2^e < n <= 2^(e + 1)
    for k in range(1, steps_per_octave + 1):
        c = int(round(2.0 ** (e + k / steps_per_octave)))
        c = (c + _ALIGN - 1) // _ALIGN * _ALIGN
        if c >= n:
            return c
    return 1 << (e + 1)


arena_stats = {"requested_bytes": 0, "class_bytes": 0, "arenas": 0}


def _arena(nbytes, dev):
    c = size_class(nbytes)
    arena_stats["requested_bytes"] += nbytes
    arena_stats["class_bytes"] += c
    arena_stats["arenas"] += 1
    return torch.empty(c, dtype=torch.uint8, device=dev)


def _rows_empty(P, inner, dev):
    """A fresh [P, *inner] float32 tensor whose STORAGE holds a size class of rows (see size_class): the gradient tensors
    of a cut of another size then reuse the cached block of the last one."""
    per = 4
    for d in inner:
        per *= int(d)
    cap = max(P, size_class(P * per) // max(per, 1))
    return torch.empty((cap,) + tuple(inner), dtype=torch.float32, device=dev)[:P]


# Instance count of the previous forward per (device, width, height, Gaussians): lets the next forward of the same
# shape size its binning workspace speculatively (1.25 x) and enqueue the whole pipeline without waiting for the host
# (hgs_raster_fwd).  A render at another resolution (the viewer's, train_single.py:76-78) has its own entry.
_last_L = {}
SPECULATIVE = True
SPEC_GROWTH = 1.25       # speculative instance capacity = SPEC_GROWTH * (previous L of this shape) + SPEC_SLACK
SPEC_SLACK = 65536
SPEC_DECAY = 0.99        # what the remembered count (per shape) / instances per row (per resolution) keeps of its maximum per call
stats = {"speculative_calls": 0, "capacity_misses": 0, "last_L": 0}     # counters (bench.py reports them)


def _build_args(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos,
                debug, interpolation_weights, num_node_kids, do_depth, sh_rest=None, activations=0, lod=None):
    """``lod`` = (render_indices, parent_indices, skybox_points): in-kernel LOD interpolation -- the attribute tensors
    hold all hierarchy Gaussians (G rows), the op renders P = len(render_indices) + skybox_points rows."""
    means3D = _require_gpu(means3D, "means3D")
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    rows = means3D.shape[0]                      # rows of the attribute arrays
    P = rows
    ri = pi = None
    if lod is not None:
        ri, pi, K = lod
        for t, name in ((ri, "render_indices"), (pi, "parent_indices")):
            if not t.is_cuda or t.dtype != torch.int32:
                raise RuntimeError(f"{name} must be an int32 GPU tensor")
        ri, pi = ri.contiguous(), pi.contiguous()
        n = int(ri.numel())
        if pi.numel() < n or K < 0 or K > rows:
            raise RuntimeError("parent_indices shorter than render_indices, or more skybox rows than rows")
        P = n + int(K)
    if sh is not None and sh.numel() == 0:
        sh = None
    if sh is not None:
        sh = _require_gpu(sh, "shs")
        if sh.dim() != 3 or sh.shape[0] != rows or sh.shape[2] != 3:
            raise RuntimeError("shs must have dimensions (num_points, num_coeffs, 3)")
    colors = _opt(colors, "colors_precomp", rows, 3)
    scales = _opt(scales, "scales", rows, 3)
    rotations = _opt(rotations, "rotations", rows, 4)
    cov3D_precomp = _opt(cov3D_precomp, "cov3D_precomp", rows, 6)
    if rows > 0:
        opacity = _require_gpu(opacity, "opacities")
        if opacity.numel() != rows:
            raise RuntimeError(f"opacities must hold {rows} values")
        if (sh is None) == (colors is None):
            raise RuntimeError("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise RuntimeError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    M = sh.shape[1] if sh is not None else 0
    if sh_rest is not None and sh_rest.numel() == 0:
        sh_rest = None
    if sh_rest is not None:
        # raw-parameter path: sh = features_dc [P,1,3], sh_rest = features_rest [P,M-1,3]
        sh_rest = _require_gpu(sh_rest, "shs_rest")
        if sh is None or sh.shape[1] != 1 or sh_rest.dim() != 3 or sh_rest.shape[0] != rows or sh_rest.shape[2] != 3:
            raise RuntimeError("split SH storage needs features_dc (num_points, 1, 3) and features_rest (num_points, M-1, 3)")
        M = 1 + sh_rest.shape[1]
    bg = _small(background, "bg", 3)
    vm = _small(viewmatrix, "viewmatrix", 16)
    pm = _small(projmatrix, "projmatrix", 16)
    cp = _small(campos, "campos", 3)
    w, k = _lod(interpolation_weights, num_node_kids, P)
    if lod is not None and (w is None or sh is None or scales is None or sh_rest is not None or activations):
        raise RuntimeError("in-op LOD interpolation needs shs, scales / rotations, interpolation_weights and "
                           "num_node_kids (no precomputed colours / covariances, no raw-parameter path)")
    a = _lib.RasterArgs()
    a.P, a.M, a.sh_degree = P, M, int(degree)
    a.width, a.height = int(image_width), int(image_height)
    a.tanfovx, a.tanfovy, a.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
    a.do_depth, a.debug, a.accumulate_grads = int(bool(do_depth)), int(bool(debug)), 0
    p = _lib.ptr
    a.bg, a.viewmatrix, a.projmatrix, a.campos = p(bg), p(vm), p(pm), p(cp)
    a.means3D, a.shs, a.colors_precomp, a.opacities = p(means3D), p(sh), p(colors), p(opacity)
    a.scales, a.rotations, a.cov3D_precomp = p(scales), p(rotations), p(cov3D_precomp)
    a.interpolation_weights, a.num_node_kids = p(w), p(k)
    if LOD_REMAP not in ("opacity", "alpha"):
        raise RuntimeError(f"LOD_REMAP must be 'opacity' or 'alpha', not {LOD_REMAP!r}")
    a.lod_per_pixel = int(LOD_REMAP == "alpha" and w is not None)
    a.shs_rest, a.activations = p(sh_rest), int(activations)
    if lod is not None:
        a.lod_render_indices, a.lod_parent_indices = p(ri), p(pi)
        a.lod_n, a.lod_rows = int(ri.numel()), rows
    keep = (bg, vm, pm, cp, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, w, k, sh_rest, ri, pi)
    return a, keep, P, M


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, render_indices, parent_indices, interpolation_weights,
                        num_node_kids, do_depth, sh_rest=None, activations=0, prepare_backward=False, lod=None):
    """Forward.  Returns (num_rendered, color[3,H,W], radii[P], geomBuffer, binningBuffer, imgBuffer,
    invdepth[1,H,W], call) -- ``call`` carries the argument block for the backward.  ``prepare_backward``: a backward
    will follow (hgs_raster_args.prepare_backward: K1 also stores d(rgb)/d(direction) for the SH backward)."""
    if lod is None and ((render_indices is not None and render_indices.numel() > 0) or
                        (parent_indices is not None and parent_indices.numel() > 0)):
        raise RuntimeError("rasterize_gaussians expects already gathered rows; non-empty render_indices / "
                           "parent_indices are resolved by GaussianRasterizer.forward (lod=...) before this call")
    lib = _lib.lib()
    a, keep, P, M = _build_args(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                cov3D_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, image_height,
                                image_width, sh, degree, campos, debug, interpolation_weights, num_node_kids,
                                do_depth, sh_rest, activations, lod)
    a.prepare_backward = int(bool(prepare_backward))
    dev = means3D.device
    H, W = int(image_height), int(image_width)
    u8 = dict(dtype=torch.uint8, device=dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    # outputs are allocated before the stage-1 sync so that only the L-sized workspace sits between the stages
    color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    invdepth = torch.empty(1, H, W, dtype=torch.float32, device=dev) if do_depth else \
        torch.zeros(1, H, W, dtype=torch.float32, device=dev)
    L = C.c_uint32(0)
    devi = dev.index or 0
    want_bwd = bool(prepare_backward)
    call = _Call()
    call.p_bin = call.p_bwd = call.n_bin = call.n_bwd = 0
    # with lod= the number of rendered rows changes with every cut: the hierarchy (its row count) is the workload
    shape_key = (devi, W, H, P) if lod is None else (devi, W, H, "lod", means3D.shape[0])
    prev = _last_L.get(shape_key) if SPECULATIVE else None
    any_key = (devi, W, H, "any")
    if prev is None and SPECULATIVE and lod is None and P > 0:
        # A row count never seen at this resolution -- train_post.py hands the op another cut every iteration
        # (gaussian_renderer/__init__.py:199-235 gathers the rows before the call), densification changes P every 300:
        # scale the last frame's instance count of this resolution by the ratio of the row counts instead of falling
        # back to the two-stage path with its host round trip (1080p run of round 5: 92 % of train_post's calls did)
        near = _last_L.get(any_key)             # (rows of the last frame, decaying maximum of instances per row)
        if near is not None and near[0] > 0 and 0.25 <= P / near[0] <= 4.0:
            prev = int(near[1] * P) + 1
    L_ws = 0
    done = False
    if prev is not None and P > 0:
        # no-bubble path: ONE arena (geometry, per-pixel state, binning, and the backward's scratch when a backward
        # was announced), everything enqueued before the host learns L.  The capacity is rounded up in steps of 3-6 % so
        # that consecutive views of a shape ask for the same plan (a dictionary hit instead of a size query).
        L_ws = int(prev * SPEC_GROWTH) + SPEC_SLACK
        q = 1 << max(L_ws.bit_length() - 5, 0)
        L_ws = (L_ws + q - 1) // q * q
        stats["speculative_calls"] += 1
        pl = _plan(lib, P, W, H, L_ws)
        n_g, n_i, n_b, n_w = pl["geom"], pl["img"], pl["bin"], (pl["bwd"] if want_bwd else 0)
        arena = _arena(n_g + n_i + n_b + n_w, dev)
        base = arena.data_ptr()
        call.bufs = [arena]
        call.p_geom, call.n_geom = base, n_g
        call.p_img, call.n_img = base + n_g, n_i
        call.p_bin, call.n_bin = base + n_g + n_i, n_b
        if want_bwd:
            call.p_bwd, call.n_bwd = base + n_g + n_i + n_b, n_w
        rc = lib.hgs_raster_fwd(C.byref(a), call.p_geom, call.p_bin, call.p_img, L_ws, _lib.ptr(radii),
                                _lib.ptr(color), _lib.ptr(invdepth) if do_depth else None, C.byref(L),
                                _stream(dev), devi)
        if rc == _lib.ERR_CAPACITY:
            stats["capacity_misses"] += 1        # the scene grew by more than 25 %: finish on the exact two-stage path
        else:
            _lib.check(rc, "hgs_raster_fwd")
            done = True
    else:
        pl = _plan(lib, P, W, H, 0)
        n_g, n_i = pl["geom"], pl["img"]
        arena = _arena(n_g + n_i, dev)
        base = arena.data_ptr()
        call.bufs = [arena]
        call.p_geom, call.n_geom, call.p_img, call.n_img = base, n_g, base + n_g, n_i
        _lib.check(lib.hgs_raster_fwd_stage1(C.byref(a), call.p_geom, _lib.ptr(radii), C.byref(L),
                                             _stream(dev), devi), "hgs_raster_fwd_stage1")
    if not done:
        L_ws = L.value
        pl = _plan(lib, P, W, H, L_ws)
        n_b, n_w = pl["bin"], (pl["bwd"] if want_bwd else 0)
        arena2 = _arena(n_b + n_w, dev)
        call.bufs.append(arena2)
        call.p_bin, call.n_bin = arena2.data_ptr(), n_b
        call.p_bwd, call.n_bwd = (call.p_bin + n_b, n_w) if want_bwd else (0, 0)
        _lib.check(lib.hgs_raster_fwd_stage2(C.byref(a), call.p_geom, call.p_bin, call.p_img, L_ws,
                                             _lib.ptr(color), _lib.ptr(invdepth) if do_depth else None,
                                             _stream(dev), devi), "hgs_raster_fwd_stage2")
    # a view (a cut) that shrank must not make the next, larger one overflow the speculative capacity: a DECAYING
    # MAXIMUM, per shape and -- as instances per row -- per resolution (train_post.py at 1080p, round 5: 78 of 1 000
    # calls outgrew a capacity taken from the last call alone, each a full retry)
    stored = _last_L.pop(shape_key, None)              # (re-inserted at the end: the dict is kept in order of last use)
    _last_L[shape_key] = max(L.value, int(SPEC_DECAY * (stored or 0)))
    if lod is None and P > 0:
        near = _last_L.pop(any_key, None)
        _last_L[any_key] = (P, max(L.value / P, SPEC_DECAY * (near[1] if near else 0.0)))
    while len(_last_L) > 64:                           # forget the shape that was used longest ago
        _last_L.pop(next(iter(_last_L)))
    stats["last_L"] = L.value
    call.args, call.keep = a, keep
    call.L, call.L_ws, call.P, call.W, call.H, call.device = L.value, L_ws, P, W, H, dev
    call.deferred = None
    # (geomBuffer, binningBuffer, imgBuffer of the upstream signature: call.geom / call.binb / call.img on demand)
    return L.value, color, radii, None, None, None, invdepth, call


# How non-empty interpolation_weights / num_node_kids are used (include/hgs.h: hgs_raster_args.lod_per_pixel): "opacity" --
# the default, a per-Gaussian remap of the opacity in the per-Gaussian kernel -- or "alpha": the same remap applied per
# pixel to alpha = o G inside the compositing kernels, under which k coincident children at weight 0 composite exactly like
# their parent.  Which of the two the reference's CUDA kernel implements cannot be read off its
# checkout (gaussian_renderer/__init__.py:258-265 passes the tensors on); the pin kit's raster_post golden decides, and
# following it is this assignment.
LOD_REMAP = "opacity"

lod_scatter_in_kernel = True     # False: row gradients through memory + lod_gather_backward (kept for 3M % 4 != 0; tests)


def rasterize_gaussians_backward(call, color, invdepth, dL_dcolor, dL_dinvdepth, out=None, accumulate=False,
                                 defer_sh=False):
    """Backward.  Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh,
    dL_dscales, dL_drotations) -- plus dL_dsh_rest as a 9th entry for a forward with split SH storage; entries
    for absent inputs are None.  ``out``: optional dict of
    preallocated float32 GPU tensors (keys means3D, shs, colors_precomp, opacities, scales, rotations,
    cov3D_precomp) the gradients are written into -- e.g. the views of a data-parallel flat bucket;
    ``accumulate``: add to their contents (gradient accumulation over the views of one optimizer step).
    ``defer_sh``: leave dL_dsh (and the view-direction part of dL_dmeans3D) to a later ``sh_backward_batched`` call
    over several views; ``call.deferred`` then holds what that call needs."""
    lib = _lib.lib()
    a, P, dev = call.args, call.P, call.device
    f32 = dict(dtype=torch.float32, device=dev)
    bg, vm, pm, cp, means3D, sh, colors, opacity, scales, rotations, cov3D, w, k, sh_rest = call.keep[:14]
    dL_dcolor = dL_dcolor.to(torch.float32).contiguous()
    use_depth = bool(a.do_depth) and dL_dinvdepth is not None
    if use_depth:
        dL_dinvdepth = dL_dinvdepth.to(torch.float32).contiguous()
    g = _lib.RasterGrads()

    def buf(name, shape):
        t = None if out is None else out.get(name)
        if t is not None:
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != int(torch.Size(shape).numel()):
                raise RuntimeError(f"gradient buffer for {name} must be a contiguous float32 GPU tensor of shape {tuple(shape)}")
            return t.view(*shape)
        return _rows_empty(shape[0], shape[1:], dev)

    # In-op LOD interpolation: the backward's per-Gaussian kernels scatter node / parent gradients themselves into
    # FULL-size, zero-filled arrays (hgs_raster_args.lod_scatter); only with 3M % 4 != 0 do the row gradients go
    # through memory and lod_gather_backward.
    lod_scatter = (lod_scatter_in_kernel and bool(a.lod_render_indices) and (int(a.M) * 3) % 4 == 0 and out is None
                   and not defer_sh)
    a.lod_scatter = int(lod_scatter)
    d_m2 = buf("means2D", (P, 3))
    if lod_scatter:
        G = int(a.lod_rows)
        d_m3 = torch.zeros(G, 3, **f32)
        d_op = torch.zeros(G, 1, **f32)
        d_sh = torch.zeros((G,) + tuple(sh.shape[1:]), **f32)
    else:
        d_m3 = buf("means3D", (P, 3))
        d_op = buf("opacities", (P, 1))
        d_sh = buf("shs", (P,) + tuple(sh.shape[1:])) if sh is not None else None  # P rows (!= sh.shape[0] with lod=)
    d_shr = buf("shs_rest", tuple(sh_rest.shape)) if sh_rest is not None else None
    d_col = buf("colors_precomp", (P, 3)) if colors is not None else None
    if lod_scatter:
        d_sc, d_rot = torch.zeros(G, 3, **f32), torch.zeros(G, 4, **f32)
    else:
        d_sc = buf("scales", (P, 3)) if scales is not None else None
        d_rot = buf("rotations", (P, 4)) if rotations is not None else None
    d_cov = buf("cov3D_precomp", (P, 6)) if cov3D is not None else None
    p = _lib.ptr
    g.dL_dmeans3D, g.dL_dmeans2D, g.dL_dshs, g.dL_dcolors = p(d_m3), p(d_m2), p(d_sh), p(d_col)
    g.dL_dopacity, g.dL_dscales, g.dL_drotations, g.dL_dcov3D = p(d_op), p(d_sc), p(d_rot), p(d_cov)
    g.dL_dshs_rest = p(d_shr)
    a.accumulate_grads = int(bool(accumulate and out is not None))
    a.defer_sh_bwd = int(bool(defer_sh and sh is not None and sh_rest is None))
    # the backward's scratch (instance records + per-Gaussian colour gradients) was carved from the forward's arena when
    # the forward knew a backward would follow; it needs no initialisation, K7 writes every instance record
    p_bwd = call.p_bwd
    if not p_bwd:
        n_w = _plan(lib, P, call.W, call.H, call.L_ws)["bwd"]
        extra = _arena(n_w, dev)
        call.bufs.append(extra)
        p_bwd, call.p_bwd, call.n_bwd = extra.data_ptr(), extra.data_ptr(), n_w
    _lib.check(lib.hgs_raster_bwd(C.byref(a), call.p_geom, call.p_bin, call.p_img, p_bwd, call.L_ws,
                                  p(color), p(invdepth) if use_depth else None, p(dL_dcolor),
                                  p(dL_dinvdepth) if use_depth else None, C.byref(g), _stream(dev),
                                  dev.index or 0), "hgs_raster_bwd")
    call.deferred = (p_bwd, d_sh, d_m3) if a.defer_sh_bwd else None
    if sh_rest is not None:
        return d_m2, d_col, d_op, d_m3, d_cov, d_sh, d_sc, d_rot, d_shr
    return d_m2, d_col, d_op, d_m3, d_cov, d_sh, d_sc, d_rot


def sh_backward_batched(calls, accumulate=False):
    """Finish the SH part of the backward of several views (``defer_sh=True``) in one pass over the coefficients.
    All views must share means3D / shs and their gradient buffers (``out=`` of the backward).  dL_dsh = (accumulate ?
    dL_dsh : 0) + sum over the views; dL_dmeans3D += the views' view-direction terms."""
    if not calls:
        return
    lib = _lib.lib()
    first = calls[0]
    means3D, sh = first.keep[4], first.keep[5]
    _, d_sh, d_m3 = first.deferred
    for c in calls:
        if c.deferred is None:
            raise RuntimeError("sh_backward_batched needs calls whose backward ran with defer_sh=True")
        if c.keep[4].data_ptr() != means3D.data_ptr() or c.keep[5].data_ptr() != sh.data_ptr() or \
                c.deferred[1].data_ptr() != d_sh.data_ptr() or c.deferred[2].data_ptr() != d_m3.data_ptr():
            raise RuntimeError("deferred views must share means3D / shs and their gradient buffers")
    dev = first.device
    for i in range(0, len(calls), _lib.MAX_DEFERRED_VIEWS):
        chunk = calls[i:i + _lib.MAX_DEFERRED_VIEWS]
        arr = (_lib.ShBwdView * len(chunk))()
        for v, c in zip(arr, chunk):
            v.geom_ws, v.bwd_ws, v.campos, v.L = c.p_geom, c.deferred[0], c.keep[3].data_ptr(), c.L_ws
        _lib.check(lib.hgs_raster_sh_bwd_batched(arr, len(chunk), first.P, first.args.M, first.args.sh_degree,
                                                 _lib.ptr(means3D), _lib.ptr(sh), _lib.ptr(d_sh), _lib.ptr(d_m3),
                                                 int(bool(accumulate or i > 0)), _stream(dev), dev.index or 0),
                   "hgs_raster_sh_bwd_batched")
    for c in calls:
        c.deferred = None


def _color_views(campos_list, rgbs, clamps, d_rgbs):
    n = len(campos_list)
    if not 1 <= n <= _lib.MAX_DEFERRED_VIEWS:
        raise RuntimeError(f"1..{_lib.MAX_DEFERRED_VIEWS} views per call")
    arr = (_lib.ShColorView * n)()
    keep = []
    for i, v in enumerate(arr):
        cp = _small(campos_list[i], "campos", 3)
        keep.append(cp)
        v.campos = cp.data_ptr()
        v.rgb = rgbs[i].data_ptr() if rgbs is not None else None
        v.clamp = clamps[i].data_ptr()
        v.d_rgb = d_rgbs[i].data_ptr() if d_rgbs is not None else None
    return arr, keep


def sh_colors_batched(means3D, shs, sh_degree, campos_list):
    """Colours of the same Gaussians seen from several cameras, one pass over the SH coefficients: the batched HIP
    form of the reference's convert_SHs_python branch (gaussian_renderer/__init__.py:84-89).  Returns (rgbs, clamps):
    per view a [P,3] float32 tensor to pass as ``colors_precomp`` and the uint8 clamp mask the backward needs."""
    means3D = _require_gpu(means3D, "means3D")
    shs = _require_gpu(shs, "shs")
    P, M = means3D.shape[0], shs.shape[1]
    dev = means3D.device
    rgbs = [torch.empty(P, 3, dtype=torch.float32, device=dev) for _ in campos_list]
    clamps = [torch.empty(P, dtype=torch.uint8, device=dev) for _ in campos_list]
    B = _lib.MAX_DEFERRED_VIEWS
    for i in range(0, len(campos_list), B):          # more views than one launch takes: one pass per group of 8
        arr, keep = _color_views(campos_list[i:i + B], rgbs[i:i + B], clamps[i:i + B], None)
        _lib.check(_lib.lib().hgs_sh_colors_batched(arr, len(arr), P, M, int(sh_degree), _lib.ptr(means3D),
                                                    _lib.ptr(shs), _stream(dev), dev.index or 0),
                   "hgs_sh_colors_batched")
    return rgbs, clamps


def sh_colors_batched_backward(means3D, shs, sh_degree, campos_list, clamps, d_rgbs, d_shs, d_means3D, accumulate=False):
    """d_rgbs: per view dL/d(colors_precomp) [P,3].  d_shs = (accumulate ? d_shs : 0) + sum over the views;
    d_means3D += the views' view-direction terms (both float32 GPU tensors shaped like shs / means3D)."""
    P, M = means3D.shape[0], shs.shape[1]
    dev = means3D.device
    d_rgbs = [g.to(torch.float32).contiguous() for g in d_rgbs]
    means3D, shs = _require_gpu(means3D, "means3D"), _require_gpu(shs, "shs")
    for t, name in ((d_shs, "d_shs"), (d_means3D, "d_means3D")):     # written in place
        if _require_gpu(t, name) is not t:
            raise RuntimeError(f"{name} must be contiguous (it is written in place)")
    B = _lib.MAX_DEFERRED_VIEWS
    for i in range(0, len(campos_list), B):
        arr, keep = _color_views(campos_list[i:i + B], None, clamps[i:i + B], d_rgbs[i:i + B])
        _lib.check(_lib.lib().hgs_sh_colors_batched_bwd(arr, len(arr), P, M, int(sh_degree), _lib.ptr(means3D),
                                                        _lib.ptr(shs), _lib.ptr(d_shs), _lib.ptr(d_means3D),
                                                        int(bool(accumulate or i > 0)), _stream(dev), dev.index or 0),
                   "hgs_sh_colors_batched_bwd")


def raster_views(call):
    """Test/introspection helper: typed torch views of the sorted keys, point list, tile ranges, ..."""
    lib = _lib.lib()
    v = _lib.RasterViews()
    geom, binb, img = call.geom, call.binb, call.img
    _lib.check(lib.hgs_raster_views_get(call.P, call.W, call.H, call.L_ws, call.p_geom, call.p_bin, call.p_img,
                                        C.byref(v)), "hgs_raster_views_get")

    def view(buf, addr, dtype, count):
        off = addr - buf.data_ptr()
        itemsize = torch.empty(0, dtype=dtype).element_size()
        return buf[off:off + count * itemsize].view(dtype)

    P, L, W, H = call.P, call.L, call.W, call.H
    T = ((W + 15) // 16) * ((H + 15) // 16)
    ranges = view(binb, v.ranges, torch.int32, T * 2).view(T, 2)
    # the sorted tile-id column from the ranges (the kernels write it only for a debug forward: include/hgs.h)
    counts = (ranges[:, 1] - ranges[:, 0]).long().clamp_(min=0)
    if int(counts.sum()) == L:
        tile_ids = torch.repeat_interleave(torch.arange(T, dtype=torch.int32, device=ranges.device), counts)
    else:                   # (ranges that do not add up: a workspace no forward has filled -- hand back the raw column)
        tile_ids = view(binb, v.tile_ids_sorted, torch.int32, L)
    return {
        "tile_ids_sorted": tile_ids,
        "point_list": view(binb, v.point_list, torch.int32, L),
        "ranges": view(binb, v.ranges, torch.int32, T * 2).view(T, 2),
        "tiles_touched": view(geom, v.tiles_touched, torch.int32, P),
        "offsets": view(geom, v.offsets, torch.int32, P),
        "depths": view(geom, v.depths, torch.float32, P),
        "rects": view(geom, v.rects, torch.int32, P * 2).view(P, 2),
        "records": view(geom, v.records, torch.float32, P * 16).view(P, 16),
        "final_T": view(img, v.final_T, torch.float32, H * W).view(H, W),
        "n_contrib": view(img, v.n_contrib, torch.int32, H * W).view(H, W),
    }


def mark_visible(means3D, viewmatrix, projmatrix):
    """Frustum test used by the upstream API (z > 0.2 in view space)."""
    means3D = _require_gpu(means3D, "means3D")
    vm = viewmatrix.to(torch.float32)
    z = means3D @ vm[:3, 2] + vm[3, 2]
    return z > 0.2


def lod_gather(render_indices, parent_indices, weights, means3D, scales, rotations, shs, opacities):
    """out_i = w_i * attr[render_indices[i]] + (1 - w_i) * attr[parent_indices[i]] for every given attribute
    (None entries stay None); rotations with the hemisphere flip of gaussian_renderer/__init__.py:212-216."""
    lib = _lib.lib()
    n = int(render_indices.numel())
    dev = render_indices.device
    if not render_indices.is_cuda or render_indices.dtype != torch.int32 or not parent_indices.is_cuda or \
            parent_indices.dtype != torch.int32 or parent_indices.numel() < n:
        raise RuntimeError("render_indices / parent_indices must be int32 GPU tensors (parent_indices at least as long)")
    if not weights.is_cuda or weights.dtype != torch.float32 or weights.numel() < n:
        raise RuntimeError("interpolation_weights must be a float32 GPU tensor with one entry per render index")
    ri, pi, w = render_indices.contiguous(), parent_indices.contiguous(), weights.contiguous()
    outs, ins = [], []
    for t, name in ((means3D, "means3D"), (scales, "scales"), (rotations, "rotations"), (shs, "shs"),
                    (opacities, "opacities")):
        if t is None:
            ins.append(None); outs.append(None)
            continue
        t = _require_gpu(t, name)
        ins.append(t)
        outs.append(torch.empty((n,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev))
    M = shs.shape[1] if shs is not None else 0
    p = _lib.ptr
    _lib.check(lib.hgs_lod_gather(p(ri), p(pi), p(w), n, M, *[p(t) for t in ins], *[p(t) for t in outs],
                                  _stream(dev), dev.index or 0), "hgs_lod_gather")
    return tuple(outs)


def lod_gather_backward(render_indices, parent_indices, weights, rotations, grads, shapes):
    """grads: 5-tuple of gradients of lod_gather's outputs (None allowed); shapes: shapes of the full inputs.
    Returns the 5 full-size gradients (zeros where nothing was rendered)."""
    lib = _lib.lib()
    n = int(render_indices.numel())
    dev = render_indices.device
    ri, pi, w = render_indices.contiguous(), parent_indices.contiguous(), weights.contiguous()
    gs = [None if g is None else g.to(torch.float32).contiguous() for g in grads]
    ds = [None if (g is None or s is None) else torch.zeros(*s, dtype=torch.float32, device=dev)
          for g, s in zip(gs, shapes)]
    M = shapes[3][1] if shapes[3] is not None else 0
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    p = _lib.ptr
    _lib.check(lib.hgs_lod_gather_bwd(p(ri), p(pi), p(w), n, M, p(rotations), *[p(g) for g in gs], *[p(d) for d in ds],
                                      p(flag), _stream(dev), dev.index or 0), "hgs_lod_gather_bwd")
    return tuple(ds)
